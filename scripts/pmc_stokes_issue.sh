#!/bin/bash
# Issue-side counters of the Stokes pair as E issues it (run on the GPU box):
#   scripts/pmc_stokes_issue.sh  ->  gpurun_out/pmc_stokes_issue.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
i=0
for set in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  REPS=3 timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_stokes_sq$i -- python3 scripts/prof_stokes.py > /dev/null 2>&1; echo "pmc sq$i rc=$?"
done
python3 - <<PY
import csv, glob, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_stokes_sq*/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'stokes' in k and 'box' in k:
      out[k[:40]][r['Counter_Name']].append(float(r['Counter_Value']))
lines = []
for k, c in out.items():
  lines.append(k)
  for name, v in sorted(c.items()):
    lines.append('  %-24s %.4g' % (name, sum(v) / len(v)))
open('$O/pmc_stokes_issue.txt', 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
