#!/bin/bash
# Kernel-level breakdown of the Taylor-Green step (run on the GPU box):
#   scripts/prof_ns_stats.sh [N] [STEPS]  ->  gpurun_out/ns_stats_<N>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-64}; STEPS=${2:-2}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
N=$N STEPS=$STEPS timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ns_$N -- python3 scripts/prof_ns.py > $O/prof_ns_$N.log 2>&1
echo "rc=$?"
grep -E "iters|total|setup" $O/prof_ns_$N.log
python3 - <<PY
import csv, glob
f = glob.glob('/tmp/prof_ns_$N/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
out = ['total kernel ms %.1f' % (tot / 1e6)]
for r in rows[:32]:
  out.append('%-100s %7s %9.3f ms avg %8.1f us  %5.1f%%' % (
      r['Name'][:100], r['Calls'], float(r['TotalDurationNs']) / 1e6,
      float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
open('$O/ns_stats_$N.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
