#!/bin/bash
# Kernel-level breakdown of Kolmogorov-generator steps (run on the GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kolmo -- python3 scripts/exp_graph_capture.py > $O/prof_kolmo.log 2>&1
echo "rc=$?"; tail -1 $O/prof_kolmo.log
python3 - <<PY
import csv, glob
f = glob.glob('/tmp/prof_kolmo/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
out = ['total kernel ms %.1f' % (tot / 1e6)]
for r in rows[:24]:
  out.append('%-90s %7s %9.3f ms avg %7.1f us  %5.1f%%' % (
      r['Name'][:90], r['Calls'], float(r['TotalDurationNs']) / 1e6,
      float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
open('$O/kolmo_stats.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
