#!/bin/bash
# Kernel breakdown of Taylor-Green steps with the Schwarz pressure
# preconditioner (rocprofv3 --kernel-trace --stats):  scripts/prof_ns_pc.sh [tgv32|tgv64]
R=${GRAFT_REPO_ROOT:-$(pwd)}
case=${1:-tgv32}
cd /tmp && export TMPDIR=/tmp
export SFEM_PRESSURE_PC=schwarz STEPS=3
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ns_pc -- python3 $R/scripts/bench_ns.py $case > $R/gpurun_out/prof_ns_pc.log 2>&1
f=$(find $R/gpurun_out/prof_ns_pc -name "*kernel_stats.csv" | head -n 1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', round(tot / 1e6, 1))
for r in rows[:22]:
  print(r['Name'][:80].ljust(80), r['Calls'].rjust(7), str(round(float(r['TotalDurationNs']) / 1e6, 1)).rjust(8), str(round(float(r['AverageNs']) / 1e3, 1)).rjust(9))
PY
find $R/gpurun_out/prof_ns_pc -name "*kernel_trace.csv" -delete
tail -n 1 $R/gpurun_out/prof_ns_pc.log | cut -c1-300
