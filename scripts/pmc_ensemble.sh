#!/bin/bash
# Memory-side atomic requests and HBM traffic of the kernels of an ensemble step
# (B flows of the Kolmogorov generator, scripts/prof_ensemble.py); separate passes.
#   B=8 STEPS=4 bash scripts/pmc_ensemble.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export B=${B:-8} STEPS=${STEPS:-4}
for c in "TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum" FETCH_SIZE WRITE_SIZE; do
  key=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_ens_$key -- python3 $R/scripts/prof_ensemble.py > $O/pmc_ens_$key.log 2>&1; echo "pmc $key rc=$?"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$O/pmc_ens_*/**/*_counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
with open('$O/pmc_ens_summary.txt', 'w') as out:
  for k, v in sorted(acc.items(), key=lambda kv: -len(next(iter(kv[1].values())))):
    if 'sfem' not in k: continue
    row = {c: sum(x) / len(x) for c, x in v.items()}
    out.write(f"{k:70s} launches {len(next(iter(v.values()))):5d}  " + '  '.join(f'{c}={x:.4g}' for c, x in sorted(row.items())) + '\n')
print(open('$O/pmc_ens_summary.txt').read())
PY
find $O -name "*kernel_trace.csv" -size +1M -delete
find $O -name "*counter_collection.csv" -size +1M -delete
