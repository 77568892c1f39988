#!/bin/bash
# Atomic requests of the fused apply, sorted vs slot-order scatter.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_atomics
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export REPS=3
for v in 1 0; do
  export SFEM_SORTED_SCATTER=$v
  timeout -k 10 200 rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_ATOMIC_sum --kernel-trace --output-format csv -d $O/s$v -- python3 $R/scripts/prof_apply.py > $O/s$v.log 2>&1
  echo "sorted=$v rc=$?"
done
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ.get('GRAFT_REPO_ROOT', os.getcwd())+'/gpurun_out/pmc_atomics'
for f in sorted(glob.glob(O+'/s*/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'helmholtz_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('/')[-3], {k: round(sum(v)/len(v)) for k,v in acc.items()})
PY
find $O -name "*kernel_trace.csv" -delete
