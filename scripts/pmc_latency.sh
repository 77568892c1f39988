#!/bin/bash
# Latency / queueing counters of the fused apply (separate --pmc passes).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_lat
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export REPS=3
i=0
for set in \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
  "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_ATOMIC_LEVEL_sum TCC_EA0_ATOMIC_sum" \
  "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
  "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
  "TCC_BUSY_avr TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
  "TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM" \
  "TCC_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 $R/scripts/prof_apply.py > $O/p$i.log 2>&1
  echo "pass $i rc=$? : $set"
done
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ.get('GRAFT_REPO_ROOT', os.getcwd())+'/gpurun_out/pmc_lat'
for f in sorted(glob.glob(O+'/p*/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'helmholtz_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('/')[-3], {k: sum(v)/len(v) for k,v in acc.items()})
PY
find $O -name "*kernel_trace.csv" -delete
