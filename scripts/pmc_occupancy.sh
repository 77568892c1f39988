#!/bin/bash
# Occupancy / dispatch counters of the fused apply (separate --pmc passes).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_occ
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export REPS=3
i=0
for set in \
  "MeanOccupancyPerCU MeanOccupancyPerActiveCU" \
  "SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
  "SPI_RA_RES_STALL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_TMP_STALL_CSN SPI_RA_WVLIM_STALL_CSN" \
  "VmemLatency MemUnitStalled" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" \
  "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" ; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 $R/scripts/prof_apply.py > $O/p$i.log 2>&1
  echo "pass $i rc=$? : $set"
done
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ.get('GRAFT_REPO_ROOT', os.getcwd())+'/gpurun_out/pmc_occ'
for f in sorted(glob.glob(O+'/p*/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'helmholtz_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('/')[-3], {k: sum(v)/len(v) for k,v in acc.items()})
PY
find $O -name "*kernel_trace.csv" -delete
