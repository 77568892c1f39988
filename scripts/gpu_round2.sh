#!/bin/bash
# Round-2 evidence run: kernel-trace stats of the bench (auto + stored + p=11),
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes), atomics and issue
# counters of the dominant kernels.   scripts/gpu_round2.sh <tag>
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r02}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P11="--p 11 --n 64 --dtype f32 --mass-coeff 0.5"
run() { # name, bench flags
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_$name -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-general "$@" > $O/prof_${TAG}_$name.log 2>&1; echo "stats $name rc=$?"
  for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    key=$(echo $c | cut -d' ' -f1)
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${TAG}_${name}_$key -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-general "$@" > /dev/null 2>&1; echo "pmc $name $key rc=$?"
  done
}
run auto
run stored --geometry stored
run jitter --jitter 0.2
run p11 $P11
for set in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
 "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_${TAG}_p11_sq$i -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-general $P11 > /dev/null 2>&1; echo "pmc p11 sq$i rc=$?"
done
find $O -name "*kernel_trace.csv" -size +2M -delete
python3 $R/scripts/collect_profiles2.py $TAG
