#!/bin/bash
# Round-3 evidence run: kernel-trace stats of the bench (box = default, affine,
# multilinear, stored, p = 11), HBM traffic (FETCH_SIZE / WRITE_SIZE, separate
# passes), atomics, issue counters; the fused Stokes kernels at 64^3.
#   scripts/gpu_round3.sh <tag> [what ...]     what: auto affine jitter stored p11 stokes issue
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r03}; shift
WHAT=${@:-auto affine jitter stored p11 stokes issue}
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
for w in $WHAT; do
  case $w in
    auto) run auto ;;
    affine) export SFEM_BOX=0; run affine; unset SFEM_BOX ;;
    jitter) run jitter --jitter 0.2 ;;
    stored) run stored --geometry stored ;;
    p11) run p11 $P11 ;;
    stokes)
      rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_stokes -- python3 $R/scripts/prof_stokes.py > $O/prof_${TAG}_stokes.log 2>&1; echo "stats stokes rc=$?"
      for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
        key=$(echo $c | cut -d' ' -f1)
        REPS=3 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${TAG}_stokes_$key -- python3 $R/scripts/prof_stokes.py > /dev/null 2>&1; echo "pmc stokes $key rc=$?"
      done ;;
    issue)
      i=0
      for set in \
       "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
       "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
        i=$((i+1))
        for v in auto affine; do
          [ $v = affine ] && export SFEM_BOX=0
          rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_${TAG}_${v}_sq$i -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-general > /dev/null 2>&1; echo "pmc $v sq$i rc=$?"
          unset SFEM_BOX
        done
      done ;;
  esac
done
find $O -name "*kernel_trace.csv" -size +2M -delete
# the sources this run measured (profiles/ does not travel back from the GPU
# box: run scripts/collect_profiles3.py $TAG again where gpurun_out/ is merged)
( cd $R && python3 -c "import bench; print(bench.kernel_source_hash())" ) > $O/src_hash_$TAG.txt
python3 $R/scripts/collect_profiles3.py $TAG
