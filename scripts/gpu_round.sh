#!/bin/bash
# One GPU session: parity tests, smoke, bench (both geometry paths), rocprof.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r01}
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu_$TAG.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
python bench.py --steps 50 --warmup 10 > $O/bench_$TAG.json 2> $O/bench_$TAG.err; echo "bench rc=$?"; cat $O/bench_$TAG.json
python bench.py --steps 50 --warmup 10 --geometry stored --no-cpu-baseline > $O/bench_${TAG}_stored.json 2>> $O/bench_$TAG.err; echo "bench stored rc=$?"; cat $O/bench_${TAG}_stored.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/prof_$TAG.log 2>&1; echo "rocprof rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_stored -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --geometry stored > $O/prof_${TAG}_stored.log 2>&1; echo "rocprof stored rc=$?"
for m in affine stored; do
  extra=""; [ $m = stored ] && extra="--geometry stored"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_${TAG}_${m}_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline $extra > /dev/null 2>&1; echo "pmc fetch $m rc=$?"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_${TAG}_${m}_write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline $extra > /dev/null 2>&1; echo "pmc write $m rc=$?"
done
find $O -name "*_kernel_stats.csv" | head
