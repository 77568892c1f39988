#!/bin/bash
# Rehearsal of the N>1 path on ONE GPU (gloo, ranks share cuda:0).
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_partitioned.py -x -q > gpurun_out/part_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/part_tests.log
tail -5 gpurun_out/part_tests.log
for mode in consistent reference; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 20 \
    --warmup 5 --elems 32 --backend gloo --partitioned $mode --no-cpu-baseline \
    > gpurun_out/part_bench_$mode.log 2>&1
  echo "bench $mode exit $?"; tail -2 gpurun_out/part_bench_$mode.log
done
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --n 32 --no-cpu-baseline --no-general > gpurun_out/part_bench_single.log 2>&1
tail -1 gpurun_out/part_bench_single.log
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 \
  --master-addr 127.0.0.1 --master-port 29514 bench.py --gpus 4 --steps 10 \
  --warmup 3 --elems 32 --backend gloo --scaling strong --no-cpu-baseline \
  > gpurun_out/part_bench_strong.log 2>&1
echo "bench strong exit $?"; tail -1 gpurun_out/part_bench_strong.log | cut -c1-400
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29515 bench.py --gpus 2 --steps 10 \
  --warmup 3 --elems 16 --backend gloo --periodic --no-cpu-baseline \
  > gpurun_out/part_bench_periodic.log 2>&1
echo "bench periodic exit $?"; grep "^{" gpurun_out/part_bench_periodic.log | cut -c1-500
