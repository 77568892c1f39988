#!/bin/bash
# A/B two builds of the library on the same box: libsfem_hip_A.so vs _B.so
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for v in A B; do
  cp swirl_fem_amd/libsfem_hip_$v.so swirl_fem_amd/libsfem_hip.so
  python bench.py --steps 50 --warmup 10 --no-cpu-baseline $BENCH_ARGS > gpurun_out/ab_$v.json 2>gpurun_out/ab_$v.err
  python -c "
import json;d=json.loads(open('gpurun_out/ab_$v.json').read().strip().splitlines()[-1]);print('$v', round(d['value'],2),round(d['ms_per_step'],3),round(d['roofline']['kernel_ms'],4),round(d['config']['apply_ms'],4),d.get('roofline_stored_factors',{}).get('kernel_ms'))"
done
done
