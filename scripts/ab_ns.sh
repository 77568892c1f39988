#!/bin/bash
# A/B two builds of the library on Navier-Stokes steps.
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for v in A B; do
  cp swirl_fem_amd/libsfem_hip_$v.so swirl_fem_amd/libsfem_hip.so
  python scripts/bench_ns.py ${CASES:-tgv32} 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
  if l.startswith('{'):
    d = json.loads(l); print('$v', d['case'][:40], round(d['ms_per_step'], 2))"
done
done
