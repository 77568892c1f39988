#!/bin/bash
# A/B two builds of the library on the Stokes kernels (48^3, p = 7).
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for v in A B; do
  cp swirl_fem_amd/libsfem_hip_$v.so swirl_fem_amd/libsfem_hip.so
  echo -n "$v "; python scripts/time_stokes.py 2>/dev/null | tail -1
done
done
