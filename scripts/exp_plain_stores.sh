#!/bin/bash
# Upper bound for an atomic-free assembly: the facet / chain kernels with the
# shared-node atomics replaced by plain stores (timing only, wrong sums):
#   t1  = -DSFEM_FACET_TIMING=1  plain stores in slot order (through LDS)
#   t13 = -DSFEM_FACET_TIMING=13 every node stored from the lane's own slots
# built with scripts/build_variant.sh; same box, same call.
set -e
mkdir -p gpurun_out
out=gpurun_out/exp_plain_stores.jsonl
: > $out
for lib in default t1 t13; do
  if [ $lib = default ]; then unset SFEM_LIB; else export SFEM_LIB=$PWD/swirl_fem_amd/libsfem_hip_$lib.so; fi
  [ $lib = default ] || [ -f "$SFEM_LIB" ] || continue
  TAG=p7_f64 N=64 P=8 GEOMETRY=auto,stored python scripts/time_apply.py >> $out
  TAG=p7_f64_jitter N=64 P=8 JITTER=0.2 GEOMETRY=auto python scripts/time_apply.py >> $out
  TAG=p11_f32 N=64 P=12 DTYPE=f32 MASS=0.5 GEOMETRY=auto python scripts/time_apply.py >> $out
done
cat $out
