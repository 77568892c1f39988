#!/bin/bash
# Upper bound for an atomic-free assembly: the facet / chain kernels with the
# shared-node atomics replaced by plain stores (timing only, wrong sums:
# -DSFEM_FACET_TIMING=1, scripts/build_variant.sh t1), same box, same call.
set -e
mkdir -p gpurun_out
out=gpurun_out/exp_plain_stores.jsonl
: > $out
for lib in default t1; do
  if [ $lib = default ]; then unset SFEM_LIB; else export SFEM_LIB=$PWD/swirl_fem_amd/libsfem_hip_$lib.so; fi
  TAG=p7_f64 N=64 P=8 GEOMETRY=auto,stored python scripts/time_apply.py >> $out
  TAG=p7_f64_jitter N=64 P=8 JITTER=0.2 GEOMETRY=auto python scripts/time_apply.py >> $out
  TAG=p11_f32 N=64 P=12 DTYPE=f32 MASS=0.5 GEOMETRY=auto python scripts/time_apply.py >> $out
done
cat $out
