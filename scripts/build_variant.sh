#!/bin/bash
# Builds swirl_fem_amd/libsfem_hip_<name>.so: the current objects with the
# translation units named in UNITS (default: the fp64 cluster kernels)
# recompiled with extra flags.   scripts/build_variant.sh <name> <flags...>
set -e
name=$1; shift
cd "$(dirname "$0")/../swirl_fem_amd/csrc"
units=${UNITS:-sfem_helmholtz_cluster_f64}
objs=""
for o in sfem_core sfem_basis sfem_helmholtz sfem_helmholtz_f64_3d sfem_helmholtz_f64_2d sfem_helmholtz_f32_3d sfem_helmholtz_f32_2d sfem_helmholtz_cluster_f64 sfem_helmholtz_cluster_f32 sfem_helmholtz_mfma sfem_helmholtz_facet_f64 sfem_helmholtz_facet_f32 sfem_helmholtz_facet_f64_hi sfem_helmholtz_facet_f32_hi sfem_stokes_facet_f64 sfem_stokes_facet_f32 sfem_stokes sfem_stokes_f64_3d sfem_stokes_f64_2d sfem_stokes_f32_3d sfem_stokes_f32_2d; do
  if [[ " $units " == *" $o "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wall -Wno-unused-function -Wno-array-bounds -I../../include -I. "$@" -c $o.hip -o /tmp/${o}_$name.o
    objs="$objs /tmp/${o}_$name.o"
  else
    objs="$objs $o.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsfem_hip_$name.so $objs
echo built libsfem_hip_$name.so
