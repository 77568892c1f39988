#!/bin/bash
# Builds swirl_fem_amd/libsfem_hip_<name>.so: the current objects with the
# translation units named in UNITS (default: the fp64 facet kernels) recompiled
# with extra flags -- and, with PATCH=<file>, from a scratch copy of csrc/ with
# that patch applied.  The shipping sources carry no timing-only code: the
# ablations of profiles/r03_facet_notes.md live in
# scripts/ablations/facet_timing.patch (-DSFEM_FACET_TIMING=1|3|..|13,
# -DSFEM_FACET_XCD=1, -DSFEM_STOKES_TIMING=10|11; results wrong, traffic right).
#   PATCH=scripts/ablations/facet_timing.patch UNITS="sfem_helmholtz_facet_f64" \
#     scripts/build_variant.sh t1 -DSFEM_FACET_TIMING=1
#   SFEM_LIB=$PWD/swirl_fem_amd/libsfem_hip_t1.so python scripts/time_apply.py
set -e
name=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
src="$root/swirl_fem_amd/csrc"
units=${UNITS:-sfem_helmholtz_facet_f64}
work="$src"
if [ -n "$PATCH" ]; then
  patch_file="$(cd "$(dirname "$PATCH")" && pwd)/$(basename "$PATCH")"
  work=$(mktemp -d /tmp/sfem_variant_XXXX)
  mkdir -p "$work/swirl_fem_amd" "$work/include"
  cp -r "$src" "$work/swirl_fem_amd/csrc"
  cp "$root/include/sfem.h" "$work/include/"
  (cd "$work" && patch -p1 -s < "$patch_file")
  work="$work/swirl_fem_amd/csrc"
fi
cd "$work"
objs=""
for o in sfem_core sfem_basis sfem_fdm sfem_cg_ensemble sfem_interp_f64 sfem_interp_f32 sfem_helmholtz sfem_helmholtz_f64_3d sfem_helmholtz_f64_2d sfem_helmholtz_f32_3d sfem_helmholtz_f32_2d sfem_helmholtz_cluster_f64 sfem_helmholtz_cluster_f32 sfem_helmholtz_mfma sfem_helmholtz_facet_f64 sfem_helmholtz_facet_f32 sfem_helmholtz_facet_f64_hi sfem_helmholtz_facet_f32_hi sfem_stokes_facet_f64 sfem_stokes_facet_f32 sfem_stokes sfem_stokes_f64_3d sfem_stokes_f64_2d sfem_stokes_f32_3d sfem_stokes_f32_2d; do
  if [[ " $units " == *" $o "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wall -Wno-unused-function -Wno-array-bounds -I../../include -I. "$@" -c $o.hip -o /tmp/${o}_$name.o
    objs="$objs /tmp/${o}_$name.o"
  else
    objs="$objs $src/$o.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/swirl_fem_amd/libsfem_hip_$name.so" $objs
echo built libsfem_hip_$name.so
