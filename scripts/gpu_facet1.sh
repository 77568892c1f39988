#!/bin/bash
# first run of the facet kernels: parity tests, then timings of the variants
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_facet.py -x -q > $O/facet_tests.log 2>&1
rc=$?; tail -5 $O/facet_tests.log; echo "tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
for spec in "facet_box:SFEM_FACET=1" "facet_affine:SFEM_BOX=0" "rows:SFEM_FACET=0"; do
  tag=${spec%%:*}; envs=${spec#*:}
  ( export $envs TAG=$tag GEOMETRY=auto,multilinear,stored; timeout -k 10 300 python scripts/time_apply.py ) > $O/facet_time_$tag.json 2> $O/facet_time_$tag.err
  echo "$tag rc=$?"; cat $O/facet_time_$tag.json
done
