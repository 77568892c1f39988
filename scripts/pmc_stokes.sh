#!/bin/bash
# PMC passes over the fused Stokes kernels (48^3, p=7): HBM traffic and issue counters.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_stokes_$tag -- python3 $R/scripts/exp_stokes.py > $O/pmc_stokes_$tag.log 2>&1; echo "pmc $tag rc=$?"
done
