#!/bin/bash
# timings of library variants / env settings:  gpu_facet2.sh "tag:ENV=.. ENV=.." ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
for spec in "$@"; do
  tag=${spec%%:*}; envs=${spec#*:}
  ( export $envs TAG=$tag; timeout -k 10 300 python scripts/time_apply.py ) > $O/facet_time_$tag.json 2> $O/facet_time_$tag.err
  echo "$tag rc=$?"; python3 -c "
import json,sys
d=json.loads(open('$O/facet_time_$tag.json').read().strip().splitlines()[-1])
print('  ', {k:(v['median_ms'],v['min_ms']) for k,v in d.items() if isinstance(v,dict)})"
done
