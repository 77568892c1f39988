#!/bin/bash
# Issue-side and memory-side counters of the fused apply (scripts/prof_apply.py)
# for several builds / settings.   scripts/pmc_facet.sh "<tag>:<env assignments>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_facet
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export REPS=${REPS:-3}
SETS=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum"
)
for spec in "$@"; do
  tag=${spec%%:*}; envs=${spec#*:}
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    ( export $envs; timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${tag}_p$i -- python3 $R/scripts/prof_apply.py > $O/${tag}_p$i.log 2>&1 )
    echo "$tag pass $i rc=$?"
  done
done
python3 - <<'PY'
import csv, glob, os, collections, json
O=os.environ.get('GRAFT_REPO_ROOT', os.getcwd())+'/gpurun_out/pmc_facet'
res=collections.defaultdict(dict)
for f in sorted(glob.glob(O+'/*_p*/**/*counter_collection.csv', recursive=True)):
    tag=f[len(O)+1:].split('/')[0].rsplit('_p',1)[0]
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if ('helmholtz' in k or 'stokes' in k) and 'setup' not in k:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            res[tag]['kernel']=k[:90]
    for k,v in acc.items(): res[tag][k]=sum(v)/len(v)
for f in sorted(glob.glob(O+'/*_p1/**/*kernel_trace.csv', recursive=True)):
    tag=f[len(O)+1:].split('/')[0].rsplit('_p',1)[0]
    d=[(float(r['End_Timestamp'])-float(r['Start_Timestamp']))/1e6 for r in csv.DictReader(open(f))
       if ('helmholtz' in r['Kernel_Name'] or 'stokes' in r['Kernel_Name']) and 'setup' not in r['Kernel_Name']]
    if d: res[tag]['kernel_ms_under_pmc']=sum(d)/len(d)
json.dump(res, open(O+'/summary.json','w'), indent=1)
for tag,d in res.items():
    print(tag)
    for k in sorted(d): print('   %-28s %s' % (k, ('%.4g' % d[k]) if not isinstance(d[k], str) else d[k]))
PY
find $O -name "*kernel_trace.csv" -delete
