#!/bin/bash
# Issue-side + matrix-core counters of the p = 11 fp32 apply: MFMA kernel vs vector-ALU kernel.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_mfma
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export P=12 N=${N:-48} DTYPE=f32 MASS=0.5 REPS=3
SETS=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"
)
for tag in mfma valu; do
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    ( [ $tag = valu ] && export SFEM_MFMA=0; timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${tag}_p$i -- python3 $R/scripts/time_apply.py > $O/${tag}_p$i.log 2>&1 )
    echo "$tag pass $i rc=$?"
  done
done
python3 - <<'PY'
import csv, glob, os, collections, json
O=os.environ.get('GRAFT_REPO_ROOT', os.getcwd())+'/gpurun_out/pmc_mfma'
res=collections.defaultdict(dict)
for f in sorted(glob.glob(O+'/*_p*/**/*counter_collection.csv', recursive=True)):
    tag=f[len(O)+1:].split('/')[0].rsplit('_p',1)[0]
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'helmholtz' in r['Kernel_Name'] and 'setup' not in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): res[tag][k]=sum(v)/len(v)
json.dump(res, open(O+'/summary.json','w'), indent=1)
keys=sorted(set(k for d in res.values() for k in d))
print('%-30s %12s %12s' % ('counter', 'mfma', 'valu'))
for k in keys: print('%-30s %12.4g %12.4g' % (k, res['mfma'].get(k, float('nan')), res['valu'].get(k, float('nan'))))
PY
find $O -name "*kernel_trace.csv" -delete
