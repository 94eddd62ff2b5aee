#!/bin/bash
# PMC passes over an arbitrary command: tools/pmc_kernel.sh <outname> <kernel-substring> -- cmd...
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; pat=$2; shift 3
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  (cd $R && timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${name}$i -- "$@" > $R/gpurun_out/pmc_${name}$i.log 2>&1)
done
python3 - <<PY
import csv,glob,collections
for d in (1,2,3):
    fs=glob.glob('$R/gpurun_out/pmc_${name}%d/*/*_counter_collection.csv'%d)
    if not fs: print('no output for pass',d); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k=r['Kernel_Name']
        if '$pat' in k:
            agg[k[:60]][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in agg.items(): print(d,k,{a:round(b) for a,b in v.items()})
PY
