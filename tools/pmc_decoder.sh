#!/bin/bash
# PMC passes over the dense-inference bench (fused decoder kernel); results under gpurun_out/pmc_dec*/
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\bSQ_[A-Z_0-9]+" | sort -u > $R/gpurun_out/sq_counters.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_IFETCH SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_dec$i -- \
    python $R/bench.py --workload dense_inference --batch 8 --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/pmc_dec$i.log 2>&1
  echo "pass $i rc=$?"
done
