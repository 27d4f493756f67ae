# PMC passes over tools/wino_one.py (run on the GPU box through gpurun): where do the Winograd kernel's waves wait?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD" \
           "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU"; do
  rm -rf /tmp/pmcout
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmcout -o p --output-format csv -- python $R/tools/wino_one.py > /tmp/pmc.log 2>&1
  echo "== $set (rc=$?)"
  D=$(dirname $(find /tmp/pmcout -name "*counter_collection.csv" | head -1))
  python $R/tools/pmc_summary.py $D 2>&1 | grep -i "wino" | head -4
done
