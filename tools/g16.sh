cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/g16
rocprofv3 --list-avail 2>/dev/null | grep -E "^\s*(Name|gpu)|TCC_HIT|TCC_MISS|TCC_REQ|TCP_TCC_READ_REQ|TCP_TOTAL_CACHE|TCC_EA_RDREQ|TCP_TCC_READ|TA_BUSY|TCP_PENDING|TCC_BUSY|TA_TA_BUSY|TCP_TA_TCP|TCP_GATE|TCC_TAG_STALL|TCP_TCR|TCC_READ" | sort | uniq | head -60 > $R/gpurun_out/g16/counters.txt
for set in "sq:GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" "l2:GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum"; do
  name=${set%%:*}; ctrs=${set#*:}
  for v in 0 2; do
    rm -rf /tmp/pmc_$name$v
    timeout 300 rocprofv3 --kernel-trace --pmc $ctrs -d /tmp/pmc_$name$v -o p --output-format csv -- python $R/tools/bench_conv_bf16.py inception bf16_bdir=$v > /tmp/pmc_$name$v.log 2>&1
    D=$(dirname $(find /tmp/pmc_$name$v -name "*counter_collection.csv" | head -1))
    python $R/tools/pmc_summary.py $D > $R/gpurun_out/g16/pmc_${name}_bdir$v.txt 2>&1
  done
done
cat $R/gpurun_out/g16/counters.txt | head -40
head -30 $R/gpurun_out/g16/pmc_sq_bdir2.txt | cut -c1-230
head -30 $R/gpurun_out/g16/pmc_l2_bdir2.txt | cut -c1-260
