R=$PWD; P=$R/gpurun_out/prof; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
        "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
        "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
        "SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES" \
        "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 100 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $P/pmc_x16q_$i -- python $R/tools/one_view.py 16 0 queue > $P/pmc_x16q_$i.log 2>&1
done
ls $P | grep x16q | head -3
