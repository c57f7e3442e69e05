#!/bin/bash
# PMC passes (counters only, with --kernel-trace) over one view for x16 (2 WG/CU), x16 (1 WG/CU) and x32.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "16 0" "16 256" "32 0"; do
  tag=$(echo $cfg | tr ' ' '_')
  i=0
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
             "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
             "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc2/${tag}_$i -- python $R/tools/one_view.py $cfg > $R/gpurun_out/pmc2/${tag}_$i.log 2>&1
  done
done
