#!/bin/bash
# Collects what profiles/rNN holds, on the GPU box (run through gpurun from the repo root):
#   bench line, rocprofv3 kernel stats of the same command, and separate --pmc passes (counters only + kernel trace)
#   for the default forward kernel (bench.py) and the x32 kernel (tools/one_view.py 32).
# Output under gpurun_out/prof/; tools/summarize_pmc.py turns the PMC CSVs into pmc_k_render.json.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_x16_$i -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_x16_$i.log 2>&1
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_x32_$i -- python $R/tools/one_view.py 32 > $O/pmc_x32_$i.log 2>&1
done
ls $O
