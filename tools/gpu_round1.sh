#!/bin/bash
# ad-hoc GPU session 1 (r02): tests, bench, VJP kernel stats + PMC
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
timeout 120 python tools/bench_vjp.py > $O/vjp.json 2>&1; cat $O/vjp.json
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/vjp_stats -- python $R/tools/bench_vjp.py > $O/vjp_stats.log 2>&1
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_vjp_$i -- python $R/tools/bench_vjp.py > $O/pmc_vjp_$i.log 2>&1
done
find $O -name "*kernel_stats.csv" | head -3
f=$(find $O/vjp_stats -name "*kernel_stats.csv" | head -1); head -8 $f
