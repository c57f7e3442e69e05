#!/bin/bash
# PMC passes over the layered renderer's GEMM kernel (kw_gemm): tools/pmc_wide.sh <outdir> [case] [hw]
# separate rocprofv3 --pmc passes with --kernel-trace only (MI355X_MICROARCH.md); summarised by tools/pmc_wide.py
R=${GRAFT_REPO_ROOT:-$PWD}
O=$1; CASE=${2:-w512}; HW=${3:-200}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
      "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA" \
      "FETCH_SIZE" "WRITE_SIZE")
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$i -- python $R/tools/bench_wide.py ${MLP:+--mlp $MLP} --cases $CASE --hw $HW --steps 1 --no-grad > $O/pmc_$i.log 2>&1
done
python $R/tools/pmc_wide.py $O > $O/pmc_summary.json 2>&1
cat $O/pmc_summary.json
