#!/bin/bash
# Collects what profiles/rNN holds, on the GPU box (run through gpurun from the repo root):
#   bench line, rocprofv3 kernel stats of the same command, separate --pmc passes (counters only + kernel trace) for
#   the default forward kernel (bench.py: k_render16p, global-phases schedule), the same kernel with the per-ray queue
#   (tools/one_view.py 16 0 queue: k_render16), the bf16x3 forward kernel (NSR_MLP=bf16x3 tools/one_view.py 16: k_render_b3) and the VJP kernel
#   (tools/bench_vjp.py), kernel stats of the bf16x3 bench line, of the VJP and of the hand-off kernels, and the sha256 of the kernel sources
#   that were measured (bench.py only reports a PMC figure whose hash matches the tree it runs from).
# Output under gpurun_out/prof/; tools/summarize_pmc.py rNN turns it into profiles/rNN/.
# usage: collect_profiles.sh [quick]      (quick: kernel stats + FETCH/WRITE + MFMA-busy passes only)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd $R && python -c "from neural_sim_nerf_amd import _lib; print(_lib.kernel_source_hash())" > $O/kernel_source_sha256.txt
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b3 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --mlp bf16x3 > $O/stats_b3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_vjp -- python $R/tools/bench_vjp.py > $O/stats_vjp.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_handoff -- python $R/tools/bench_handoff.py > $O/stats_handoff.log 2>&1
if [ "$1" = queue_only ]; then ONLY_QUEUE=1; fi
if [ "$1" = quick ]; then
  SETS=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE")
else
  SETS=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
        "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
        "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
        "SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES" \
        "FETCH_SIZE" "WRITE_SIZE")
fi
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_x16_$i -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $O/pmc_x16_$i.log 2>&1
  NSR_MLP=bf16x3 timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_b3_$i -- python $R/tools/one_view.py 16 > $O/pmc_b3_$i.log 2>&1
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_vjp_$i -- python $R/tools/bench_vjp.py 400 1 > $O/pmc_vjp_$i.log 2>&1
  NSR_MLP=bf16x3 timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_vjpb3_$i -- python $R/tools/bench_vjp.py 400 1 > $O/pmc_vjpb3_$i.log 2>&1
  NSR_SCHEDULE=queue timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_vjpq_$i -- python $R/tools/bench_vjp.py 400 1 > $O/pmc_vjpq_$i.log 2>&1
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_x16q_$i -- python $R/tools/one_view.py 16 0 queue > $O/pmc_x16q_$i.log 2>&1
done
timeout 100 python $R/tools/bench_vjp.py 400 3 > $O/vjp_phases.log 2>&1
NSR_SCHEDULE=queue timeout 100 python $R/tools/bench_vjp.py 400 3 > $O/vjp_queue.log 2>&1
timeout 100 python $R/tools/bench_vjp.py 400 3 32 > $O/vjp_x32.log 2>&1
NSR_MLP=bf16x3 timeout 100 python $R/tools/bench_vjp.py 400 3 > $O/vjp_bf16x3.log 2>&1
timeout 100 python $R/tools/one_view.py 16 0 queue 4 > $O/schedule_queue.log 2>&1
timeout 100 python $R/tools/one_view.py 16 0 phases 4 > $O/schedule_phases.log 2>&1
NSR_MLP=bf16x3 timeout 100 python $R/tools/one_view.py 16 0 phases 4 > $O/schedule_bf16x3.log 2>&1
timeout 100 python $R/tools/one_view.py 32 0 queue 4 > $O/schedule_x32.log 2>&1
MODES=2,12,11,2,12,11 timeout 100 python $R/tools/probe_bf16x3.py > $O/probe_bf16x3.log 2>&1
ls $O
