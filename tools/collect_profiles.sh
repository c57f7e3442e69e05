#!/bin/bash
# Collects what profiles/rNN holds, on the GPU box (run through gpurun from the repo root):
#   the default bench line (f16x2 since r03) and the rocprofv3 kernel stats of the same command; kernel stats of the fp32 and
#   bf16x3 bench lines, of the three input-gradient kernels and of the hand-off kernels; separate --pmc passes (counters only +
#   kernel trace) per forward kernel (k_render_h2, k_render16p, k_render_b3) and per input-gradient kernel (k_render_vjp_h2,
#   k_render_vjp16p, k_render_vjp_b3); un-profiled timings; in-kernel phase timers if the diagnostic build is there; and the
#   sha256 of the kernel sources that were measured (bench.py only reports a PMC figure whose hash matches its tree).
# Output under gpurun_out/prof/; tools/summarize_pmc.py rNN turns it into profiles/rNN/.
# usage: collect_profiles.sh [quick]      (quick: MFMA-busy + FETCH/WRITE passes only)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd $R && python -c "from neural_sim_nerf_amd import _lib; print(_lib.kernel_source_hash())" > $O/kernel_source_sha256.txt
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
for mlp in f16x2 fp32 bf16x3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$mlp -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --mlp $mlp > $O/stats_$mlp.log 2>&1
  NSR_MLP=$mlp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_vjp_$mlp -- python $R/tools/bench_vjp.py 400 2 > $O/stats_vjp_$mlp.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_handoff -- python $R/tools/bench_handoff.py > $O/stats_handoff.log 2>&1
FULL=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
      "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
      "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" \
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM" \
      "FETCH_SIZE" "WRITE_SIZE")
QUICK=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE")
# the default kernels (f16x2 forward and input-gradient) get every pass, the others MFMA-busy / clock / traffic only
pmc() {   # tag, mlp, command..., with the counter sets in SETS
  local tag=$1 mlp=$2; shift 2
  local i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    NSR_MLP=$mlp timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_${tag}_$i -- "$@" > $O/pmc_${tag}_$i.log 2>&1
  done
}
if [ "$1" = quick ]; then SETS=("${QUICK[@]}"); else SETS=("${FULL[@]}"); fi
pmc h2 f16x2 python $R/tools/one_view.py 0
pmc vjph2 f16x2 python $R/tools/bench_vjp.py 400 1
SETS=("${QUICK[@]}")
pmc x16 fp32 python $R/tools/one_view.py 0
pmc b3 bf16x3 python $R/tools/one_view.py 0
pmc vjp fp32 python $R/tools/bench_vjp.py 400 1
pmc vjpb3 bf16x3 python $R/tools/bench_vjp.py 400 1
for mlp in f16x2 fp32 bf16x3; do
  NSR_MLP=$mlp timeout 100 python $R/tools/one_view.py 0 0 "" 4 > $O/schedule_$mlp.log 2>&1
  NSR_MLP=$mlp timeout 100 python $R/tools/bench_vjp.py 400 3 > $O/vjp_$mlp.log 2>&1
done
T=$R/neural_sim_nerf_amd/csrc/ab/libnsr_timing.so
if [ -f $T ]; then
  for mlp in f16x2 bf16x3; do
    echo "== $mlp" >> $O/phase_timers.txt; NSR_MLP=$mlp NSR_LIB_PATH=$T V=32 timeout 120 python $R/tools/phase_timers.py >> $O/phase_timers.txt 2>&1
  done
fi
NSR_MLP=f16x2 timeout 200 python $R/tools/bench_path_grad.py > $O/path_grad_f16x2.json 2> /dev/null
timeout 200 python $R/tools/bench_api_overhead.py > $O/api_overhead.json 2> /dev/null
# the kernels specialised to other sample counts (r04: N_importance 64 / 32; r05: 96, N_samples 32 / 128), forward and input gradient
timeout 200 python - > $O/importance_counts.txt 2>&1 <<PY
import sys, numpy as np
sys.path.insert(0, "$R")
from neural_sim_nerf_amd import synthetic as S
from neural_sim_nerf_amd.engine import NsrModel
c = S.synth_weights(0); f = S.synth_weights(1000, fine_of=c)
pose = S.sweep_poses(1, 0)[0]
cot = np.random.RandomState(1).standard_normal((160000, 3)).astype(np.float32)
for ns, ni in ((64, 128), (64, 96), (64, 64), (64, 32), (32, 64), (128, 128)):
    m = NsrModel(c, f, n_importance=ni, n_samples=ns)
    t = []
    for _ in range(5):
        m.render_views(pose, 400, 400, S.YCBV_K, S.YCBV_NEAR, S.YCBV_FAR); t.append(m.last_kernel_ms())
    ro, rd = m.get_rays(400, 400, S.YCBV_K, pose)
    v = []
    for _ in range(4):
        m.render_rays_vjp(ro.reshape(-1, 3), rd.reshape(-1, 3), S.YCBV_NEAR, S.YCBV_FAR, cot); v.append(m.last_kernel_ms())
    print("N_samples %3d, N_importance %3d (kernels for %3d + %3d): forward %.2f ms, forward + input gradient %.2f ms per 400x400 view (medians after warm-up)"
          % (ns, ni, m.n_samples, m.ni_kernel, float(np.median(t[1:])), float(np.median(v[1:]))))
    m.close()
PY
ls $O
# ---- the layered renderer (r06): one 400x400 view per network and arithmetic (whole launch calls), kernel stats of the default arithmetic
# forward + input gradient, PMC passes over its GEMM kernels at 200x200 (counters summed over all dispatches of the run)
cd $R
for mlp in f16x2 bf16x3 fp32; do
  timeout 900 python tools/bench_wide.py --mlp $mlp --cases ycbv,w512,d10w384,small,w1024 --steps 2 2>/dev/null | grep -v "^{" > $O/layered_bench_$mlp.txt
done
cd /tmp
for c in ycbv w512; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_layered_$c -- python $R/tools/bench_wide.py --mlp f16x2 --cases $c --steps 2 > $O/stats_layered_$c.log 2>&1
done
cd $R
MLP=f16x2 bash tools/pmc_wide.sh $O/layered_pmc_f16x2_w512 w512 200 > /dev/null 2>&1
MLP=f16x2 bash tools/pmc_wide.sh $O/layered_pmc_f16x2_ycbv ycbv 200 > /dev/null 2>&1
MLP=bf16x3 bash tools/pmc_wide.sh $O/layered_pmc_bf16x3_w512 w512 200 > /dev/null 2>&1
ls $O
