# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session F: tall tiles for 128-column tiles, all cases, debug build)
cd $GRAFT_REPO_ROOT
echo "== wide tests, bf16x3"; NSR_WIDE_MLP=bf16x3 timeout 900 python -m pytest tests/test_gpu_wide.py -q -m gpu 2>&1 | tail -4
echo "== wide tests, bf16x3, bounds-checked library"; NSR_LIB_PATH=$GRAFT_REPO_ROOT/neural_sim_nerf_amd/csrc/libnsr_debug.so NSR_WIDE_MLP=bf16x3 timeout 900 python -m pytest tests/test_gpu_wide.py -q -m gpu 2>&1 | tail -4
NSR_LIB_PATH=$GRAFT_REPO_ROOT/neural_sim_nerf_amd/csrc/libnsr_debug.so python -c "
from neural_sim_nerf_amd import wide
import ctypes as C
lib = wide.load(); b, l = C.c_int(), C.c_uint(); print('rc', lib.nsrw_debug_bounds_status(C.byref(b), C.byref(l)), 'built', b.value, 'line', l.value)"
echo "== bench bf16x3"; timeout 600 python tools/bench_wide.py --mlp bf16x3 --cases ycbv,w512,d10w384,small,w1024 --steps 3 2>&1 | grep -v "^{" | cut -c1-1500
