R=$PWD
export NSR_MLP=f16x2
for lib in neural_sim_nerf_amd/csrc/libnsr.so neural_sim_nerf_amd/csrc/ab/libnsr_v2.so neural_sim_nerf_amd/csrc/libnsr.so neural_sim_nerf_amd/csrc/ab/libnsr_v2.so; do
  echo "== $lib"; NSR_LIB_PATH=$R/$lib timeout 120 python tools/one_view.py 0 0 "" 4 2>&1 | tail -3
  NSR_LIB_PATH=$R/$lib timeout 120 python tools/bench_vjp.py 400 3 2>&1 | tail -1 | cut -c1-120
done
NSR_LIB_PATH=$R/neural_sim_nerf_amd/csrc/ab/libnsr_timing.so V=32 timeout 120 python tools/phase_timers.py 2>/dev/null | grep -E "kernel ms|GEMMs|between"
unset NSR_MLP
timeout 900 python -m pytest tests/ -q -m gpu -k "f16x2 and not full_size" 2>&1 | tail -3
