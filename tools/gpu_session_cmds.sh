# scratch command list for tools/gpu_session.sh (edited per GPU session)
R=$PWD
export NSR_MLP=f16x2
for lib in libnsr_timing t_mfmaonly libnsr_timing; do
  echo "== $lib"; NSR_LIB_PATH=$R/neural_sim_nerf_amd/csrc/ab/$lib.so V=32 timeout 120 python tools/phase_timers.py 2>&1 | grep -E "kernel ms|GEMMs|between"
done
for lib in neural_sim_nerf_amd/csrc/libnsr.so neural_sim_nerf_amd/csrc/libnsr.so; do
  echo "== $lib"; NSR_LIB_PATH=$R/$lib timeout 120 python tools/one_view.py 32 0 queue 4 2>&1 | tail -3
done
unset NSR_MLP
timeout 1500 python -m pytest tests/ -q -m gpu -k "f16x2 and not full_size" -s 2>&1 | grep -v "^$" | grep -E "^census|raw0 max|passed|failed|FAILED|Error|assert" | cut -c1-600 | tail -30
