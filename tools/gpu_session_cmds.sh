A=neural_sim_nerf_amd/csrc
python tools/ab_h2.py --n 8 $A/libnsr.so $A/ab/libnsr_pp1.so $A/ab/libnsr_pp1e.so $A/libnsr.so
NSR_MLP=f16x2 NSR_LIB_PATH=$A/ab/libnsr_timing_pp1e.so V=32 timeout 120 python tools/phase_timers.py 2>&1 | grep -v amdgpu.ids
