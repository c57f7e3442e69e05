# round 4, session l: phase timers on the timing build
cd $GRAFT_REPO_ROOT
T=neural_sim_nerf_amd/csrc/ab/libnsr_timing.so
for mlp in f16x2 bf16x3 fp32; do echo "== $mlp"; NSR_MLP=$mlp NSR_LIB_PATH=$T V=32 timeout 120 python tools/phase_timers.py 2>&1 | grep -v amdgpu.ids; done > $O/phase_timers.txt
cat $O/phase_timers.txt | cut -c1-300
