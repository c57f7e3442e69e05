# round 4, session p: compiler-flag lottery on the shipped kernels (A/B, one box)
cd $GRAFT_REPO_ROOT
L=neural_sim_nerf_amd/csrc
timeout 600 python tools/ab_h2.py --n 8 $L/libnsr.so $L/ab/libnsr_nomisched.so $L/ab/libnsr_rcprio.so $L/ab/libnsr_o2.so $L/ab/libnsr_maxilp.so $L/libnsr.so $L/ab/libnsr_nomisched.so 2>&1 | tee $O/ab_flags.txt
for lib in $L/libnsr.so $L/ab/libnsr_nomisched.so; do echo "vjp $lib"; NSR_LIB_PATH=$lib NSR_MLP=f16x2 python tools/bench_vjp.py 400 4 2>/dev/null | tail -1 | cut -c1-120; done | tee -a $O/ab_flags.txt
