# round 4, session f: kernels specialised to N_importance 64 / 32
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -s -k "fewer_importance" > $O/ni.log 2>&1; tail -15 $O/ni.log
timeout 1500 python -m pytest tests/test_gpu_r4.py -q -s -x -k "native_importance or full_size" > $O/r4.log 2>&1; tail -5 $O/r4.log; grep "kernel ms per" $O/r4.log
L=neural_sim_nerf_amd/csrc
timeout 300 python tools/ab_h2.py --n 8 $L/libnsr.so $L/ab/libnsr_norange.so $L/libnsr.so 2>&1 | tee $O/ab.txt
