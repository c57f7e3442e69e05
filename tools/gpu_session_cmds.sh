# round 4, session e: ray-granular fallback; census-based tests of the render options
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_r4.py -q -x -k "range or nan or warns or full_size" > $O/r4_tests.log 2>&1; tail -4 $O/r4_tests.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "degenerate or stochastic or noviewdirs or other_shapes or fewer_importance or f16x2 or c_host or graph" > $O/parity_sel.log 2>&1; tail -12 $O/parity_sel.log
L=neural_sim_nerf_amd/csrc
timeout 300 python tools/ab_h2.py --n 8 $L/libnsr.so 2>&1 | tee $O/ab.txt
