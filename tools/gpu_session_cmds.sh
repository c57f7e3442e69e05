# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r05 session H: the native sample counts + the trust change)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_r5.py -q -m gpu -x --durations=10 > $O/r5_tests.log 2>&1; tail -30 $O/r5_tests.log
timeout 300 python tools/ab_h2.py --n 8 neural_sim_nerf_amd/csrc/libnsr.so neural_sim_nerf_amd/csrc/libnsr.so 2>&1 | tee $O/ab.txt
