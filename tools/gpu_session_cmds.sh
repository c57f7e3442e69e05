# round 4, session h: sigma-only coarse pass; A/B against the previous forward kernel
cd $GRAFT_REPO_ROOT
L=neural_sim_nerf_amd/csrc
timeout 300 python tools/ab_h2.py --n 8 $L/libnsr.so $L/ab/libnsr_norange.so $L/libnsr.so $L/ab/libnsr_norange.so 2>&1 | tee $O/ab.txt
timeout 1500 python -m pytest tests/test_gpu_r4.py -q -s -x -k "coarse_pass_without or native_importance or out_of_range" > $O/r4.log 2>&1; tail -5 $O/r4.log; grep "400x400 views\|kernel ms per" $O/r4.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "render_path_api or stagewise or f16x2 or graph or c_host" > $O/par.log 2>&1; tail -3 $O/par.log
