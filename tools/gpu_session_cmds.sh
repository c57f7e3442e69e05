# round 4, session a: the new tests (VJP census, range safety net, full-size default kernels) + A/B of the experiment builds
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_r4.py -q -s -x > $O/r4_tests.log 2>&1; tail -5 $O/r4_tests.log
grep -E "vjp census|range status|per-ray" $O/r4_tests.log | cut -c1-900
timeout 600 python -m pytest tests/test_gpu_parity.py -q -s -k "f16x2_vjp_over or out_of_range or test_render_rays_vjp or odd_ray_count" > $O/parity_sel.log 2>&1; tail -3 $O/parity_sel.log
grep -E "f16x2 VJP, cot|per-ray relative" $O/parity_sel.log | cut -c1-900
L=neural_sim_nerf_amd/csrc
timeout 600 python tools/ab_h2.py --n 8 $L/libnsr.so $L/ab/libnsr_norange.so $L/ab/libnsr_samenet.so $L/ab/libnsr_halfbar.so $L/ab/libnsr_nofrag.so $L/libnsr.so $L/ab/libnsr_norange.so 2>&1 | tee $O/ab.txt
