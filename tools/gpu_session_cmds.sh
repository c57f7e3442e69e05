# round 4, session n: tapless attribution of the fp32 x16 input-gradient kernels' outlier rays
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_r4.py -q -s -k "fp32_x16_vjp" > $O/t.log 2>&1; tail -5 $O/t.log | cut -c1-400; grep "k_render_vjp16" $O/t.log | cut -c1-900
