# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session P: layered f16x2 with the gradient GEMMs on fp16 MFMAs)
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_wide.py tests/test_gpu_r6.py -q -m gpu -s 2>&1 | grep -v "^$" | grep "f16x2\|passed\|failed\|Error\|assert" | cut -c1-400 | tail -60
timeout 600 python tools/bench_wide.py --mlp f16x2 --cases ycbv,w512,d10w384,small,w1024 --steps 2 2>&1 | grep -v "^{" | cut -c1-1300 | tee $O/layered_bench_f16x2.txt
