# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session W: kw_head -- tests and timings)
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_wide.py tests/test_gpu_r6.py -q -m gpu -x 2>&1 | tail -15
timeout 600 python tools/bench_wide.py --mlp f16x2 --cases ycbv,w512,small --steps 2 2>&1 | grep -v "^{" | cut -c1-700
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python $GRAFT_REPO_ROOT/tools/bench_wide.py --mlp f16x2 --cases ycbv --steps 2 --no-grad > $O/ks.log 2>&1
f=$(ls $O/ks/*/*_kernel_stats.csv | head -1); cut -c1-150 $f | head -12
