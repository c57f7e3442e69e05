cd $GRAFT_REPO_ROOT
python tools/experiments/debug_f32.py 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_gpu_wide.py tests/test_gpu_r6.py -q -m gpu -x 2>&1 | tail -6
for old in 0 1; do
echo "NSRW_F32_OLD=$old"
NSRW_F32_OLD=$old timeout 900 python tools/bench_wide.py --mlp fp32 --cases ycbv,w512,d10w384,small,w1024 --steps 2 2>&1 | grep -v "^{" | cut -c1-330
done
