# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r05 session R: persistent GEMM tiles, K-stage A/B)
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_wide.py -x -q > $O/wide.log 2>&1; tail -3 $O/wide.log
for ks in 32 16; do
  echo "== NSRW_GEMM_KS=$ks"
  NSRW_GEMM_KS=$ks timeout 600 python tools/bench_wide.py --cases ycbv,w512,small --steps 2 2>&1 | grep -v "^{" | cut -c1-700
done > $O/bench_wide_ab.txt 2>&1
cat $O/bench_wide_ab.txt
