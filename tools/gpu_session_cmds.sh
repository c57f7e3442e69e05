# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r05 session AB: staggered wave priorities in the layered GEMM)
cd $GRAFT_REPO_ROOT
for p in 0 1 0 1; do
  echo "== NSRW_GEMM_STAGGER=$p"
  NSRW_GEMM_STAGGER=$p timeout 300 python tools/bench_wide.py --cases ycbv,w512 --steps 2 --no-grad 2>&1 | grep -v "^{" | cut -c1-330
done
