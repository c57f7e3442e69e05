# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session Q: ablations of kw_gemm_h2, then the whole GPU suite)
cd $GRAFT_REPO_ROOT
MLP=f16x2 bash tools/ab_wide.sh $O/ab_wide_h2_w512.txt w512
timeout 1800 python -m pytest tests -q -m gpu --durations=12 2>&1 | tail -25
