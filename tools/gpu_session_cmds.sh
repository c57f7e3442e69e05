# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session I: whole GPU suite)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu --durations=12 2>&1 | tail -32
