# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r05 session W: gradient at the reference's depths)
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_wide.py -q -x -s 2>&1 | grep -E "autograd|passed|failed|Error|error" | head -30
