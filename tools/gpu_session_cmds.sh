# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r05 session Y: chunking at scale, the C host of the layered renderer)
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_wide.py -q -x -s -k "scale or c_host" 2>&1 | grep -E "chunks|c_host_wide|passed|failed|Error|error|assert" | head -30
