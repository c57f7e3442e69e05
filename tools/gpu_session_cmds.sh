# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session O: layered f16x2 tests)
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_wide.py tests/test_gpu_r6.py -q -m gpu 2>&1 | tail -30
