# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session AN: the layered renderer's tests incl. the bounds-checked build on all three arithmetics)
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_wide.py tests/test_gpu_r6.py -q -m gpu --durations=5 2>&1 | tail -14
