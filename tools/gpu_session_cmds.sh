# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session AH: random network shapes through the layered renderer)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_wide.py -q -m gpu -k "random_network_shapes or c_host or options" 2>&1 | tail -40
