# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session AK: random fused-expressible shapes through the drop-in API)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_r6.py -q -m gpu -k "re_expression" 2>&1 | tail -40
