# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session U: the API pieces r05 refused)
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_r6.py -q -m gpu -x -k "sample_pdf or embedder or adopted" 2>&1 | tail -30
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "noviewdirs or importance or census_full" --durations=5 2>&1 | tail -30
