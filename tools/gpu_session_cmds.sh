# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r05 session AC: layered tests + smoke + the API tests that
# route to the layered renderer, after the shared-workspace change on the Python side)
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_wide.py tests/test_gpu_r5.py tests/test_gpu_parity.py -q -k "layered or dropin or importance or wide or c_host" 2>&1 | tail -4
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 200 python tools/bench_wide.py --cases small --steps 1 2>&1 | grep -v "^{" | cut -c1-300
