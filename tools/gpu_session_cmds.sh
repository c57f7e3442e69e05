# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r05 session Z: the layered renderer's tests after the relu-NaN change)
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_wide.py -q -x 2>&1 | tail -15
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
