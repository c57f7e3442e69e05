# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session AR: whole GPU suite + smoke on the final tree)
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu -x --durations=4 2>&1 | tail -10
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
