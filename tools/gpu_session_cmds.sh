# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r05 session AD: the whole GPU suite and smoke on the final tree)
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/ -q -m gpu --durations=5 > $O/gpu_suite.log 2>&1; tail -9 $O/gpu_suite.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
