# what the last GPU session of round 4 ran (scratch file: tools/gpu_session.sh <label> executes it on the gpurun box)
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/ -q -m gpu > $O/gpu_suite.log 2>&1; tail -6 $O/gpu_suite.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
