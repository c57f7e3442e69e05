# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r05 session L: the whole GPU suite on the final tree,
# smoke, then the profile collection of profiles/r05)
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/ -q -m gpu --durations=8 > $O/gpu_suite.log 2>&1; tail -14 $O/gpu_suite.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/collect_profiles.sh > $O/collect.log 2>&1; tail -3 $O/collect.log
timeout 120 python tools/probe_h2.py > $O/probe_h2.txt 2>&1; tail -3 $O/probe_h2.txt
