# round 4, session q: N_importance 32 against the reference (g20)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "fewer_importance" > $O/t.log 2>&1; tail -8 $O/t.log | cut -c1-600
