# round 4, session o: use_viewdirs=False: all five output rows vs the reference's raw, c2w_staticcam ignored (g19)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "noviewdirs" > $O/t.log 2>&1; tail -8 $O/t.log | cut -c1-500
