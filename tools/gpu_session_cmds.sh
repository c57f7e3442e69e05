# round 4, session c: new tests after the gradient entry scale 2^7; full parity suite; VJP A/B timing
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_r4.py -q -s > $O/r4_tests.log 2>&1; tail -6 $O/r4_tests.log
grep -E "vjp census" $O/r4_tests.log | grep -o "^vjp census[^{]*\|'rays_above_thr': [0-9]*\|'unattributed': [0-9]*\|'flipped_units_total': [0-9]*\|'per_point_[a-z0-9]*': [0-9.e-]*\|'replay_max': [0-9.e-]*\|'err_max': [0-9.e-]*" | tr '\n' ' '; echo
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_handoff.py tests/test_data_readers.py -q -m gpu -x > $O/parity.log 2>&1; tail -6 $O/parity.log
NSR_MLP=f16x2 python tools/bench_vjp.py 400 4 2>&1 | tail -2
