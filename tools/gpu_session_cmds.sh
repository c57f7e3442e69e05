# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r05 session K: sample-count kernels with many items per workgroup)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_r5.py -q -m gpu -x -k "native_sample or stage_methods or dropin_api_takes" --durations=5 > $O/r5_tests.log 2>&1; tail -15 $O/r5_tests.log
