# scratch command list for tools/gpu_session.sh (edited per GPU session)
timeout 1800 python -m pytest tests/ -q -m gpu 2>&1 | tail -8
bash tools/collect_profiles.sh > $O/collect.log 2>&1; tail -5 $O/collect.log
