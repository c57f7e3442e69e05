timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 900 bash tools/collect_profiles.sh > $O/collect.log 2>&1; tail -2 $O/collect.log
