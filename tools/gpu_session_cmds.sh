timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 900 bash tools/collect_profiles.sh > $O/collect.log 2>&1; tail -1 $O/collect.log
