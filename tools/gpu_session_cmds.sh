timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 100 python tools/bench_vjp.py 400 3
bash tools/collect_profiles.sh > $O/collect.log 2>&1; tail -3 $O/collect.log
