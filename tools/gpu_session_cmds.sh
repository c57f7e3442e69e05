timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 100 python tools/bench_vjp.py 400 3
timeout 100 python tools/one_view.py 16
