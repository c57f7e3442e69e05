timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25
timeout 100 python tools/bench_vjp.py
