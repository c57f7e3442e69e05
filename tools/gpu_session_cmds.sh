timeout 900 python -m pytest tests/ -q -m gpu -k "graph_capturable or degenerate or debug_bounds" 2>&1 | tail -15
