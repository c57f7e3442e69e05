timeout 900 python -m pytest tests -m gpu -x -q -k "full_size" 2>&1 | tail -15
