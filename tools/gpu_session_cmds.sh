timeout 600 python -m pytest tests/ -q -m gpu -k "twelve_orders" -s 2>&1 | grep -E "per-ray|passed|failed|assert" | cut -c1-300
