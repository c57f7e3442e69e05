timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('fwd', d['value'], d['roofline'], 'vjp', d['roofline_vjp']['kernel_ms'], d['roofline_vjp']['frac'], d['parity'])"
