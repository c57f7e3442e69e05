timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('fwd', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], 'vjp', d['roofline_vjp']['kernel_ms'], d['roofline_vjp']['frac'], 'c1', d['extra_workloads']['config1']['gpu'])"
python tools/one_view.py 32
