timeout 1800 python -m pytest tests/ -q -m gpu 2>&1 | tail -4
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'P'
import json, os
d = json.load(open(os.environ['O'] + '/bench.json'))
print({k: d[k] for k in ('value', 'ms_per_step', 'dtype')}); print('traffic', d['roofline'].get('traffic'), d['roofline_vjp'].get('traffic'), d['extra_workloads']['bf16x3']['roofline_vjp'].get('traffic')); print('cpu', d['cpu_baseline']['value'])
P
