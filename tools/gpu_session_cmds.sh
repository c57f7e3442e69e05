timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'P'
import json, os
d = json.load(open(os.environ['O'] + '/bench.json'))
print({k: d[k] for k in ('value', 'ms_per_step', 'dtype')}); print('traffic', d['roofline'].get('traffic'), d['roofline_vjp'].get('traffic')); print('cpu', d['cpu_baseline']['value'])
P
