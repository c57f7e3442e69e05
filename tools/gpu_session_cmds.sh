# final collection of the round (kernel sources frozen after this)
timeout 1800 python -m pytest tests/ -q -m gpu 2>&1 | tail -5
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/collect_profiles.sh > $O/collect.log 2>&1; tail -3 $O/collect.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'P'
import json, os
d = json.load(open(os.environ['O'] + '/bench.json'))
print({k: d[k] for k in ('value', 'ms_per_step', 'dtype')}); print('roofline', json.dumps(d['roofline'])[:400]); print('cpu', d['cpu_baseline']['value']); print('parity', json.dumps(d['parity'])[:300])
P
