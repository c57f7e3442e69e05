mkdir -p $O/extra
timeout 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bf16x3 or b3 or graph" 2>&1 | tail -15
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/extra/bench_default.json 2> $O/extra/bench_default.err; python - <<'PY'
import json,os
l=[x for x in open(os.environ['O']+'/extra/bench_default.json') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('value', d['value'], 'roofline', d['roofline']['achieved'], d['roofline']['frac'])
    print(json.dumps(d['extra_workloads'].get('bf16x3'), indent=0)[:1800])
else:
    print(open(os.environ['O']+'/extra/bench_default.err').read()[-1500:])
PY
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --mlp bf16x3 2>/dev/null | tail -1 | cut -c1-1500
