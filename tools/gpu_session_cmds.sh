timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "per_ray_bounds or stochastic or given_view or ndc_staticcam" 2>&1 | tail -15
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<'P'
import json, os
d = json.load(open(os.environ['O'] + '/bench.json'))
print(d['value'], d['ms_per_step']); print(json.dumps(d['extra_workloads']['render_options']))
P
tail -3 $O/bench.err
