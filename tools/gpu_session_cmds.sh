# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session AL: smoke + default bench on the final tree; two ranks sharing the GPU)
cd $GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "scaling", "vs_baseline")})
print(d["roofline"]["frac"], d["roofline"]["issued_frac"], d["cpu_baseline"]["value"], d["parity_summary"], sorted(d["extra_workloads"]))
PY
timeout 600 python bench.py --gpus 2 --backend gloo --share-gpu --steps 3 --warmup 1 2>&1 | tail -1 | cut -c1-400
