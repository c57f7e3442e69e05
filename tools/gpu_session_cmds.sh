# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session M: whole GPU suite + smoke + bench)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu --durations=15 2>&1 | tail -40
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "dtype")}, d["roofline"]["frac"], d["roofline"]["issued_frac"])
print(json.dumps(d["extra_workloads"].get("trained"))[:3000])
print(json.dumps(d["extra_workloads"].get("api_overhead"))[:1500])
PY
