# scratch command list for tools/gpu_session.sh (edited per GPU session)
timeout 1800 python -m pytest tests/ -q -m gpu 2>&1 | tail -8
timeout 900 python bench.py --cpu-sample-side 128 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err; python - <<'P'
import json,os
d=json.loads(open(os.environ["O"]+"/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","dtype")}); print("roofline_vjp", json.dumps(d["roofline_vjp"])[:600])
for k in d["extra_workloads"]:
    e=d["extra_workloads"][k]
    if "value" in e: print(k, e["value"], e.get("ms_per_view"), (e.get("roofline_vjp") or {}).get("kernel_ms"))
P
