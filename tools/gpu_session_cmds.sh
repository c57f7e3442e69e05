# scratch command list for tools/gpu_session.sh (edited per GPU session)
tools/build/probe_f16_mfma 2>&1 | tail -5
timeout 1500 python -m pytest tests/ -x -q -m gpu -k "census" -s 2>&1 | grep -v "^$" | tail -40
timeout 900 python -m pytest tests/ -x -q -m gpu -k "not census" 2>&1 | tail -3
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err; python - <<'P'
import json,os
d=json.loads(open(os.environ["O"]+"/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","dtype")}); print("cpu", d["cpu_baseline"]); print("parity", json.dumps(d["parity"]))
print("b3 parity", json.dumps(d["extra_workloads"]["bf16x3"].get("parity"))); print("config1", json.dumps(d["extra_workloads"]["config1"].get("parity")))
P
