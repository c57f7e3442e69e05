# scratch command list for tools/gpu_session.sh (edited per GPU session)
timeout 1500 python -m pytest tests/ -q -m gpu -k "f16x2 or bf16x3_stagewise or options or (census and not full_size)" -s 2>&1 | grep -v "^$" | grep -E "census|raw0 max|passed|failed|FAILED|Error|assert" | cut -c1-600 | tail -40
timeout 900 python bench.py --cpu-sample-side 128 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err; python - <<'P'
import json,os
d=json.loads(open(os.environ["O"]+"/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","dtype")}); print("parity", json.dumps(d["parity"])[:900])
for k in ("bf16x3","f16x2"):
    e=d["extra_workloads"][k]; print(k, e["value"], e["ms_per_view"], json.dumps(e["roofline"])[:300]); print("   parity", json.dumps(e.get("parity"))[:900]); print("   vs main", e["vs_main_line_kernel_same_view"])
print("config1", json.dumps(d["extra_workloads"]["config1"].get("parity")))
P
