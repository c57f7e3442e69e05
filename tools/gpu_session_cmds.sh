# scratch file: tools/gpu_session.sh <label> executes it on the gpurun box (r06 session AM: the trained-network census on the FULL 400x400 view)
cd $GRAFT_REPO_ROOT
timeout 1500 python bench.py --steps 3 --warmup 1 --trained-side 400 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
t = d["extra_workloads"]["trained"]
print(t["workload"]); print(t["network"])
for k, v in t.items():
    if isinstance(v, dict) and "rays" in v: print(k, {x: v[x] for x in ("rays", "rays_above_tol", "unattributed", "cliff_rays", "index_flip_rays", "denom_switch_rays", "illconditioned_shift_rays", "psnr_delta_db", "max_abs_rgb", "inds_exact", "passes")}, v["range_status"])
json.dump(t, open("$O/trained_full_view.json", "w"), indent=1)
print(d["value"], t["passes"])
PY
