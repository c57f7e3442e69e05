#!/bin/bash
# First-hardware-run kit for N > 1.  Nothing in this repository has ever run on more than one GPU: the
# gpurun boxes have one.  On the first node with several MI355X, run
#     bash tools/scale_sweep.sh [max_gpus=8] [out_dir=gpurun_out/scale]
# It runs bench.py at N = 1, 2, 4, 8 (up to the devices present) for the three workloads -- view400 (BASELINE configs[1] per
# rank, image all-gather), sweep100 (configs[2]: 100 views, view i -> rank i mod N) and models21 (configs[4]) -- with RCCL's
# own log (NCCL_DEBUG=INFO) captured per run, checks ranks_seen == N in every line, and prints the measured numbers NEXT TO
# what DESIGN.md section 6 expects: per-phase seconds of the sweep against its budget, and the N = 1 value of every workload
# against the single-GPU bench line.  It fabricates nothing: a run that fails is reported as failed.
R=$(cd "$(dirname "$0")/.." && pwd)
MAXG=${1:-8}
O=${2:-$R/gpurun_out/scale}
mkdir -p "$O"
export HSA_ENABLE_IPC_MODE_LEGACY=0
NDEV=$(python -c "import torch; print(torch.cuda.device_count())")
echo "devices visible: $NDEV"
for wl in view400 sweep100 models21; do
  for n in 1 2 4 8; do
    [ "$n" -gt "$NDEV" ] && continue
    [ "$n" -gt "$MAXG" ] && continue
    steps=3; [ "$wl" = view400 ] && steps=10
    echo "== $wl, $n GPU(s)"
    NCCL_DEBUG=INFO NCCL_DEBUG_FILE="$O/rccl_${wl}_n${n}_%h_%p.log" timeout 1200 \
      python "$R/bench.py" --gpus "$n" --steps "$steps" --warmup 1 --workload "$wl" --no-cpu-baseline --no-extras \
      > "$O/${wl}_n${n}.json" 2> "$O/${wl}_n${n}.err" || echo "   FAILED (rc $?): see $O/${wl}_n${n}.err"
  done
done
python - "$O" <<'PY'
import glob, json, os, sys
O = sys.argv[1]
BUDGET = {"render": 1.47, "png": 0.065, "gather_u8": 0.003, "gather_f32": 0.003}     # 13 views on the busiest of 8 ranks
rows = {}
for f in sorted(glob.glob(os.path.join(O, "*_n*.json"))):
    wl, n = os.path.basename(f)[:-5].rsplit("_n", 1)
    lines = [l for l in open(f) if l.startswith("{")]
    if not lines:
        print("%-9s N=%s: no bench line (failed)" % (wl, n)); continue
    j = json.loads(lines[-1])
    assert j["ranks_seen"] == int(n) == j["n_gpus"], "%s N=%s: ranks_seen %r" % (wl, n, j.get("ranks_seen"))
    rows.setdefault(wl, {})[int(n)] = j
for wl, by_n in rows.items():
    base = by_n.get(1)
    print("\n%s" % wl)
    for n, j in sorted(by_n.items()):
        sp = (j["value"] / base["value"]) if base else float("nan")
        ideal = j.get("ideal_speedup_over_1_gpu", n)
        print("  N=%d  %10.2f %s  ms/step %9.2f  x%.2f over N=1 (ideal %.2f)  kernel ms per rank min %.2f max %.2f  ranks_seen %d"
              % (n, j["value"], j["unit"], j["ms_per_step"], sp, ideal, j["kernel_ms_per_rank"]["min"],
                 j["kernel_ms_per_rank"]["max"], j["ranks_seen"]))
        ph = j.get("seconds_per_sweep_by_phase_max_over_ranks")
        if ph:
            k_max = j["config"]["views_on_busiest_rank"]
            print("        per-phase seconds (max over ranks): " + ", ".join(
                "%s %.4f (budget at 8 GPUs %.3f%s)" % (k, v, BUDGET.get(k, float("nan")), ", scaled to %d views: %.3f" % (k_max, BUDGET[k] * k_max / 13.0) if k in ("render", "png") else "")
                for k, v in ph.items()))
logs = glob.glob(os.path.join(O, "rccl_*.log"))
xgmi = sum(1 for f in logs for l in open(f, errors="replace") if "XGMI" in l.upper() or "P2P" in l.upper())
print("\nRCCL logs: %d files under %s, %d lines naming a P2P / xGMI transport" % (len(logs), O, xgmi))
single = os.path.join(os.path.dirname(O), "..", "profiles", "r04", "bench_final_kernel.json")
if os.path.exists(single) and "view400" in rows and 1 in rows["view400"]:
    b = json.loads(open(single).read())
    print("view400 at N=1: %.2f Mray-samples/s here, %.2f in profiles/r04/bench_final_kernel.json" % (rows["view400"][1]["value"], b["value"]))
PY
