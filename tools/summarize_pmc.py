"""gpurun_out/prof (tools/collect_profiles.sh) -> profiles/<round>/pmc_k_render.json + kernel stats + bench line.
    python tools/summarize_pmc.py r01"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles", sys.argv[1] if len(sys.argv) > 1 else "r01")
os.makedirs(dst, exist_ok=True)
out = {}
FLOP = {"x16": 160000 * 303824896, "x16q": 160000 * 303824896, "x32": 160000 * 303824896, "vjp": 160000 * 531693568,
        "vjpq": 160000 * 531693568, "b3": 160000 * 303824896, "vjpb3": 160000 * 531693568, "h2": 160000 * 303824896,
        "vjph2": 160000 * 531693568}
for tag, key, kname in (("h2", "f16x2", "k_render_h2"), ("vjph2", "vjp_f16x2", "k_render_vjp_h2"),
                        ("x16", "x16_phases_schedule", "k_render16p"), ("x16q", "x16_queue_schedule", "k_render16("),
                        ("x32", "x32", "k_render("), ("b3", "bf16x3", "k_render_b3"), ("vjp", "vjp", "k_render_vjp16p"),
                        ("vjpq", "vjp_queue_schedule", "k_render_vjp16("), ("vjpb3", "vjp_bf16x3", "k_render_vjp_b3")):
    tot, disp, ns, first_id = {}, {}, None, {}
    for d in sorted(glob.glob(os.path.join(src, "pmc_%s_*" % tag))):
        if not os.path.isdir(d):
            continue
        newest = lambda pat: sorted(glob.glob(os.path.join(d, "*", pat)), key=os.path.getmtime)[-1:]   # gpurun merges runs
        for f in newest("*_counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if kname in r["Kernel_Name"]:
                    first = first_id.setdefault(f, r["Dispatch_Id"])      # one launch per pass: the first dispatch of the kernel
                    if r["Dispatch_Id"] != first:
                        continue
                    tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                    disp = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size") if k in r}
        for f in newest("*_kernel_trace.csv"):
            for r in csv.DictReader(open(f)):
                if kname in r["Kernel_Name"] and ns is None:
                    ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if not tot:
        continue
    n_simd = 256 * 4
    der = {}
    if "GRBM_GUI_ACTIVE" in tot and ns:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs of the dispatch
        der["clock_GHz"] = tot["GRBM_GUI_ACTIVE"] / 8 / ns
        der["mfma_busy_frac"] = tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (n_simd * tot["GRBM_GUI_ACTIVE"] / 8)
    if "SQ_WAVE_CYCLES" in tot:
        der["wait_any_frac_of_wave_cycles"] = tot.get("SQ_WAIT_ANY", 0) / tot["SQ_WAVE_CYCLES"]
        der["wait_inst_frac"] = tot.get("SQ_WAIT_INST_ANY", 0) / tot["SQ_WAVE_CYCLES"]
    if "SQ_INSTS_VALU" in tot and "SQ_INSTS_MFMA" in tot:
        der["valu_insts_per_mfma_inst"] = (tot["SQ_INSTS_VALU"] - tot["SQ_INSTS_MFMA"]) / tot["SQ_INSTS_MFMA"]
    if "SQ_LDS_IDX_ACTIVE" in tot:
        der["lds_bank_conflict_frac_of_lds_active"] = tot.get("SQ_LDS_BANK_CONFLICT", 0) / tot["SQ_LDS_IDX_ACTIVE"]
    if "FETCH_SIZE" in tot:
        der["hbm_traffic_bytes_per_launch"] = (2 * tot["FETCH_SIZE"] + tot.get("WRITE_SIZE", 0)) * 1024
    if ns:
        der["algorithmic_TFLOPs_under_pmc"] = FLOP[tag] / ns / 1e3
    der["note"] = ("one 400x400x(64+128) view per launch; separate rocprofv3 --pmc passes with --kernel-trace only "
                   "(tools/collect_profiles.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md; counters summed over the XCDs")
    out[key] = {"counters": tot, "dispatch": disp, "kernel_ns_under_pmc": ns, "derived": der}
hf = os.path.join(src, "kernel_source_sha256.txt")
out["kernel_source_sha256"] = open(hf).read().strip() if os.path.exists(hf) else None
sched = {}
for name in ("queue", "phases", "bf16x3", "x32", "f16x2", "fp32"):
    f = os.path.join(src, "schedule_%s.log" % name)
    if os.path.exists(f):
        ms = [float(l.split()[-1]) for l in open(f) if l.startswith("variant")]
        if len(ms) > 1:
            sched[name] = {"kernel_ms_after_warmup": ms[1:], "mean_ms": sum(ms[1:]) / len(ms[1:])}
if sched:
    out["schedule_timing_unprofiled"] = sched
vj = {}
for name in ("phases", "queue", "x32", "bf16x3", "f16x2", "fp32"):
    f = os.path.join(src, "vjp_%s.log" % name)
    if os.path.exists(f):
        for l in open(f):
            if l.startswith("{"):
                vj[name] = json.loads(l)
if vj:
    out["vjp_timing_unprofiled"] = vj
pf = os.path.join(src, "probe_bf16x3.log")
if os.path.exists(pf):
    out["layer_gemm_probe"] = {"lines": [l.strip() for l in open(pf) if l.lstrip().startswith("mode")],
                               "modes": "2 = fp32 x32 production segment, 12 = bf16x3 MFMAs only, 11 = bf16x3 production groups (tools/probe_bf16x3.py)"}
json.dump(out, open(os.path.join(dst, "pmc_k_render.json"), "w"), indent=1)
for sub, name in (("stats", "kernel_stats_bench_steps3.csv"), ("stats_b3", "kernel_stats_bench_bf16x3.csv"), ("stats_vjp", "kernel_stats_vjp.csv"),
                  ("stats_f16x2", "kernel_stats_bench_steps3.csv"), ("stats_fp32", "kernel_stats_bench_fp32.csv"),
                  ("stats_bf16x3", "kernel_stats_bench_bf16x3.csv"), ("stats_vjp_f16x2", "kernel_stats_vjp.csv"),
                  ("stats_vjp_fp32", "kernel_stats_vjp_fp32.csv"), ("stats_vjp_bf16x3", "kernel_stats_vjp_bf16x3.csv"),
                  ("stats_handoff", "kernel_stats_handoff.csv")):
    for f in sorted(glob.glob(os.path.join(src, sub, "*", "*_kernel_stats.csv")), key=os.path.getmtime)[-1:]:
        rows = list(csv.reader(open(f)))          # keep our kernels + the top rows, drop torch's kilobyte-long template names
        with open(os.path.join(dst, name), "w", newline="") as g:
            w = csv.writer(g)
            for r in rows[:1] + [r for r in rows[1:] if r and (r[0].startswith("nsr::") or len(r[0]) < 120)]:
                w.writerow(r)
for sub, name in (("stats_layered_ycbv", "kernel_stats_layered_8x256_f16x2.csv"), ("stats_layered_w512", "kernel_stats_layered_8x512_f16x2.csv")):
    for f in sorted(glob.glob(os.path.join(src, sub, "*", "*_kernel_stats.csv")), key=os.path.getmtime)[-1:]:
        rows = list(csv.reader(open(f)))
        with open(os.path.join(dst, name), "w", newline="") as g:
            w = csv.writer(g)
            for r in rows[:1] + [r for r in rows[1:] if r and ("nsrw" in r[0] or "nsr::" in r[0] or len(r[0]) < 120)]:
                w.writerow(r)
os.makedirs(os.path.join(dst, "extra"), exist_ok=True)
for mlp in ("f16x2", "bf16x3", "fp32"):
    f = os.path.join(src, "layered_bench_%s.txt" % mlp)
    if os.path.exists(f) and os.path.getsize(f):
        shutil.copy(f, os.path.join(dst, "extra", "layered_bench_%s.txt" % mlp))
lp = {}
for tag in ("f16x2_w512", "f16x2_ycbv", "bf16x3_w512", "fp32_w512"):
    f = os.path.join(src, "layered_pmc_%s" % tag, "pmc_summary.json")
    if os.path.exists(f):
        try:
            lp[tag] = {k: v for k, v in json.load(open(f)).items() if "kw_" in k}
        except Exception as e:          # (a failed pass leaves a traceback there)
            lp[tag] = {"error": repr(e)}
if lp:
    json.dump({"note": "tools/pmc_wide.sh: separate rocprofv3 --pmc passes (counters + kernel trace only) over tools/bench_wide.py at 200x200, "
                       "one forward view; per kernel of the layered renderer the counters are summed over ALL its dispatches of the run",
               "runs": lp}, open(os.path.join(dst, "extra", "layered_pmc.json"), "w"), indent=1)
for extra in ("phase_timers.txt", "path_grad_f16x2.json", "importance_counts.txt", "api_overhead.json"):
    if os.path.exists(os.path.join(src, extra)):
        os.makedirs(os.path.join(dst, "extra"), exist_ok=True)
        shutil.copy(os.path.join(src, extra), os.path.join(dst, "extra", extra))
if os.path.exists(os.path.join(src, "bench.json")):
    lines = [l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")]
    if lines:
        open(os.path.join(dst, "bench_final_kernel.json"), "w").write(lines[-1])
print(json.dumps({k: v["derived"] for k, v in out.items() if isinstance(v, dict) and "derived" in v}, indent=1))
