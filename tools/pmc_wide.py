"""Summary of tools/pmc_wide.sh: per kernel of the layered renderer, counters summed over ALL its dispatches of the run and
the derived figures (clock = GRBM_GUI_ACTIVE / 8 XCDs / time; MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles);
HBM-side traffic = (2 FETCH_SIZE + WRITE_SIZE) KiB, MI355X_MICROARCH.md).    python tools/pmc_wide.py <dir>"""
import csv, glob, json, os, sys
d0 = sys.argv[1]
out = {}
for d in sorted(glob.glob(os.path.join(d0, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    ns = {}
    for f in glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            ns[k] = ns.get(k, 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            e = out.setdefault(k, {"counters": {}, "ns": {}})
            e["counters"][r["Counter_Name"]] = e["counters"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            e["vgpr"] = r.get("VGPR_Count")
            e["lds"] = r.get("LDS_Block_Size")
    for k, v in ns.items():
        if k in out:
            out[k]["ns"][os.path.basename(d)] = v
res = {}
for k, e in out.items():
    c = e["counters"]
    t = e["ns"].get("pmc_1")
    der = {"kernel_ms_under_pmc": {p: round(v / 1e6, 3) for p, v in sorted(e["ns"].items())}, "vgpr": e.get("vgpr"), "lds": e.get("lds")}
    if t and "GRBM_GUI_ACTIVE" in c:
        der["clock_GHz"] = round(c["GRBM_GUI_ACTIVE"] / 8 / t, 3)
        der["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8), 3)
    if "SQ_WAVE_CYCLES" in c:
        der["wait_any_frac_of_wave_cycles"] = round(c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], 3)
        der["wait_inst_frac"] = round(c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], 3)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        der["lds_bank_conflict_frac_of_lds_active"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"], 4)
    if "FETCH_SIZE" in c:
        der["hbm_side_traffic_GB"] = round((2 * c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0)) * 1024 / 1e9, 2)
        tt = e["ns"].get("pmc_4")
        if tt:
            der["fetch_TBps"] = round(2 * c["FETCH_SIZE"] * 1024 / tt / 1e3, 2)
    res[k] = {"derived": der, "counters": c}
print(json.dumps(res, indent=1))
