#!/usr/bin/env python
"""Compile libnsr for gfx950 with -Rpass-analysis=kernel-resource-usage and print one row per kernel
(registers, spills, scratch, occupancy, LDS).  Runs without a GPU.  usage: tools/resource_usage.py [filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "neural_sim_nerf_amd", "csrc")
def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    extra = sys.argv[2:]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
           "-Wno-unused-value", "-Rpass-analysis=kernel-resource-usage", "nsr_api.hip", "-o", "/tmp/libnsr_ru.so"] + extra
    p = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, text=True)
    rows, cur = [], None
    for line in p.stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|[A-Za-z ]+(?:\[[^\]]*\])?):\s+(\S+)", line)
        if not m:
            if "error" in line: print(line)
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    print("%-40s %5s %5s %5s %6s %6s %7s %4s %7s" % ("kernel", "SGPR", "VGPR", "AGPR", "sSpill", "vSpill", "scratch", "occ", "LDS"))
    for r in rows:
        if flt in r["name"]:
            print("%-40s %5s %5s %5s %6s %6s %7s %4s %7s" % (r["name"][-40:], r.get("TotalSGPRs"), r.get("VGPRs"), r.get("AGPRs"),
                  r.get("SGPRs Spill"), r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"),
                  r.get("LDS Size [bytes/block]")))
    return p.returncode
if __name__ == "__main__":
    sys.exit(main())
