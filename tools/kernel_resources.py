"""Register / spill / scratch figures of the kernels INSIDE a built library, read from the code object's metadata -- no
compiler run, no GPU: python tools/kernel_resources.py [neural_sim_nerf_amd/csrc/libnsr.so].

.hip_fatbin (llvm-objcopy) -> the gfx950 code object (clang-offload-bundler) -> its amdhsa metadata note (llvm-readelf).
`vgpr_count` is arch VGPRs + AGPRs (the unified file of gfx90a+).  tests/test_host_logic.py holds the shipped library to
these numbers: the fused kernels must not spill."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_resources(lib):
    """{demangled-ish kernel name: dict(vgpr_count, agpr_count, sgpr_count, vgpr_spill_count, sgpr_spill_count,
    private_segment_fixed_size, group_segment_fixed_size)} and the list of offload targets in the library."""
    MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
    notes, gpu = [], []
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "k.co")
        subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]         # one bundle per translation unit of the library
        for i, a in enumerate(starts):
            one = os.path.join(tmp, "fat%d.bin" % i)
            open(one, "wb").write(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            targets = subprocess.run([LLVM + "/clang-offload-bundler", "--list", "--type=o", "--input=" + one], check=True,
                                     capture_output=True, text=True).stdout.split()
            g = [t for t in targets if "amdgcn" in t]
            gpu += [t for t in g if t not in gpu]
            subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + one, "--targets=" + g[0],
                            "--output=" + co], check=True)
            notes.append(subprocess.run([LLVM + "/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout)
    out = {}
    for block in re.split(r"\n\s+- \.agpr_count:", "\n" + "\n".join(notes))[1:]:
        block = ".agpr_count:" + block
        get = lambda k: re.search(r"\.%s:\s+(\S+)" % k, block)
        name = get("name")
        if not name:
            continue
        m = re.match(r"_ZN3nsr\d+(\w+?)E", name.group(1))
        key = m.group(1) if m else name.group(1)
        if key.startswith("_ZN4nsrw"):              # the layered renderer's kernels (templates): kw_gemm<128, 1, 16> etc.
            mm = re.match(r"_ZN4nsrw(\d+)", key)
            base = key[len(mm.group(0)):][:int(mm.group(1))]
            targs = re.match(r"I((?:Li\d+E)+)E", key[len(mm.group(0)) + int(mm.group(1)):])
            key = base + ("<%s>" % ", ".join(re.findall(r"Li(\d+)E", targs.group(1))) if targs else "")
        out[key] = {k: int(get(k).group(1)) for k in (
            "agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
            "group_segment_fixed_size") if get(k)}
    return out, gpu


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "neural_sim_nerf_amd", "csrc", "libnsr.so")
    res, targets = kernel_resources(lib)
    print("offload targets:", ", ".join(targets))
    print("%-28s %5s %5s %5s %7s %7s %8s" % ("kernel", "vgpr", "agpr", "sgpr", "v-spill", "s-spill", "scratch"))
    for k in sorted(res):
        r = res[k]
        print("%-28s %5d %5d %5d %7d %7d %8d" % (k, r["vgpr_count"], r["agpr_count"], r["sgpr_count"], r["vgpr_spill_count"],
                                               r["sgpr_spill_count"], r["private_segment_fixed_size"]))
