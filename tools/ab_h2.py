"""A/B of library builds on one 400x400 view (f16x2 by default): python tools/ab_h2.py [--mlp f16x2] [--n 8] lib1.so lib2.so ...

Each library is loaded in its own process (NSR_LIB_PATH); prints the median / minimum kernel time of n launches and a hash
of the rendered rgb so that builds that must agree bit for bit can be told apart from builds that do not."""
import os, subprocess, sys

CHILD = r"""
import sys, hashlib, statistics
sys.path.insert(0, %(root)r)
from neural_sim_nerf_amd import synthetic as S
from neural_sim_nerf_amd.engine import NsrModel
sd_c = S.synth_weights(0); sd_f = S.synth_weights(1000, fine_of=sd_c)
m = NsrModel(sd_c, sd_f, mlp=%(mlp)r, range_fallback=%(fb)r)
ms = []
for _ in range(%(n)d):
    out = m.render_views(S.sweep_poses(1, 0)[0], 400, 400, S.YCBV_K, S.YCBV_NEAR, S.YCBV_FAR)
    ms.append(m.last_kernel_ms())
rgb = out[0] if isinstance(out, (tuple, list)) else out["rgb_map"]
h = hashlib.sha256(rgb.detach().cpu().numpy().tobytes()).hexdigest()[:16]
print("%(tag)s: median %%.2f ms  min %%.2f ms  rgb %%s" %% (statistics.median(ms[1:]), min(ms[1:]), h))
"""

def main():
    args = sys.argv[1:]
    mlp, n = "f16x2", 8
    while args and args[0].startswith("--"):
        k = args.pop(0)
        if k == "--mlp": mlp = args.pop(0)
        elif k == "--n": n = int(args.pop(0))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for lib in args:
        env = dict(os.environ, NSR_LIB_PATH=os.path.abspath(lib))
        # (libraries older than r05 have no bf16x3 fallback images on f16x2 handles)
        code = CHILD % dict(root=root, mlp=mlp, n=n, tag=os.path.basename(lib), fb="fp32" if "r04" in os.path.basename(lib) else "bf16x3")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        print((r.stdout.strip() or "(no output)") + ("" if r.returncode == 0 else "\n  FAILED: " + r.stderr.strip()[-400:]))

if __name__ == "__main__":
    main()
