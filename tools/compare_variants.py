"""x32 (1 workgroup/CU, 32 points/wave) vs x16 (2 workgroups/CU, 16 points/wave) forward kernels, one 400x400 view."""
import sys, json
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from neural_sim_nerf_amd import synthetic as S
from neural_sim_nerf_amd.engine import NsrModel
sd_c = S.synth_weights(0); sd_f = S.synth_weights(1000, fine_of=sd_c)
c2w = S.sweep_poses(1, 0)[0]
res = {}
models = {v: NsrModel(sd_c, sd_f, variant=v) for v in (32, 16)}
for rnd in range(3):
    for v, m in models.items():
        m.render_views(c2w, 400, 400, S.YCBV_K, S.YCBV_NEAR, S.YCBV_FAR)
        res.setdefault(v, []).append(m.last_kernel_ms())
for v in res:
    ms = min(res[v])
    print(json.dumps({"variant": v, "ms": ["%.2f" % x for x in res[v]], "tflops": 160000 * 256 * S.FLOP_PER_POINT / ms / 1e9}))
