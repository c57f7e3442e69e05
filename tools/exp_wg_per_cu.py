import sys, json
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from neural_sim_nerf_amd import synthetic as S
from neural_sim_nerf_amd.engine import NsrModel
sd_c = S.synth_weights(0); sd_f = S.synth_weights(1000, fine_of=sd_c)
c2w = S.sweep_poses(1, 0)[0]
for v, wg in [(16, 256), (16, 512), (16, 384), (32, 256)]:
    m = NsrModel(sd_c, sd_f, variant=v, max_workgroups=wg)
    ms = []
    for r in range(3):
        m.render_views(c2w, 400, 400, S.YCBV_K, S.YCBV_NEAR, S.YCBV_FAR); ms.append(m.last_kernel_ms())
    print(v, wg, ["%.2f" % x for x in ms], 160000 * 256 * S.FLOP_PER_POINT / min(ms) / 1e9)
