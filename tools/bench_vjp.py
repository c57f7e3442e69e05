"""Timing of the forward+input-gradient kernel (render_path_grad's per-pose work): one 400x400 image."""
import sys, json, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from neural_sim_nerf_amd import synthetic as S
from neural_sim_nerf_amd.engine import NsrModel
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 400
sd_c = S.synth_weights(0); sd_f = S.synth_weights(1000, fine_of=sd_c)
m = NsrModel(sd_c, sd_f, variant=int(sys.argv[3]) if len(sys.argv) > 3 else 0)
K = S.scaled_K(400.0 / H)
c2w = S.sweep_poses(1, 0)[0]
o, d = m.get_rays(H, W, K, c2w)
cot = torch.randn(H * W, 3, device=m.device)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    m.render_rays_vjp(o.reshape(-1, 3), d.reshape(-1, 3), S.YCBV_NEAR, S.YCBV_FAR, cot)
    ms = m.last_kernel_ms()
flop = H * W * (64 + 192 + 192) * S.FLOP_PER_POINT
print(json.dumps({"variant": m.variant, "mlp": m.mlp, "vjp_kernel_ms": ms, "rays": H * W, "tflops_algorithmic(fwd 256 + bwd 192 evals/ray)": flop / ms / 1e9,
                  "Mray-samples/s": H * W * 192 / ms / 1e3}))
