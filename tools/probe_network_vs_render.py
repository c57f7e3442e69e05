"""Perf probes: run_network alone (no per-item phases) vs the fused render, per network pass."""
import sys, json, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from neural_sim_nerf_amd import synthetic as S
from neural_sim_nerf_amd.engine import NsrModel
sd_c = S.synth_weights(0); sd_f = S.synth_weights(1000, fine_of=sd_c)
m = NsrModel(sd_c, sd_f)
P = 160000 * 256 if len(sys.argv) < 2 else 1024
pts = torch.rand(P, 3, device=m.device) * 2 - 1
dirs = torch.nn.functional.normalize(torch.randn(P, 3, device=m.device), dim=-1)
for _ in range(2):
    torch.cuda.synchronize(); t = time.perf_counter(); m.run_network(pts, dirs, 0); torch.cuda.synchronize(); dt = time.perf_counter() - t
print(json.dumps({"run_network_ms": dt * 1e3, "tflops": P * S.FLOP_PER_POINT / dt / 1e12}))
c2w = torch.as_tensor(S.sweep_poses(1, 0)[0][:3, :4], device=m.device)
for _ in range(2):
    m.render_views(c2w, 400, 400, S.YCBV_K, S.YCBV_NEAR, S.YCBV_FAR); ms = m.last_kernel_ms()
print(json.dumps({"render_ms": ms, "tflops": 160000 * 256 * S.FLOP_PER_POINT / ms / 1e9}))
