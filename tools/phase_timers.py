"""Per-phase cycle totals of k_render from the NSR_PHASE_TIMING diagnostic build (thread 0 of every workgroup)."""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from neural_sim_nerf_amd import synthetic as S, _lib
from neural_sim_nerf_amd.engine import NsrModel, _dev, _stream_ptr
sd_c = S.synth_weights(0); sd_f = S.synth_weights(1000, fine_of=sd_c)
import os
V = int(os.environ.get('V', '32')); WG = int(os.environ.get('WG', '0'))
m = NsrModel(sd_c, sd_f, variant=V, max_workgroups=WG, mlp=os.environ.get('NSR_MLP', 'fp32'))   # V names the fp32 kernel; NSR_MLP the x32-structured bf16x3 / f16x2 ones
H = W = 400
c2w = torch.as_tensor(S.sweep_poses(1, 0)[0][:3, :4], device=m.device)[None].contiguous()
n = H * W
o, ro, _ = m._outs(n, False)
t = torch.zeros(512 * 12, dtype=torch.int64, device=m.device)
dbg = _lib.NsrDebugOut(None, None, _dev(t), None, None, None)
K9 = (C.c_double * 9)(*[float(S.YCBV_K[i][j]) for i in range(3) for j in range(3)])
for _ in range(2):
    _lib.check(m.lib.nsr_render_views(m.h, _dev(c2w), 1, H, W, K9, S.YCBV_NEAR, S.YCBV_FAR, C.byref(ro), C.byref(dbg), _stream_ptr(m.device)))
    ms = m.last_kernel_ms()
raw_t = t.cpu().numpy().astype(np.float64)
x32 = (V == 32 or m.mlp != "fp32") and m.schedule != "phases"      # the x32-structured kernels: one workgroup per CU,
G = torch.cuda.get_device_properties(m.device).multi_processor_count if x32 else 512   # [grid][8] totals + [grid][4] pass breakdown
tt = raw_t[:G * 8].reshape(G, 8)
tt = tt[tt.sum(1) > 0]
print('variant', V, 'mlp', m.mlp, 'workgroups', len(tt))
names = ["stage rays", "network passes", "coarse composite+out", "sample_pdf", "z_std+dbg", "merge sort", "fine composite+out", "-"]
if m.schedule == "phases":
    names = ["task start (wait, ray, z load)", "network passes", "coarse post-phase", "z publish", "fine post-phase", "queue pull", "-", "-"]
tot = tt.sum(1).mean()
print("kernel ms %.2f  total cycles/WG %.3e (100 MHz counter? ratio to ms: %.1f MHz)" % (ms, tot, tot / ms / 1e3))
for i, nm in enumerate(names):
    print("%-22s %6.2f %%   %.3e cycles/WG" % (nm, 100 * tt[:, i].mean() / tot, tt[:, i].mean()))

if x32:                                          # the x32-structured kernels also break the network passes down
    tp = raw_t[G * 8:G * 8 + G * 4].reshape(G, 4)
    tp = tp[tp.sum(1) > 0]
    ptot = tp.sum(1).mean()
    for i, nm in enumerate(["encodings", "GEMMs", "between GEMMs (epilogue, bias, first split)", "heads + output"]):
        print("  pass: %-44s %6.2f %% of the passes   %.3e cycles/WG" % (nm, 100 * tp[:, i].mean() / ptot, tp[:, i].mean()))
