"""Hand-off throughput: to8b + find_bbox for 100 rendered-size views on the GPU, against the
HBM roofline (these are byte kernels).  GPU only: the host route of the reference (PNG encode + decode + the oracle's
restatement of cv2 grey / threshold / connected components) is timed beside it by bench.py
(`extra_workloads.handoff.cpu_baseline`), the only place outside tests/ that may touch oracle/.  Prints one JSON object."""
import json, os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from neural_sim_nerf_amd import png
from neural_sim_nerf_amd.run_nerf_noscale import _util_model

K, H, W = 100, 400, 400
m = _util_model()
rng = np.random.RandomState(0)
yy, xx = np.mgrid[:H, :W]
rgb = np.zeros((K, H, W, 3), np.float32)
for i in range(K):                                   # an object-like blob on black, plus a few specks
    cy, cx, r = rng.randint(120, 280), rng.randint(120, 280), rng.randint(40, 110)
    sel = (yy - cy) ** 2 + ((xx - cx) * rng.uniform(0.6, 1.4)) ** 2 <= r * r
    rgb[i][sel] = rng.uniform(0.05, 1.0, (sel.sum(), 3))
    for _ in range(5):
        rgb[i][rng.randint(0, H), rng.randint(0, W)] = 0.5
x = torch.as_tensor(rgb, device=m.device)

def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

img8 = m.to8b(x)
ms_to8b = timed(lambda: m.to8b(x))
ms_bbox = timed(lambda: m.find_bbox(img8, with_mask=True))
bbox, count, mask = m.find_bbox(img8, with_mask=True)
n = K * H * W
out = {"views": K, "H": H, "W": W,
       "to8b": {"ms": round(ms_to8b, 4), "algorithmic_bytes": n * 3 * 5, "GBps": round(n * 3 * 5 / ms_to8b / 1e6, 1),
                "frac_of_8TBps": round(n * 3 * 5 / ms_to8b / 1e6 / 8000, 4)},
       "find_bbox": {"ms": round(ms_bbox, 4), "algorithmic_bytes": n * 4, "GBps": round(n * 4 / ms_bbox / 1e6, 1),
                     "note": "5 kernels per batch of 16 images; 24 B/pixel of L2-resident scratch (union-find parents + per-root statistics)"},
       "views_per_s_gpu": round(K / ((ms_to8b + ms_bbox) * 1e-3), 1)}
print(json.dumps(out))
