"""debug: kw_gemm_f32 vs the oracle on a tiny network (run_network)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nerf_oracle as O
from neural_sim_nerf_amd.wide import WideModel
rng = np.random.RandomState(0)
for (D, W) in ((2, 32), (2, 64), (3, 256), (3, 384)):
    sd = O.synth_weights_shape(5, D, W, 4, 2, [], True)
    m = WideModel(sd, None, n_samples=8, n_importance=0, mlp="fp32")
    pts = (rng.rand(300, 3).astype(np.float32) - 0.5) * 0.4
    dirs = O.normalize_dirs(rng.standard_normal((300, 3)).astype(np.float32))
    want = O.run_network(sd, pts[:, None], dirs)[:, 0]
    got = m.run_network(pts, dirs, 0).cpu().numpy()
    print(D, W, "max abs err", np.abs(got - want).max(), "want max", np.abs(want).max(), got[0], want[0])
    m.close()
