"""Synthetic inputs for benchmarks and smoke runs (there is no network for the pretrained YCB-V checkpoints the
reference downloads, README.md:48-50): seeded weights with the reference's state_dict names, the YCB-V object-2
camera (logs/nerfdata/nerf_traindata_info.json:2-5,35-51, near/far widened by 0.5 as LL:197-198 does) and a
seeded pose sweep on the radius-1.01 shell the reference samples from (LL:292-293)."""
import numpy as np

YCBV_K = [[1333.3333740234375, 0.0, 195.4293212890625],
          [0.0, 1334.2196044921875, 200.63180541992188],
          [0.0, 0.0, 1.0]]
YCBV_NEAR = 0.8103964843749999 - 0.5
YCBV_FAR = 1.4297681884765627 + 0.5
YCBV_HW = 400

_LAYERS = ([("pts_linears.0", 256, 63)]
           + [("pts_linears.%d" % i, 256, 256 + (63 if i == 5 else 0)) for i in range(1, 8)]
           + [("feature_linear", 256, 256), ("alpha_linear", 1, 256), ("views_linears.0", 128, 283),
              ("rgb_linear", 3, 128)])
FLOP_PER_POINT = 2 * sum(o * i for _, o, i in _LAYERS)     # 1 186 816


def scaled_K(scale):
    K = [list(r) for r in YCBV_K]
    K[0] = [v / scale for v in K[0]]
    K[1] = [v / scale for v in K[1]]
    return K


def synth_weights(seed, fine_of=None):
    """nn.Linear-style U(+-1/sqrt(in)) init; trunk x1.6, alpha weight x50, alpha bias -0.5 so that rays cross
    both empty and opaque space (default init is degenerate: acc ~ 0).  `fine_of` derives a
    distinct-but-consistent fine network from a coarse one."""
    rng = np.random.RandomState(seed)
    sd = {}
    if fine_of is not None:
        for k, v in fine_of.items():
            sd[k] = (v * (1.0 + 0.05 * rng.standard_normal(v.shape))).astype(np.float32)
        return sd
    for name, o, i in _LAYERS:
        bound = 1.0 / np.sqrt(i)
        w = rng.uniform(-bound, bound, size=(o, i))
        b = rng.uniform(-bound, bound, size=(o,))
        if name.startswith("pts_linears"):
            w = w * 1.6
        if name == "alpha_linear":
            w = w * 50.0
            b = b * 0.0 - 0.5
        sd[name + ".weight"] = w.astype(np.float32)
        sd[name + ".bias"] = b.astype(np.float32)
    return sd


def pose_spherical(theta_deg, phi_deg, radius):
    """camera-to-world on a sphere, same construction as LL:89-94 (fp32 4x4)."""
    th, ph = theta_deg / 180.0 * np.pi, phi_deg / 180.0 * np.pi
    trans = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]], np.float32)
    rphi = np.array([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0],
                     [0, 0, 0, 1]], np.float32)
    rth = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0],
                    [0, 0, 0, 1]], np.float32)
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float32)
    return (flip @ (rth @ (rphi @ trans))).astype(np.float32)


def sweep_poses(n_views, seed=0):
    rng = np.random.RandomState(seed)
    return np.stack([pose_spherical(rng.uniform(85, 95), rng.uniform(0, 360) - 180.0, 1.01)
                     for _ in range(n_views)], 0)
