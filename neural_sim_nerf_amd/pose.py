"""Pose sampling for the bilevel loop: the psi -> camera pose map of the reference, so that the
whole psi -> image -> d/dpsi chain can be driven from this package (and pinned end to end in the tests).

    LL = optimization/utils/load_LINEMOD_noscale.py, GU = optimization/utils/gumble.py
    pose_spherical_nograd  LL:89-94     sample_pose_nograd  LL:250-301 (numpy, fp64 draws, records the noise)
    pose_spherical         LL:62-71     sample_pose         LL:202-247 (torch fp32, differentiable w.r.t. the
                                                                        categorical probabilities, replays the noise)

Host-side torch/numpy: O(K) scalar work per outer epoch, not on the render path.  The only deliberate difference:
`sample_pose_nograd` takes an explicit `seed` instead of seeding numpy with datetime.now().second (LL:273)."""
import numpy as np
import torch

DEGREES = np.array([0, 45, 90, 135, 180, 225, 270, 315], dtype=np.float64) + 22.5     # bin centres (LL:217, LL:266)
RADIUS = 1.01                                                                           # LL:245, LL:293
_FLIP = [[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]]                       # LL:70, LL:93


def _chain(rot_theta, rot_phi, radius):
    trans = torch.eye(4, dtype=torch.float32)
    trans[2, 3] = radius
    return torch.tensor(_FLIP, dtype=torch.float32) @ (rot_theta @ (rot_phi @ trans))


def pose_spherical_nograd(theta, phi, radius):
    """LL:89-94: angles in degrees (python / numpy scalars), trigonometry in fp64, matrices and products in fp32."""
    th, ph = theta / 180. * np.pi, phi / 180. * np.pi
    rphi = torch.tensor([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0],
                         [0, 0, 0, 1]], dtype=torch.float64).float()
    rth = torch.tensor([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0],
                        [0, 0, 0, 1]], dtype=torch.float64).float()
    return _chain(rth, rphi, radius)


def pose_spherical(theta, phi, radius):
    """LL:62-71: theta, phi fp32 tensors (degrees); differentiable w.r.t. phi."""
    th, ph = theta / 180. * np.pi, phi / 180. * np.pi
    one, zero = torch.ones(()), torch.zeros(())
    c, s = torch.cos(ph).reshape(()), torch.sin(ph).reshape(())
    rphi = torch.stack([torch.stack([one, zero, zero, zero]), torch.stack([zero, c, -s, zero]),
                        torch.stack([zero, s, c, zero]), torch.stack([zero, zero, zero, one])])
    c, s = torch.cos(th).reshape(()), torch.sin(th).reshape(())
    rth = torch.stack([torch.stack([c, zero, -s, zero]), torch.stack([zero, one, zero, zero]),
                       torch.stack([s, zero, c, zero]), torch.stack([zero, zero, zero, one])])
    return _chain(rth, rphi, radius)


def sample_pose_nograd(categorical_prob, num_K, gumble_T, seed=0):
    """LL:250-301: Gumbel-softmax draw of the azimuth bin (GU:64-70), uniform jitter inside the 45-degree bin,
    elevation ~ U(85, 95), radius 1.01.  Returns (poses [K,4,4] fp32, sample_log) -- the log replays in sample_pose."""
    logits = np.log(np.asarray(categorical_prob))
    rng = np.random.RandomState(seed)
    gumbels, phis = [], []
    for _ in range(num_K):
        g = rng.gumbel(size=len(logits))
        z = np.exp((logits + g) / gumble_T)
        phis.append(np.sum(z / np.sum(z) * DEGREES))
        gumbels.append(g.tolist())
    uniforms = [rng.uniform(0, 1) for _ in range(num_K)]
    phis = [p - 22.5 + 45 * u for p, u in zip(phis, uniforms)]
    thetas = [rng.uniform(85, 95) for _ in range(num_K)]
    poses = torch.stack([pose_spherical_nograd(t, p - 180, RADIUS) for t, p in zip(thetas, phis)], 0)
    return poses, {"gumbel_noises": gumbels, "uniform_noises": uniforms, "thetas": thetas}


def sample_pose(categorical_prob, num_K, gumble_T, sample_log):
    """LL:202-247: the same poses as a differentiable function of `categorical_prob` (fp32 tensor), replaying the
    recorded noise (GU:57-63)."""
    logits = torch.log(categorical_prob)
    degrees = torch.tensor(DEGREES, dtype=torch.float32)
    poses = []
    for n in range(num_K):
        noise = torch.tensor(sample_log["gumbel_noises"][n], dtype=torch.float32)
        w = torch.softmax((logits + noise) / gumble_T, dim=0)
        phi = torch.sum(w * degrees) - 22.5 + 45 * sample_log["uniform_noises"][n]
        theta = torch.tensor([sample_log["thetas"][n]], dtype=torch.float32)
        poses.append(pose_spherical(theta, phi - 180, RADIUS))
    return torch.stack(poses, 0)
