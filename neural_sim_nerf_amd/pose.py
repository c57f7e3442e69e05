"""Pose sampling for the bilevel loop: the psi -> camera pose map of the reference, so that the
whole psi -> image -> d/dpsi chain can be driven from this package (and pinned end to end in the tests).

    LL = optimization/utils/load_LINEMOD_noscale.py, GU = optimization/utils/gumble.py
    pose_spherical_nograd  LL:89-94     sample_pose_nograd  LL:250-301 (numpy, fp64 draws, records the noise)
    pose_spherical         LL:62-71     sample_pose         LL:202-247 (torch fp32, differentiable w.r.t. the
                                                                        categorical probabilities, replays the noise)

Two implementations of the same maps:
  * host (`sample_pose`, `sample_pose_nograd`): torch/numpy, bit-equal to the reference on its own host arithmetic;
  * device (`sample_pose_device`, `sample_pose_nograd_device`): csrc/nsr_pose.hip through the C ABI -- the poses are
    written into device memory where nsr_render_views reads them, and the d(pose)/d(psi) Jacobian comes out of the same
    kernel, so render_path_grad needs no host autograd graph.  The RANDOM DRAWS stay on the host in both (numpy's
    Mersenne Twister, as the reference; O(K) numbers), recorded in `sample_log` and uploaded.
The only deliberate difference from the reference: `sample_pose_nograd*` take an explicit `seed` instead of seeding
numpy with datetime.now().second (LL:273)."""
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


# ----------------------------------------------------------------------------------------------------------
# device versions (csrc/nsr_pose.hip)
# ----------------------------------------------------------------------------------------------------------
def _util(device=None):
    from .run_nerf_noscale import _util_model
    return _util_model(device)


def draw_noise(categorical_prob, num_K, seed=0):
    """The reference's random draws (LL:273-292) in its order: K Gumbel vectors, then K uniforms, then K thetas."""
    n_cat = len(categorical_prob)
    rng = np.random.RandomState(seed)
    gumbels = [rng.gumbel(size=n_cat).tolist() for _ in range(num_K)]
    uniforms = [rng.uniform(0, 1) for _ in range(num_K)]
    thetas = [rng.uniform(85, 95) for _ in range(num_K)]
    return {"gumbel_noises": gumbels, "uniform_noises": uniforms, "thetas": thetas}


def sample_pose_nograd_device(categorical_prob, num_K, gumble_T, seed=0, device=None):
    """LL:250-301 with the deterministic part on the device: returns (poses [K,4,4] fp32 DEVICE tensor, sample_log).
    `categorical_prob` is what NM:88 passes: a numpy array (float16 there); np.log keeps its dtype as in LL:268."""
    log = draw_noise(categorical_prob, num_K, seed)
    logits = np.log(np.asarray(categorical_prob)).astype(np.float64)
    poses = _util(device).sample_pose_nograd(logits, log["gumbel_noises"], log["uniform_noises"], log["thetas"], gumble_T,
                                             RADIUS)
    return poses, log


class _SamplePose(torch.autograd.Function):
    """poses = f(categorical_prob) with the kernel's own Jacobian as the backward (rank one per pose: only the azimuth
    depends on psi).  vmap-able, so torch.autograd.grad(..., is_grads_batched=True) (render_path_grad) works on it."""
    generate_vmap_rule = True

    @staticmethod
    def forward(prob, jac, poses):
        return poses.clone()

    @staticmethod
    def setup_context(ctx, inputs, output):
        ctx.save_for_backward(inputs[1])

    @staticmethod
    def backward(ctx, g):
        (jac,) = ctx.saved_tensors                             # [K,12,n_cat]
        g12 = g[:, :3, :4].reshape(g.shape[0], 12, 1).to(jac.dtype)
        return (g12 * jac).sum((0, 1)), None, None


def sample_pose_device(categorical_prob, num_K, gumble_T, sample_log, device=None):
    """LL:202-247 on the device: poses [K,4,4] fp32 DEVICE tensor, differentiable w.r.t. `categorical_prob` (a CPU or
    device fp32 tensor); the gradient reaches it on its own device.  The Jacobian is also attached as `poses.nsr_jac`
    ([K,12,n_cat], device) for callers that want it without autograd."""
    m = _util(device)
    prob = categorical_prob.detach()
    poses, jac = m.sample_pose(prob, sample_log["gumbel_noises"][:num_K], sample_log["uniform_noises"][:num_K],
                               sample_log["thetas"][:num_K], gumble_T, RADIUS)
    if categorical_prob.requires_grad:
        jac_p = jac if categorical_prob.is_cuda else jac.cpu()
        out = _SamplePose.apply(categorical_prob, jac_p, poses if categorical_prob.is_cuda else poses.cpu())
        out = out if categorical_prob.is_cuda else _ToDevice.apply(out, m.device)
    else:
        out = poses
    out.nsr_jac = jac
    return out


class _ToDevice(torch.autograd.Function):
    """Identity that moves the pose tensor to the render device and its gradient back (psi may live on the host)."""
    generate_vmap_rule = True

    @staticmethod
    def forward(x, device):
        return x.to(device)

    @staticmethod
    def setup_context(ctx, inputs, output):
        ctx.src = inputs[0].device

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.src), None
