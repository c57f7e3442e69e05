"""g26: a TRAINED pair of NeRFs, produced by the reference's own code.

Every other fixture of tests/golden uses one synthetic weight family (nn.Linear-uniform init, scaled).  The reference's
consumers load trained checkpoints (RN:296-314): sparse, saturated densities, long empty stretches, opaque rays whose
resampling denominators sit at the 1e-5 switch (RH:238-239).  The 21 YCB-V checkpoints are not in the repository and there is
no network, so this script TRAINS two networks here, with the reference's modules and functions only:

    RH.NeRF (RH:70-122) x 2, torch's default initialisation, torch.optim.Adam as RN:287 builds it,
    RN.render(rays=batch) with render_kwargs_train (perturb = 1, RN:318-330), loss = img2mse(rgb) + img2mse(rgb0) (RN:696-703),
    the exponential learning-rate schedule of RN:711-715,

against an ANALYTIC scene (no dataset): a textured box of the YCB-V cracker box's proportions at the origin, black
background, seen from the radius-1.01 shell of LL:89-94 with the YCB-V object-2 camera (nerf_traindata_info.json).

    python oracle/train_g26.py train   [steps]     # checkpoints -> /tmp/g26/ckpt_<step>.npz   (CPU, about 1.3 s per step here)
    python oracle/train_g26.py fixture <ckpt.npz>  # the reference's render of the trained pair -> tests/golden/g26_trained.npz
    python oracle/train_g26.py check               # re-renders from the weights stored IN the fixture: must reproduce bit for bit

The fixture holds the two state dicts and, from the REFERENCE's render() at 64 + 128 samples: a 40x40 view with everything
oracle/census.py takes, and autograd's d rgb / d rays on 512 rays.  `check` needs only the fixture and /root/reference, so
the stored OUTPUTS are reproducible from the stored weights whether or not the training run itself is (multi-threaded sgemm).

TEST INFRASTRUCTURE ONLY.  Runs only in the authoring container (needs /root/reference, read-only); nothing of the
reference travels -- inputs, weights it trained and its outputs only.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import nerf_oracle as O  # noqa: E402  (camera constants only)
import gen_golden as G  # noqa: E402  (import_reference, Capture, save)

WORK = os.environ.get("G26_WORK", "/tmp/g26")
HALF = np.array([0.030, 0.079, 0.105], np.float32)           # half extents of the box, metres (6 x 15.8 x 21 cm)
BASE = np.array([[0.90, 0.15, 0.10], [0.95, 0.75, 0.10], [0.15, 0.35, 0.85],
                 [0.10, 0.70, 0.30], [0.85, 0.85, 0.85], [0.55, 0.15, 0.65]], np.float32)   # a colour per face


def scene(o, d):
    """The analytic target: colour seen along rays o + t d, [N,3] each -> [N,3] in [0,1].  Slab test against the box; the
    texture is the face's colour times a 2.5 cm checker plus a smooth sinusoid (so that both edges and gradients exist)."""
    o = o.astype(np.float64); d = d.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        t0 = (-HALF - o) / d
        t1 = (HALF - o) / d
    tn = np.minimum(t0, t1).max(-1)
    tf = np.maximum(t0, t1).min(-1)
    hit = (tn < tf) & (tn > 0)
    p = o + d * np.where(hit, tn, 0.0)[:, None]
    q = np.abs(p) / HALF
    ax = q.argmax(-1)
    sign = np.take_along_axis(p, ax[:, None], -1)[:, 0] > 0
    face = ax * 2 + sign
    uv = np.stack([np.take_along_axis(p, ((ax + 1) % 3)[:, None], -1)[:, 0],
                   np.take_along_axis(p, ((ax + 2) % 3)[:, None], -1)[:, 0]], -1)
    checker = (np.floor(uv[:, 0] / 0.025) + np.floor(uv[:, 1] / 0.025)) % 2
    wave = 0.5 + 0.5 * np.sin(55.0 * uv[:, 0] + 1.3 * face) * np.cos(38.0 * uv[:, 1])
    col = BASE[face] * (0.45 + 0.40 * checker + 0.15 * wave)[:, None]
    return np.where(hit[:, None], np.clip(col, 0.0, 1.0), 0.0).astype(np.float32)


def make_nets(RN, RH):
    embed_fn, _ = RH.get_embedder(10, 0)
    embeddirs_fn, _ = RH.get_embedder(4, 0)
    nets = [RH.NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True) for _ in range(2)]
    query = lambda inputs, viewdirs, fn: RN.run_network(inputs, viewdirs, fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                                                        netchunk=1 << 20)
    kw = dict(network_query_fn=query, N_samples=64, network_fn=nets[0], network_fine=nets[1], use_viewdirs=True,
              white_bkgd=False, ndc=False, lindisp=False, near=O.YCBV_NEAR, far=O.YCBV_FAR)
    return nets, kw


def train_poses(LL, n, seed):
    rng = np.random.RandomState(seed)
    return [LL.pose_spherical_nograd(float(rng.uniform(55, 125)), float(rng.uniform(-180, 180)), 1.01) for _ in range(n)]


def train(steps):
    os.makedirs(WORK, exist_ok=True)
    torch.set_num_threads(int(os.environ.get("G26_THREADS", "5")))
    RN, RH, LL = G.import_reference()
    torch.manual_seed(26)
    nets, kw = make_nets(RN, RH)
    kw_train = dict(kw, perturb=1.0, raw_noise_std=0.0, N_importance=int(os.environ.get("G26_NI", "64")))
    Kq = O.scaled_K(4.0)                                   # 100 x 100 training views of the 400 x 400 camera
    ros, rds = [], []
    for c2w in train_poses(LL, 80, 2600):
        ro, rd = RH.get_rays(100, 100, Kq, c2w[:3, :4])
        ros.append(ro.reshape(-1, 3).numpy()); rds.append(rd.reshape(-1, 3).numpy())
    ro = np.concatenate(ros); rd = np.concatenate(rds)
    target = scene(ro, rd)
    print("pool: %d rays, %.1f %% on the box" % (len(ro), 100.0 * (target.max(-1) > 0).mean()), flush=True)
    ro, rd, target = torch.from_numpy(ro), torch.from_numpy(rd), torch.from_numpy(target)
    params = list(nets[0].parameters()) + list(nets[1].parameters())
    lrate = float(os.environ.get("G26_LR", "5e-4"))
    opt = torch.optim.Adam(params=params, lr=lrate, betas=(0.9, 0.999))         # RN:287
    n_rand = int(os.environ.get("G26_BATCH", "512"))
    gen = torch.Generator().manual_seed(2626)
    start = 0
    resume = os.environ.get("G26_RESUME")
    if resume:
        ck = torch.load(resume)
        for net, sd in zip(nets, ck["nets"]):
            net.load_state_dict(sd)
        opt.load_state_dict(ck["opt"])
        start = ck["step"]
        gen.set_state(ck["gen"])
        torch.set_rng_state(ck["rng"])
    t0 = time.time()
    for step in range(start + 1, steps + 1):
        sel = torch.randint(0, ro.shape[0], (n_rand,), generator=gen)
        batch = torch.stack([ro[sel], rd[sel]], 0)
        rgb, disp, acc, extras = RN.render(100, 100, Kq, chunk=1 << 20, rays=batch, **kw_train)
        opt.zero_grad()
        loss = RH.img2mse(rgb, target[sel])
        psnr = RH.mse2psnr(loss.detach())
        loss = loss + RH.img2mse(extras["rgb0"], target[sel])                   # RN:701-703
        loss.backward()
        opt.step()
        new_lr = lrate * (0.1 ** (step / 250000.0))                             # RN:711-715 (lrate_decay = 250)
        for g in opt.param_groups:
            g["lr"] = new_lr
        if step % 25 == 0 or step == 1:
            print("step %5d  loss %.5f  psnr %.2f  acc mean %.3f  %.2f s/step" % (step, loss.item(), psnr.item(), acc.mean().item(),
                                                                               (time.time() - t0) / (step - start)), flush=True)
        if step % 250 == 0 or step == steps:
            np.savez(os.path.join(WORK, "ckpt_%05d.npz" % step),
                     **{"c." + k: v.detach().numpy() for k, v in nets[0].state_dict().items()},
                     **{"f." + k: v.detach().numpy() for k, v in nets[1].state_dict().items()}, step=np.int64(step))
            torch.save({"nets": [n.state_dict() for n in nets], "opt": opt.state_dict(), "step": step, "gen": gen.get_state(),
                        "rng": torch.get_rng_state()}, os.path.join(WORK, "resume.pt"))


def load_pair(npz):
    sd_c = {k[2:]: np.asarray(npz[k], np.float32) for k in npz.files if k.startswith("c.")}
    sd_f = {k[2:]: np.asarray(npz[k], np.float32) for k in npz.files if k.startswith("f.")}
    return sd_c, sd_f


def _sigma_stats(s):
    """[max, 99.9 %, share > 100, share <= 0] of the fine pass's densities: what makes the network a TRAINED one"""
    return np.array([s.max(), np.percentile(s, 99.9), (s > 100).mean(), (s <= 0).mean()], np.float64)


VIEW = dict(theta=88.0, phi=35.0)          # sees three faces of the box
NG = 512                                   # rays of the gradient leg


def reference_outputs(sd_c, sd_f):
    """The reference's render() of the pair at 64 + 128 samples, perturb = 0: a 40 x 40 view with what the census takes, and
    autograd's d rgb / d rays on NG rays of a second view."""
    RN, RH, LL = G.import_reference()
    nets, kw = make_nets(RN, RH)
    for net, sd in zip(nets, (sd_c, sd_f)):
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    kw_test = dict(kw, perturb=False, raw_noise_std=0.0, N_importance=128)
    sig_last = []
    orig_r2o = RN.raw2outputs

    def r2o(raw, *a, **k):
        sig_last.append(raw[..., -1, 3].detach().numpy().copy())
        return orig_r2o(raw, *a, **k)
    RN.raw2outputs = r2o
    try:
        K40 = O.scaled_K(10.0)
        c2w = LL.pose_spherical_nograd(VIEW["theta"], VIEW["phi"], 1.01)
        with G.Capture(RN, RH) as cap:
            with torch.no_grad():
                rgb, disp, acc, ex = RN.render(40, 40, K40, chunk=512, c2w=c2w[:3, :4], retraw=True, **kw_test)
        catc = lambda k: np.concatenate([c[k] for c in cap.log], 0)
        s0 = np.concatenate(sig_last[0::2]); s1 = np.concatenate(sig_last[1::2])
        del sig_last[:]
        out = dict(c2w=c2w.numpy(), K40=np.array(K40), rgb=rgb.numpy().reshape(-1, 3), disp=disp.numpy().ravel(),
                   acc=acc.numpy().ravel(), rgb0=ex["rgb0"].numpy().reshape(-1, 3), disp0=ex["disp0"].numpy().ravel(),
                   acc0=ex["acc0"].numpy().ravel(), z_std=ex["z_std"].numpy().ravel(), pdf_weights=catc("weights"),
                   inds=catc("inds").astype(np.int8), z_samples=catc("samples"), sigma0_last=s0, sigma_last=s1,
                   sigma_max=ex["raw"].numpy().reshape(1600, 192, 4)[..., 3].max(-1),
                   sigma_stats=_sigma_stats(ex["raw"].numpy().reshape(1600, 192, 4)[..., 3]))
        # the gradient leg: NG rays of another view, a seeded cotangent, the reference's own autograd (RN:168-178)
        c2w2 = LL.pose_spherical_nograd(97.0, -140.0, 1.01)
        ro, rd = RH.get_rays(40, 40, K40, c2w2[:3, :4])
        sel = np.random.RandomState(260).choice(1600, NG, replace=False)
        rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0).clone().requires_grad_()
        cot = torch.from_numpy(np.random.RandomState(261).standard_normal((NG, 3)).astype(np.float32))
        with G.Capture(RN, RH) as cap:
            rgb_g, _, _, _ = RN.render(40, 40, K40, chunk=128, rays=rays, **kw_test)
        (g,) = torch.autograd.grad(rgb_g, rays, grad_outputs=cot)
        out.update(grad_c2w=c2w2.numpy(), grad_sel=sel.astype(np.int32), grad_rays_in=rays.detach().numpy(), cot=cot.numpy(),
                   grad_rgb=rgb_g.detach().numpy(), grad_rays=g.numpy(),
                   grad_z_samples=np.concatenate([c["samples"] for c in cap.log], 0))
    finally:
        RN.raw2outputs = orig_r2o
    return out


def fixture(ckpt):
    npz = np.load(ckpt)
    sd_c, sd_f = load_pair(npz)
    out = reference_outputs(sd_c, sd_f)
    out["target"] = scene(*[a.reshape(-1, 3) for a in O.get_rays(40, 40, O.scaled_K(10.0), out["c2w"][:3, :4])])
    out["train_steps"] = np.int64(npz["step"])
    G.save("g26_trained", **{"c." + k: v for k, v in sd_c.items()}, **{"f." + k: v for k, v in sd_f.items()}, **out)
    print("sigma (fine, 40x40 view): max %.1f, 99.9 %% %.1f, share > 100: %.4f, share <= 0: %.3f; acc in (0.01, 0.99): %.3f of the rays;"
          " PSNR against the analytic scene %.2f dB" % (tuple(out["sigma_stats"]) + (((out["acc"] > 0.01) & (out["acc"] < 0.99)).mean(),
                                                                                     O.psnr(out["rgb"], out["target"]))))


def check():
    path = os.path.join(G.OUT, "g26_trained.npz")
    g = np.load(path)
    out = reference_outputs(*load_pair(g))
    bad = [k for k, v in out.items() if not np.array_equal(np.asarray(v), g[k], equal_nan=True)]
    print("g26_trained: %d keys re-rendered from the stored weights, %d differ %s" % (len(out), len(bad), bad))
    return 1 if bad else 0


if __name__ == "__main__":
    if os.environ.get("GOLDEN_OUT"):              # e.g. GOLDEN_OUT=/tmp/regen: write / check somewhere else
        G.OUT = os.environ["GOLDEN_OUT"]
        os.makedirs(G.OUT, exist_ok=True)
    mode = sys.argv[1] if len(sys.argv) > 1 else "train"
    if mode == "train":
        train(int(sys.argv[2]) if len(sys.argv) > 2 else 3000)
    elif mode == "fixture":
        fixture(sys.argv[2])
    else:
        sys.exit(check())
