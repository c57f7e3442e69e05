"""Generate the golden vectors under tests/golden/ by IMPORTING AND RUNNING the reference.

Runs only in the authoring container (needs /root/reference, read-only).  Nothing of the reference
travels: this script imports it in place, feeds it seeded inputs and stores inputs -> outputs as small
.npz fixtures.  Shims: stub modules for `imageio`/`cv2` (only used for PNG I/O) and a
no-op `Tensor.cuda` (the reference hard-codes .cuda() calls; there is no GPU here).

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz

TEST INFRASTRUCTURE ONLY.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, HERE)
import nerf_oracle as O  # noqa: E402  (only for the synthetic-weight recipe and the camera constants)

REF = "/root/reference/optimization"


def import_reference():
    for m in ("imageio", "cv2"):
        sys.modules.setdefault(m, types.ModuleType(m))
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    import utils.run_nerf_noscale as RN
    import utils.run_nerf_helpers as RH
    import utils.load_LINEMOD_noscale as LL
    return RN, RH, LL


def build_nets(RN, RH, seed):
    """Reference NeRF modules (RH:70) loaded with the oracle's synthetic weights."""
    sd_c = O.synth_weights(seed)
    sd_f = O.synth_weights(seed + 1000, fine_of=sd_c)
    nets = []
    for sd in (sd_c, sd_f):
        net = RH.NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        nets.append(net)
    embed_fn, _ = RH.get_embedder(10, 0)
    embeddirs_fn, _ = RH.get_embedder(4, 0)
    query = lambda inputs, viewdirs, fn: RN.run_network(inputs, viewdirs, fn, embed_fn=embed_fn,
                                                        embeddirs_fn=embeddirs_fn, netchunk=65536)
    kwargs = dict(network_query_fn=query, perturb=False, N_importance=128, network_fine=nets[1],
                  N_samples=64, network_fn=nets[0], use_viewdirs=True, white_bkgd=False,
                  raw_noise_std=0.0, ndc=False, lindisp=False,
                  near=O.YCBV_NEAR, far=O.YCBV_FAR)
    return nets, kwargs, embed_fn, embeddirs_fn


class Capture:
    """Records what the reference's sample_pdf saw and produced (bins, weights, cdf, inds, samples)."""

    def __init__(self, RN, RH):
        self.RN, self.RH = RN, RH
        self.log = []

    def __enter__(self):
        self.orig_pdf = self.RN.sample_pdf
        self.orig_ss = torch.searchsorted
        cap = self

        def ss(cdf, u, right=False, **kw):
            out = cap.orig_ss(cdf, u, right=right, **kw)
            cap.cur.update(cdf=cdf.detach().numpy().copy(), inds=out.numpy().copy(),
                           u=u.detach().numpy().copy())
            return out

        def pdf(bins, weights, N_samples, det=False, pytest=False):
            cap.cur = dict(bins=bins.detach().numpy().copy(), weights=weights.detach().numpy().copy())
            torch.searchsorted = ss
            try:
                s = cap.orig_pdf(bins, weights, N_samples, det=det, pytest=pytest)
            finally:
                torch.searchsorted = cap.orig_ss
            cap.cur["samples"] = s.detach().numpy().copy()
            cap.log.append(cap.cur)
            return s

        self.RN.sample_pdf = pdf
        return self

    def __exit__(self, *a):
        self.RN.sample_pdf = self.orig_pdf
        torch.searchsorted = self.orig_ss


ONLY = os.environ.get("GOLDEN_ONLY")          # e.g. GOLDEN_ONLY=g11_options: leave the other fixtures untouched


def save(name, **arrs):
    if ONLY and name != ONLY:
        return
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024))


def main():
    os.makedirs(OUT, exist_ok=True)
    RN, RH, LL = import_reference()
    torch.manual_seed(0)
    rng = np.random.RandomState(1234)
    SEED = 7
    nets, kwargs, embed_fn, embeddirs_fn = build_nets(RN, RH, SEED)

    # the census (oracle/census.py) needs the sigma of the LAST sample of both passes (the 1e10 interval, RN:358-359):
    # raw2outputs is wrapped to note the raw outputs it is given (coarse call, then fine call, per chunk)
    sig_last = []
    orig_r2o = RN.raw2outputs

    def r2o(raw, *a, **k):
        sig_last.append(raw[..., -1, 3].detach().numpy().copy())
        return orig_r2o(raw, *a, **k)
    RN.raw2outputs = r2o

    def take_sig(fine=True):
        """(coarse, fine) sigma_last of the render calls since the last take, concatenated over chunks"""
        c = np.concatenate(sig_last[0::2] if fine else sig_last)
        f = np.concatenate(sig_last[1::2]) if fine else None
        del sig_last[:]
        return c, f

    # ---- G9 poses (LL:89-94) ----------------------------------------------------------------
    angles = np.array([[90.0, 30.0 - 180.0], [85.5, 200.0 - 180.0], [94.2, 311.0 - 180.0], [88.0, -170.0]])
    poses = np.stack([LL.pose_spherical_nograd(t, p, 1.01).numpy() for t, p in angles])
    save("g9_pose", angles=angles, radius=np.float64(1.01), c2w=poses)
    c2w = torch.from_numpy(poses[0])

    # ---- G1 get_rays (RH:156-165) -----------------------------------------------------------
    K8 = O.scaled_K(50.0)
    o8, d8 = RH.get_rays(8, 8, K8, c2w[:3, :4])
    o400, d400 = RH.get_rays(400, 400, O.YCBV_K, c2w[:3, :4])
    pix = np.array([[0, 0], [0, 399], [399, 0], [399, 399], [200, 195], [17, 301], [250, 3]])
    save("g1_get_rays", c2w=poses[0], K8=np.array(K8), o8=o8.numpy(), d8=d8.numpy(),
         K400=np.array(O.YCBV_K), pix=pix, o400=o400.numpy()[pix[:, 0], pix[:, 1]],
         d400=d400.numpy()[pix[:, 0], pix[:, 1]])

    # ---- G2 embedding (RH:18-66) ------------------------------------------------------------
    pts = rng.uniform(-2, 2, size=(512, 3)).astype(np.float32)
    dirs = rng.standard_normal((128, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    save("g2_embed", pts=pts, e_pts=embed_fn(torch.from_numpy(pts)).numpy(),
         dirs=dirs, e_dirs=embeddirs_fn(torch.from_numpy(dirs)).numpy())

    # ---- G3 MLP (RH:99-122) -----------------------------------------------------------------
    x = np.concatenate([embed_fn(torch.from_numpy(pts[:256])).numpy(),
                        embeddirs_fn(torch.from_numpy(dirs[np.arange(256) % 128])).numpy()], -1)
    hid = {}
    hooks = [nets[0].pts_linears[i].register_forward_hook(
        lambda m, a, out, i=i: hid.__setitem__("h%d" % i, torch.relu(out).detach().numpy().copy()))
        for i in (0, 4, 7)]
    with torch.no_grad():
        y_c = nets[0](torch.from_numpy(x)).numpy()
        for h in hooks:
            h.remove()
        y_f = nets[1](torch.from_numpy(x)).numpy()
    save("g3_mlp", seed=np.int64(SEED), x=x, y_coarse=y_c, y_fine=y_f, **hid)

    # ---- G4 raw2outputs (RN:343-387) --------------------------------------------------------
    def analytic_raw(n, s):
        zc = np.linspace(0, 1, s)[None, :]
        mu = rng.uniform(0.2, 0.8, size=(n, 1))
        sg = rng.uniform(0.02, 0.2, size=(n, 1))
        amp = rng.uniform(5, 400, size=(n, 1))
        sigma = amp * np.exp(-0.5 * ((zc - mu) / sg) ** 2) - 2.0
        raw = np.concatenate([rng.standard_normal((n, s, 3)) * 2, sigma[..., None]], -1).astype(np.float32)
        raw[0, :, 3] = -1.0          # empty ray: acc = 0 -> disp = NaN (RN:381)
        raw[1, :, 3] = 1e4           # saturated ray
        raw[2, :, 3] = 0.0
        raw[2, -1, 3] = 1e-9         # only the 1e10 last interval contributes
        return raw
    g4 = {}
    for s in (64, 192):
        n = 48
        raw = analytic_raw(n, s)
        z = np.sort(rng.uniform(O.YCBV_NEAR, O.YCBV_FAR, size=(n, s)).astype(np.float32), -1)
        rd = rng.standard_normal((n, 3)).astype(np.float32)
        with torch.no_grad():
            outs = RN.raw2outputs(torch.from_numpy(raw), torch.from_numpy(z), torch.from_numpy(rd), 0, False)
        for nm, v in zip(("rgb", "disp", "acc", "weights", "depth"), outs):
            g4["%s_%d" % (nm, s)] = v.numpy()
        g4.update({"raw_%d" % s: raw, "z_%d" % s: z, "rays_d_%d" % s: rd})
    save("g4_raw2outputs", **g4)

    # ---- G5 sample_pdf (RH:199-243) ---------------------------------------------------------
    n = 96
    z = O.coarse_z(np.full(n, O.YCBV_NEAR, np.float32), np.full(n, O.YCBV_FAR, np.float32))
    bins = (0.5 * (z[:, 1:] + z[:, :-1])).astype(np.float32)
    w = rng.uniform(0, 1, size=(n, 62)).astype(np.float32) ** 4
    w[0] = 0.0                                   # all-zero weights
    w[1] = 0.25                                  # uniform
    w[2] = 0.0; w[2, 30] = 0.9                   # single spike
    w[3] = 0.0; w[3, 5] = 0.4; w[3, 50] = 0.5    # two spikes
    w[4] = 0.0; w[4, 0] = 1.0                    # mass in the first bin
    w[5] = 0.0; w[5, 61] = 1.0                   # mass in the last bin
    w[6:40] *= 1e-4                              # tiny weights: the +1e-5 dominates
    with Capture(RN, RH) as cap:
        with torch.no_grad():
            RN.sample_pdf(torch.from_numpy(bins), torch.from_numpy(w), 128, det=True)
    c = cap.log[0]
    save("g5_sample_pdf", bins=bins, weights=w, cdf=c["cdf"], inds=c["inds"], samples=c["samples"], u=c["u"][0])

    # ---- G6 render_rays end to end (RN:390-501) ---------------------------------------------
    o32, d32 = RH.get_rays(400, 400, O.YCBV_K, c2w[:3, :4])
    sel = rng.choice(160000, size=192, replace=False)
    ro = o32.reshape(-1, 3)[sel]
    rd = d32.reshape(-1, 3)[sel]
    del sig_last[:]
    with Capture(RN, RH) as cap:
        with torch.no_grad():
            rgb, disp, acc, ex = RN.render(400, 400, O.YCBV_K, chunk=64, rays=torch.stack([ro, rd], 0),
                                           retraw=True, **kwargs)
    cat = lambda k: np.concatenate([c[k] for c in cap.log], 0)
    s0_6, s1_6 = take_sig()
    save("g6_render_rays", sigma0_last=s0_6, sigma_last=s1_6, seed=np.int64(SEED), rays_o=ro.numpy(), rays_d=rd.numpy(),
         near=np.float64(O.YCBV_NEAR), far=np.float64(O.YCBV_FAR),
         rgb=rgb.numpy(), disp=disp.numpy(), acc=acc.numpy(), raw=ex["raw"].numpy(),
         rgb0=ex["rgb0"].numpy(), disp0=ex["disp0"].numpy(), acc0=ex["acc0"].numpy(), z_std=ex["z_std"].numpy(),
         pdf_bins=cat("bins"), pdf_weights=cat("weights"), cdf=cat("cdf"), inds=cat("inds"),
         z_samples=cat("samples"))

    # ---- G7 full-image render (RN:58-123) ---------------------------------------------------
    K64 = O.scaled_K(6.25)
    kw0 = dict(kwargs, N_importance=0)
    with torch.no_grad():
        rgb, disp, acc, _ = RN.render(64, 64, K64, chunk=512, c2w=c2w[:3, :4], **kw0)
    g7 = dict(seed=np.int64(SEED), c2w=poses[0], K64=np.array(K64), rgb_c1=rgb.numpy(), disp_c1=disp.numpy(),
              acc_c1=acc.numpy())
    K32 = O.scaled_K(12.5)
    with torch.no_grad():
        rgb, disp, acc, ex = RN.render(32, 32, K32, chunk=512, c2w=torch.from_numpy(poses[2])[:3, :4], **kwargs)
    g7.update(c2w_b=poses[2], K32=np.array(K32), rgb_c2=rgb.numpy(), disp_c2=disp.numpy(), acc_c2=acc.numpy(),
              rgb0_c2=ex["rgb0"].numpy(), disp0_c2=ex["disp0"].numpy(), acc0_c2=ex["acc0"].numpy(),
              z_std_c2=ex["z_std"].numpy())
    save("g7_render", **g7)

    # ---- G8 backward: d(rgb)/d(rays) VJP (RN:168-178) -----------------------------------------
    nb = 96
    sel = rng.choice(160000, size=nb, replace=False)
    rays = torch.stack([o32.reshape(-1, 3)[sel], d32.reshape(-1, 3)[sel]], 0).clone().requires_grad_(True)
    cot = torch.from_numpy(rng.standard_normal((nb, 3)).astype(np.float32))
    with Capture(RN, RH) as cap:
        rgb_p, _, _, _ = RN.render(400, 400, O.YCBV_K, chunk=nb, rays=rays, retraw=True, **kwargs)
    (g,) = torch.autograd.grad(rgb_p, rays, grad_outputs=cot)
    save("g8_backward", seed=np.int64(SEED), rays=rays.detach().numpy(), cot=cot.numpy(),
         rgb=rgb_p.detach().numpy(), grad_rays=g.numpy(), z_samples=cap.log[0]["samples"])

    # ---- G10 the bilevel gradient end to end: psi -> poses (LL:202-301) -> render_path_grad (RN:126-210) -------
    class _FixedSecond:                       # LL:273 seeds numpy with datetime.now().second
        @staticmethod
        def now():
            return types.SimpleNamespace(second=5)
    LL.datetime = _FixedSecond
    psi = torch.tensor([0.02, 0.02, 0.02, 0.86, 0.02, 0.02, 0.02, 0.02])
    prob16 = np.array(torch.softmax(psi / 0.25, 0), dtype=np.float16)               # NM:85-87
    poses_ng, log = LL.sample_pose_nograd(prob16, 2, 0.1)
    prob = torch.softmax(psi / 0.25, 0).requires_grad_()                             # NM:141-142
    poses_g = LL.sample_pose(prob, 2, 0.1, log)
    Hs = 8
    Ks = O.scaled_K(50.0)
    gE = [{"grad_E": [torch.from_numpy(rng.standard_normal((3, Hs, Hs)).astype(np.float32))]} for _ in range(2)]
    kw10 = {k: v for k, v in kwargs.items()}
    rgbs10, dl = RN.render_path_grad(prob, poses_g, [Hs, Hs, Ks[0][0]], Ks, 16, gE, kw10, savedir=None)
    save("g10_path_grad", psi=psi.numpy(), prob16=prob16, gumbel=np.array(log["gumbel_noises"]),
         uniform=np.array(log["uniform_noises"]), thetas=np.array(log["thetas"]), poses_nograd=poses_ng.numpy(),
         poses_grad=poses_g.detach().numpy(), K=np.array(Ks), grad_E=np.stack([g["grad_E"][0].numpy() for g in gE]),
         rgbs=rgbs10, dLdpsis=np.stack([d.numpy() for d in dl]), seed=np.int64(SEED))

    # ---- G11 render options: white_bkgd (RN:384-385) and lindisp (RN:443), forward and VJP ------
    kw11 = dict(kwargs, white_bkgd=True, lindisp=True)
    sel = rng.choice(160000, size=160, replace=False)
    ro11 = o32.reshape(-1, 3)[sel]
    rd11 = d32.reshape(-1, 3)[sel]
    del sig_last[:]
    with Capture(RN, RH) as cap:
        with torch.no_grad():
            rgb, disp, acc, ex = RN.render(400, 400, O.YCBV_K, chunk=160, rays=torch.stack([ro11, rd11], 0),
                                           retraw=True, **kw11)
    s0_11, s1_11 = take_sig()
    fwd11 = dict(sigma0_last=s0_11, sigma_last=s1_11, pdf_weights=cap.log[0]["weights"], rays_o=ro11.numpy(), rays_d=rd11.numpy(), rgb=rgb.numpy(), disp=disp.numpy(), acc=acc.numpy(),
                 rgb0=ex["rgb0"].numpy(), disp0=ex["disp0"].numpy(), acc0=ex["acc0"].numpy(),
                 z_std=ex["z_std"].numpy(), raw=ex["raw"].numpy(), inds=cap.log[0]["inds"],
                 z_samples=cap.log[0]["samples"])
    rays = torch.stack([ro11[:64], rd11[:64]], 0).clone().requires_grad_(True)
    cot = torch.from_numpy(rng.standard_normal((64, 3)).astype(np.float32))
    with Capture(RN, RH) as cap:
        rgb_p, _, _, _ = RN.render(400, 400, O.YCBV_K, chunk=64, rays=rays, retraw=True, **kw11)
    (g,) = torch.autograd.grad(rgb_p, rays, grad_outputs=cot)
    save("g11_options", seed=np.int64(SEED), cot=cot.numpy(), vjp_rgb=rgb_p.detach().numpy(), grad_rays=g.numpy(),
         vjp_z_samples=cap.log[0]["samples"], **fwd11)

    # ---- G13 what the cliff / flip census (oracle/census.py) needs from the reference itself: a config-2-shaped view
    # (40x40, 64+128, the reference's chunk 512, CF:25) with the coarse weights its sample_pdf saw, its indices and
    # samples, and the sigma of the LAST sample of both passes (the 1e10 interval of RN:358-359); and the same sigma
    # for the config-1 view of g7 (64x64, coarse only).
    del sig_last[:]
    K40 = O.scaled_K(10.0)
    with Capture(RN, RH) as cap:
        with torch.no_grad():
            rgb, disp, acc, ex = RN.render(40, 40, K40, chunk=512, c2w=torch.from_numpy(poses[1])[:3, :4], **kwargs)
    catc = lambda k: np.concatenate([c[k] for c in cap.log], 0)
    s0, s1 = take_sig()
    g13 = dict(seed=np.int64(SEED), c2w=poses[1], K40=np.array(K40), rgb=rgb.numpy().reshape(-1, 3),
               disp=disp.numpy().ravel(), acc=acc.numpy().ravel(), rgb0=ex["rgb0"].numpy().reshape(-1, 3),
               disp0=ex["disp0"].numpy().ravel(), acc0=ex["acc0"].numpy().ravel(), z_std=ex["z_std"].numpy().ravel(),
               pdf_weights=catc("weights"), inds=catc("inds").astype(np.int8), z_samples=catc("samples"),
               sigma0_last=s0, sigma_last=s1)
    with torch.no_grad():
        rgb, disp, acc, _ = RN.render(64, 64, K64, chunk=512, c2w=c2w[:3, :4], **kw0)
    assert np.array_equal(rgb.numpy(), g7["rgb_c1"])             # the very render g7 holds
    g13.update(c1_sigma0_last=take_sig(fine=False)[0])
    save("g13_census", **g13)

    # ---- G14 the stochastic options and the other ray set-ups of render() -----------------------
    # (a) perturb > 0 (RN:447-459), sample_pdf with det=False (RH:211) and raw_noise_std > 0 (RN:365-374): the reference
    #     draws from torch's global generator; torch.rand / torch.randn are wrapped to RECORD what it drew, in call order
    #     (one chunk: t_rand [N,64], randn [N,64], u [N,128], randn [N,192]).  Forward and the gradient w.r.t. the rays.
    # (b) c2w_staticcam (RN:91-96): rays of one camera, view directions of another.
    # (c) ndc=True (RN:101-103, ndc_rays RH:168-186) on a forward-facing camera, near=0, far=1; forward and gradient.
    drawn = []
    orig_rand, orig_randn = torch.rand, torch.randn

    def rec(fn):
        def w(*a, **k):
            out = fn(*a, **k)
            drawn.append(out.detach().numpy().copy())
            return out
        return w
    NOISE_STD = 0.7
    sel = rng.choice(160000, size=96, replace=False)
    ro14 = o32.reshape(-1, 3)[sel]
    rd14 = d32.reshape(-1, 3)[sel]
    kw14 = dict(kwargs, perturb=1.0, raw_noise_std=NOISE_STD)
    rays = torch.stack([ro14, rd14], 0).clone().requires_grad_(True)
    cot14 = torch.from_numpy(rng.standard_normal((96, 3)).astype(np.float32))
    del sig_last[:]
    z_seen = []
    orig_r2o_b = RN.raw2outputs

    def r2o_z(raw, z_vals, *a, **k):
        z_seen.append(z_vals.detach().numpy().copy())
        return orig_r2o_b(raw, z_vals, *a, **k)
    RN.raw2outputs = r2o_z
    torch.rand, torch.randn = rec(orig_rand), rec(orig_randn)
    try:
        with Capture(RN, RH) as cap:
            rgb, disp, acc, ex = RN.render(400, 400, O.YCBV_K, chunk=96, rays=rays, retraw=True, **kw14)
    finally:
        torch.rand, torch.randn = orig_rand, orig_randn
        RN.raw2outputs = orig_r2o_b
    (g14g,) = torch.autograd.grad(rgb, rays, grad_outputs=cot14)
    assert [d.shape for d in drawn] == [(96, 64), (96, 64), (96, 128), (96, 192)], [d.shape for d in drawn]
    s0_14, s1_14 = take_sig()
    g14 = dict(seed=np.int64(SEED), noise_std=np.float64(NOISE_STD), rays_o=ro14.numpy(), rays_d=rd14.numpy(),
               t_rand=drawn[0], randn0=drawn[1], u=drawn[2], randn1=drawn[3], z_coarse=z_seen[0], z_fine=z_seen[1],
               rgb=rgb.detach().numpy(), disp=disp.detach().numpy(), acc=acc.detach().numpy(),
               rgb0=ex["rgb0"].detach().numpy(), disp0=ex["disp0"].detach().numpy(), acc0=ex["acc0"].detach().numpy(),
               z_std=ex["z_std"].detach().numpy(), raw16=ex["raw"].detach().numpy()[:16], inds=cap.log[0]["inds"].astype(np.int8),
               z_samples=cap.log[0]["samples"], pdf_weights=cap.log[0]["weights"], sigma0_last=s0_14, sigma_last=s1_14,
               cot=cot14.numpy(), grad_rays=g14g.numpy())
    # (b)
    K16 = O.scaled_K(25.0)
    with torch.no_grad():
        rgb, disp, acc, ex = RN.render(16, 16, K16, chunk=512, c2w=torch.from_numpy(poses[0])[:3, :4],
                                       c2w_staticcam=torch.from_numpy(poses[2])[:3, :4], **kwargs)
    g14.update(sc_K=np.array(K16), sc_c2w=poses[0], sc_c2w_static=poses[2], sc_rgb=rgb.numpy(), sc_disp=disp.numpy(),
               sc_acc=acc.numpy(), sc_rgb0=ex["rgb0"].numpy(), sc_z_std=ex["z_std"].numpy())
    # (c) a forward-facing camera at the z = 0 plane (the LLFF set-up NDC is made for), looking down -z, slightly rotated
    ang = 0.12
    c2w_ndc = np.array([[np.cos(ang), 0.0, np.sin(ang), 0.15], [0.0, 1.0, 0.0, -0.1], [-np.sin(ang), 0.0, np.cos(ang), 0.05],
                        [0.0, 0.0, 0.0, 1.0]], np.float32)
    Hn, Wn, fn_ = 12, 16, 20.0
    Kn = [[fn_, 0.0, Wn / 2.0], [0.0, fn_, Hn / 2.0], [0.0, 0.0, 1.0]]
    ro_n, rd_n = RH.get_rays(Hn, Wn, Kn, torch.from_numpy(c2w_ndc)[:3, :4])
    o_ndc, d_ndc = RH.ndc_rays(Hn, Wn, Kn[0][0], 1., ro_n, rd_n)
    rays_n = torch.stack([ro_n.reshape(-1, 3), rd_n.reshape(-1, 3)], 0).clone().requires_grad_(True)
    cot_n = torch.from_numpy(rng.standard_normal((Hn * Wn, 3)).astype(np.float32))
    kwn = dict(kwargs, near=0.0, far=1.0, ndc=True)
    with Capture(RN, RH) as cap:
        rgb, disp, acc, ex = RN.render(Hn, Wn, Kn, chunk=512, rays=rays_n, **kwn)
    (gn,) = torch.autograd.grad(rgb, rays_n, grad_outputs=cot_n)
    g14.update(ndc_H=np.int64(Hn), ndc_W=np.int64(Wn), ndc_K=np.array(Kn), ndc_c2w=c2w_ndc, ndc_rays_o=ro_n.numpy(),
               ndc_rays_d=rd_n.numpy(), ndc_o=o_ndc.numpy(), ndc_d=d_ndc.numpy(), ndc_rgb=rgb.detach().numpy(),
               ndc_disp=disp.detach().numpy(), ndc_acc=acc.detach().numpy(), ndc_rgb0=ex["rgb0"].detach().numpy(),
               ndc_z_std=ex["z_std"].detach().numpy(), ndc_z_samples=cap.log[0]["samples"], ndc_cot=cot_n.numpy(),
               ndc_grad_rays=gn.numpy())
    # (d) near / far as per-ray arrays (RN:106-108 multiplies them into [N,1])
    nb = (O.YCBV_NEAR + rng.uniform(0.0, 0.3, (96, 1))).astype(np.float32)
    fb = (O.YCBV_FAR - rng.uniform(0.0, 0.3, (96, 1))).astype(np.float32)
    with torch.no_grad():
        with Capture(RN, RH) as cap:
            rgb, disp, acc, ex = RN.render(400, 400, O.YCBV_K, chunk=96, rays=torch.stack([ro14, rd14], 0),
                                           **dict(kwargs, near=torch.from_numpy(nb), far=torch.from_numpy(fb)))
    g14.update(nf_near=nb[:, 0], nf_far=fb[:, 0], nf_rgb=rgb.numpy(), nf_disp=disp.numpy(), nf_acc=acc.numpy(),
               nf_rgb0=ex["rgb0"].numpy(), nf_z_std=ex["z_std"].numpy(), nf_z_samples=cap.log[0]["samples"])
    del sig_last[:]
    save("g14_stochastic", **g14)

    # ---- G15 the use_viewdirs=False network (RH:95-96, RH:119-120; create_nerf RN:262-267: output_ch = 5) ------------
    sd_nc = O.synth_weights_noviews(SEED)
    sd_nf = O.synth_weights_noviews(SEED + 1000, fine_of=sd_nc)
    nv = []
    for sd in (sd_nc, sd_nf):
        net = RH.NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=0, use_viewdirs=False)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        nv.append(net)
    q_nv = lambda inputs, viewdirs, fn: RN.run_network(inputs, viewdirs, fn, embed_fn=embed_fn, embeddirs_fn=None,
                                                      netchunk=65536)
    kw15 = dict(kwargs, network_query_fn=q_nv, network_fn=nv[0], network_fine=nv[1], use_viewdirs=False)
    sel = rng.choice(160000, size=64, replace=False)
    ro15, rd15 = o32.reshape(-1, 3)[sel], d32.reshape(-1, 3)[sel]
    rays = torch.stack([ro15, rd15], 0).clone().requires_grad_(True)
    cot15 = torch.from_numpy(rng.standard_normal((64, 3)).astype(np.float32))
    with Capture(RN, RH) as cap:
        rgb, disp, acc, ex = RN.render(400, 400, O.YCBV_K, chunk=64, rays=rays, retraw=True, **kw15)
    (g15g,) = torch.autograd.grad(rgb, rays, grad_outputs=cot15)
    del sig_last[:]
    save("g15_noviewdirs", seed=np.int64(SEED), rays_o=ro15.numpy(), rays_d=rd15.numpy(), rgb=rgb.detach().numpy(),
         disp=disp.detach().numpy(), acc=acc.detach().numpy(), rgb0=ex["rgb0"].detach().numpy(),
         acc0=ex["acc0"].detach().numpy(), z_std=ex["z_std"].detach().numpy(), raw16=ex["raw"].detach().numpy()[:16],
         z_samples=cap.log[0]["samples"], cot=cot15.numpy(), grad_rays=g15g.numpy())

    # ---- G16 networks of other shapes (create_nerf reads netdepth / netwidth / multires / multires_views, RN:260-278) ----
    g16 = dict(seed=np.int64(SEED))
    sel = rng.choice(160000, size=48, replace=False)
    ro16, rd16 = o32.reshape(-1, 3)[sel], d32.reshape(-1, 3)[sel]
    cot16 = torch.from_numpy(rng.standard_normal((48, 3)).astype(np.float32))
    pts16 = rng.uniform(-1.5, 1.5, (128, 3)).astype(np.float32)
    dir16 = rng.standard_normal((128, 3)).astype(np.float32)
    dir16 /= np.linalg.norm(dir16, axis=-1, keepdims=True)
    g16.update(rays_o=ro16.numpy(), rays_d=rd16.numpy(), cot=cot16.numpy(), pts=pts16, dirs=dir16)
    for tag, (D, Wd, L, Lv, skips, uv) in (("a", (6, 128, 6, 2, [2], True)), ("b", (4, 64, 10, 4, [4], False))):
        sdc = O.synth_weights_shape(SEED + 31, D, Wd, L, Lv, skips, uv)
        sdf = {k: (v * (1.0 + 0.05 * np.random.RandomState(SEED + 32).standard_normal(v.shape))).astype(np.float32)
               for k, v in sdc.items()}
        ef, in_ch = RH.get_embedder(L, 0)
        edf, in_v = RH.get_embedder(Lv, 0) if uv else (None, 0)
        nn_ = []
        for sd in (sdc, sdf):
            net = RH.NeRF(D=D, W=Wd, input_ch=in_ch, output_ch=5, skips=skips, input_ch_views=in_v, use_viewdirs=uv)
            net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
            nn_.append(net)
        q = lambda inputs, viewdirs, fn, ef=ef, edf=edf: RN.run_network(inputs, viewdirs, fn, embed_fn=ef, embeddirs_fn=edf,
                                                                        netchunk=65536)
        kw16 = dict(kwargs, network_query_fn=q, network_fn=nn_[0], network_fine=nn_[1], use_viewdirs=uv)
        with torch.no_grad():
            net_out = q(torch.from_numpy(pts16)[:, None, :], torch.from_numpy(dir16) if uv else None, nn_[0])[:, 0]
        rays = torch.stack([ro16, rd16], 0).clone().requires_grad_(True)
        with Capture(RN, RH) as cap:
            rgb, disp, acc, ex = RN.render(400, 400, O.YCBV_K, chunk=48, rays=rays, **kw16)
        (gg,) = torch.autograd.grad(rgb, rays, grad_outputs=cot16)
        g16.update({tag + "_shape": np.array([D, Wd, L, Lv, skips[0], int(uv)]), tag + "_net_out": net_out.numpy(),
                    tag + "_rgb": rgb.detach().numpy(), tag + "_acc": acc.detach().numpy(),
                    tag + "_disp": disp.detach().numpy(), tag + "_rgb0": ex["rgb0"].detach().numpy(),
                    tag + "_acc0": ex["acc0"].detach().numpy(), tag + "_z_samples": cap.log[0]["samples"],
                    tag + "_grad_rays": gg.numpy()})
    del sig_last[:]
    save("g16_other_shapes", **g16)

    # ---- G17 fewer importance samples (N_importance = 64, NM:1260 is an argument) --------------------------------
    sel = rng.choice(160000, size=48, replace=False)
    ro17, rd17 = o32.reshape(-1, 3)[sel], d32.reshape(-1, 3)[sel]
    kw17 = dict(kwargs, N_importance=64)
    rays = torch.stack([ro17, rd17], 0).clone().requires_grad_(True)
    cot17 = torch.from_numpy(rng.standard_normal((48, 3)).astype(np.float32))
    with Capture(RN, RH) as cap:
        rgb, disp, acc, ex = RN.render(400, 400, O.YCBV_K, chunk=48, rays=rays, **kw17)
    (g17g,) = torch.autograd.grad(rgb, rays, grad_outputs=cot17)
    del sig_last[:]
    save("g17_importance64", seed=np.int64(SEED), rays_o=ro17.numpy(), rays_d=rd17.numpy(), rgb=rgb.detach().numpy(),
         disp=disp.detach().numpy(), acc=acc.detach().numpy(), rgb0=ex["rgb0"].detach().numpy(),
         z_std=ex["z_std"].detach().numpy(), z_samples=cap.log[0]["samples"], inds=cap.log[0]["inds"].astype(np.int8),
         pdf_weights=cap.log[0]["weights"], cot=cot17.numpy(), grad_rays=g17g.numpy())

    # ---- G18 the reference's pytest hook (RN:454-457, RH:214-222): numpy's global generator reseeded with 0 at every
    # draw site and once per chunk; 80 rays in chunks of 32; stratified (perturb=1) and deterministic (perturb=0) ---------
    sel = rng.choice(160000, size=80, replace=False)
    ro18, rd18 = o32.reshape(-1, 3)[sel], d32.reshape(-1, 3)[sel]
    g18 = dict(seed=np.int64(SEED), rays_o=ro18.numpy(), rays_d=rd18.numpy(), chunk=np.int64(32))
    state = np.random.get_state()
    for tag, pert in (("p", 1.0), ("d", 0.0)):
        with torch.no_grad():
            with Capture(RN, RH) as cap:
                rgb, disp, acc, ex = RN.render(400, 400, O.YCBV_K, chunk=32, rays=torch.stack([ro18, rd18], 0),
                                               **dict(kwargs, perturb=pert, pytest=True))
        g18.update({tag + "_rgb": rgb.numpy(), tag + "_acc": acc.numpy(), tag + "_rgb0": ex["rgb0"].numpy(),
                    tag + "_z_std": ex["z_std"].numpy(), tag + "_u": np.concatenate([c["u"] for c in cap.log], 0),
                    tag + "_z_samples": np.concatenate([c["samples"] for c in cap.log], 0)})
    np.random.set_state(state)
    del sig_last[:]
    save("g18_pytest_hook", **g18)

    # ---- G19 c2w_staticcam WITHOUT view directions: RN:91-96 sits inside `if use_viewdirs:`, so the reference ignores the
    # static camera there and renders c2w's own rays ----------------------------------------------------------
    K16 = O.scaled_K(25.0)
    with torch.no_grad():
        rgb, disp, acc, ex = RN.render(16, 16, K16, chunk=512, c2w=torch.from_numpy(poses[0])[:3, :4],
                                       c2w_staticcam=torch.from_numpy(poses[2])[:3, :4], **kw15)
        plain = RN.render(16, 16, K16, chunk=512, c2w=torch.from_numpy(poses[0])[:3, :4], **kw15)[0]
    assert torch.equal(rgb, plain)
    del sig_last[:]
    save("g19_noviews_staticcam", seed=np.int64(SEED), K=np.array(K16), c2w=poses[0], c2w_static=poses[2], rgb=rgb.numpy(),
         disp=disp.numpy(), acc=acc.numpy(), rgb0=ex["rgb0"].numpy(), z_std=ex["z_std"].numpy())

    # ---- G20 N_importance = 32 on g17's rays and cotangent (the kernels specialised to 32 samples run their second fine pass
    # with two idle waves; this pins them to the reference) ------------------------------------------------------------
    rays = torch.stack([ro17, rd17], 0).clone().requires_grad_(True)
    with Capture(RN, RH) as cap:
        rgb, disp, acc, ex = RN.render(400, 400, O.YCBV_K, chunk=48, rays=rays, **dict(kwargs, N_importance=32))
    (g20g,) = torch.autograd.grad(rgb, rays, grad_outputs=cot17)
    del sig_last[:]
    save("g20_importance32", seed=np.int64(SEED), rays_o=ro17.numpy(), rays_d=rd17.numpy(), rgb=rgb.detach().numpy(),
         disp=disp.detach().numpy(), acc=acc.detach().numpy(), rgb0=ex["rgb0"].detach().numpy(),
         z_std=ex["z_std"].detach().numpy(), z_samples=cap.log[0]["samples"], inds=cap.log[0]["inds"].astype(np.int8),
         pdf_weights=cap.log[0]["weights"], cot=cot17.numpy(), grad_rays=g20g.numpy())

    # ---- linspace tables the host glue must reproduce (RN:439, RH:208) ------------------------
    save("g0_tables", t64=torch.linspace(0., 1., 64).numpy(), t128=torch.linspace(0., 1., 128).numpy())


if __name__ == "__main__":
    main()
