"""Round-5 golden vectors, produced by IMPORTING AND RUNNING the reference (same rules and shims as oracle/gen_golden.py,
whose helpers it uses; runs only in the authoring container, needs /root/reference read-only; nothing of the reference
travels -- inputs and outputs only).

    python oracle/gen_golden_r5.py            # writes tests/golden/g21_path_options.npz, g22 / g23 / g24_counts_*.npz, g25_wide_networks.npz
    GOLDEN_OUT=/tmp/regen python oracle/gen_golden_r5.py     # ... somewhere else, to check that they reproduce bit for bit

g21_path_options: the reference's render_path (RN:213-255) and render_path_grad (RN:126-210) called with
render_kwargs_TRAIN -- perturb = 1, raw_noise_std > 0 (RN:318-330): both functions forward **render_kwargs to render()
unchanged (RN:233, RN:168), so the stochastic options apply there too.  The reference draws from torch's global generator
inside render_rays, once per chunk of rays and in the order t_rand, coarse noise, u, fine noise; torch.rand / torch.randn are
wrapped to RECORD every draw in call order, so that a build fed the same numbers must reproduce the reference's images and
per-patch psi-gradients.

g25_wide_networks: NeRFs the fused kernels cannot hold (RH:70-97 is generic in D, W, skips; RN:439 / RN:474 in the sample
counts) -- wider than 256, deeper than 8, two skips, odd widths and sample counts, no view directions, coarse only --
instantiated from the reference's own class with the oracle's seeded weights (O.synth_weights_shape) and rendered by the
reference's render(): forward, what sample_pdf saw and produced, run_network on given points, and autograd's d rgb / d rays.

TEST INFRASTRUCTURE ONLY.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import nerf_oracle as O  # noqa: E402  (synthetic-weight recipe and camera constants only)
import gen_golden as G  # noqa: E402  (import_reference, build_nets, Capture, save)


def main():
    src = G.OUT                                   # g10 / g17 (inputs reused here) are read from the committed fixtures
    if os.environ.get("GOLDEN_OUT"):              # e.g. GOLDEN_OUT=/tmp/regen: write somewhere else (reproducibility check)
        G.OUT = os.environ["GOLDEN_OUT"]
        os.makedirs(G.OUT, exist_ok=True)
    RN, RH, LL = G.import_reference()
    sys.modules["imageio"].imwrite = lambda *a, **k: None        # render_path / render_path_grad write PNGs (RN:206, RN:250)
    torch.manual_seed(0)
    rng = np.random.RandomState(4321)
    SEED = 7
    nets, kwargs, embed_fn, embeddirs_fn = G.build_nets(RN, RH, SEED)
    g10 = np.load(os.path.join(src, "g10_path_grad.npz"))

    sig_last = []
    orig_r2o = RN.raw2outputs

    def r2o(raw, *a, **k):
        sig_last.append(raw[..., -1, 3].detach().numpy().copy())
        return orig_r2o(raw, *a, **k)
    RN.raw2outputs = r2o

    drawn = []
    orig_rand, orig_randn = torch.rand, torch.randn

    def rec(fn):
        def w(*a, **k):
            out = fn(*a, **k)
            drawn.append(out.detach().numpy().copy())
            return out
        return w

    NOISE_STD = 0.5
    kw_train = dict(kwargs, perturb=1.0, raw_noise_std=NOISE_STD)          # = render_kwargs_train of RN:318-330
    log = {"gumbel_noises": g10["gumbel"].tolist(), "uniform_noises": g10["uniform"].tolist(), "thetas": g10["thetas"].tolist()}
    psi = torch.tensor(g10["psi"])
    prob = torch.softmax(psi / 0.25, 0).requires_grad_()                   # NM:141-142
    poses = LL.sample_pose(prob, 2, 0.1, log)                              # the graph to psi (LL:202-247)
    assert np.array_equal(poses.detach().numpy(), g10["poses_grad"])
    Hs = 8
    Ks = O.scaled_K(50.0)
    hwf = [Hs, Hs, Ks[0][0]]
    chunk = 16

    # ---- render_path(render_kwargs_train): 2 poses x 4 chunks of 16 rays, 4 draws per chunk -------------------------
    tmp = tempfile.mkdtemp()
    torch.rand, torch.randn = rec(orig_rand), rec(orig_randn)
    try:
        with G.Capture(RN, RH) as cap:
            rgbs, disps = RN.render_path(None, poses.detach(), hwf, Ks, chunk, kw_train, savedir=tmp, object_id=2)
    finally:
        torch.rand, torch.randn = orig_rand, orig_randn
    n_chunks = 2 * (Hs * Hs // chunk)
    assert [d.shape for d in drawn] == [(16, 64), (16, 64), (16, 128), (16, 192)] * n_chunks, [d.shape for d in drawn]
    path_draws = list(drawn)
    del drawn[:]
    s0 = np.concatenate(sig_last[0::2])
    s1 = np.concatenate(sig_last[1::2])
    del sig_last[:]
    catc = lambda k: np.concatenate([c[k] for c in cap.log], 0)
    g = dict(seed=np.int64(SEED), noise_std=np.float64(NOISE_STD), K=np.array(Ks), chunk=np.int64(chunk),
             poses=poses.detach().numpy(),
             path_t_rand=np.stack(path_draws[0::4]), path_randn0=np.stack(path_draws[1::4]), path_u=np.stack(path_draws[2::4]),
             path_randn1=np.stack(path_draws[3::4]), path_rgbs=rgbs, path_disps=disps,
             path_pdf_weights=catc("weights"), path_inds=catc("inds").astype(np.int8), path_z_samples=catc("samples"),
             path_sigma0_last=s0, path_sigma_last=s1)
    # the coarse image, accumulation etc. are not returned by render_path: one more call of render() per pose with the SAME
    # draws replayed gives them (rgb0 / acc0 / acc for the census' coarse leg)
    replay = []

    def feed(*a, **k):
        return torch.from_numpy(replay.pop(0).copy())
    extra = {k: [] for k in ("acc", "rgb0", "acc0", "disp0", "z_std")}
    torch.rand, torch.randn = feed, feed
    try:
        for i in range(2):
            replay.extend(path_draws[16 * i:16 * (i + 1)])
            with torch.no_grad():
                rgb, disp, acc, ex = RN.render(Hs, Hs, Ks, chunk=chunk, c2w=poses[i, :3, :4].detach(), **kw_train)
            assert np.array_equal(rgb.numpy(), rgbs[i]) and np.array_equal(disp.numpy(), disps[i])      # the draws replay exactly
            extra["acc"].append(acc.numpy())
            for k in ("rgb0", "acc0", "disp0", "z_std"):
                extra[k].append(ex[k].numpy())
    finally:
        torch.rand, torch.randn = orig_rand, orig_randn
    del sig_last[:]
    g.update({"path_" + k: np.stack(v) for k, v in extra.items()})

    # ---- render_path_grad(render_kwargs_train): per pose, per 16-ray patch one render() call = one chunk = 4 draws ----
    gE = [{"grad_E": [torch.from_numpy(rng.standard_normal((3, Hs, Hs)).astype(np.float32))]} for _ in range(2)]
    torch.rand, torch.randn = rec(orig_rand), rec(orig_randn)
    try:
        rgbs_g, dl = RN.render_path_grad(prob, poses, hwf, Ks, chunk, gE, kw_train, savedir=None)
    finally:
        torch.rand, torch.randn = orig_rand, orig_randn
    assert [d.shape for d in drawn] == [(16, 64), (16, 64), (16, 128), (16, 192)] * n_chunks
    g.update(grad_E=np.stack([x["grad_E"][0].numpy() for x in gE]),
             grad_t_rand=np.stack(drawn[0::4]), grad_randn0=np.stack(drawn[1::4]), grad_u=np.stack(drawn[2::4]),
             grad_randn1=np.stack(drawn[3::4]), grad_rgbs=rgbs_g, dLdpsis=np.stack([d.numpy() for d in dl]))
    G.save("g21_path_options", **g)
    RN.raw2outputs = orig_r2o

    # ---- g22 / g23 / g24: other sample counts (RN:439 N_samples, RN:474 N_importance are arguments, NM:1258-1260) on g17's
    # rays and cotangent: (64, 96), (32, 64), (128, 128) -- forward, what sample_pdf saw and produced, and the gradient
    g17 = np.load(os.path.join(src, "g17_importance64.npz"))
    ro, rd, cot = torch.from_numpy(g17["rays_o"]), torch.from_numpy(g17["rays_d"]), torch.from_numpy(g17["cot"])
    for name, ns, ni in (("g22_counts_64_96", 64, 96), ("g23_counts_32_64", 32, 64), ("g24_counts_128_128", 128, 128)):
        kw = dict(kwargs, N_samples=ns, N_importance=ni)
        rays = torch.stack([ro, rd], 0).clone().requires_grad_(True)
        with G.Capture(RN, RH) as cap:
            rgb, disp, acc, ex = RN.render(400, 400, O.YCBV_K, chunk=len(ro), rays=rays, **kw)
        (gr,) = torch.autograd.grad(rgb, rays, grad_outputs=cot)
        assert cap.log[0]["weights"].shape == (len(ro), ns - 2) and cap.log[0]["samples"].shape == (len(ro), ni)
        G.save(name, seed=np.int64(SEED), n_samples=np.int64(ns), n_importance=np.int64(ni), rays_o=ro.numpy(), rays_d=rd.numpy(),
               rgb=rgb.detach().numpy(), disp=disp.detach().numpy(), acc=acc.detach().numpy(), rgb0=ex["rgb0"].detach().numpy(),
               acc0=ex["acc0"].detach().numpy(), z_std=ex["z_std"].detach().numpy(), z_samples=cap.log[0]["samples"],
               inds=cap.log[0]["inds"].astype(np.int16), pdf_weights=cap.log[0]["weights"], cdf=cap.log[0]["cdf"],
               cot=cot.numpy(), grad_rays=gr.numpy())

    # ---- g25: networks and sample counts only the layered renderer (include/nsr_wide.h) serves -----------------------------
    g25 = dict(seed=np.int64(SEED), rays_o=ro.numpy(), rays_d=rd.numpy(), cot=cot.numpy())
    prng = np.random.RandomState(99)
    pts = prng.uniform(-1.5, 1.5, (96, 3)).astype(np.float32)
    dirs = prng.standard_normal((96, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    g25.update(pts=pts, dirs=dirs)
    for tag, (D, Wd, L, Lv, skips, uv, ns, ni) in WIDE_CASES.items():
        sdc = O.synth_weights_shape(SEED + 41, D, Wd, L, Lv, skips, uv)
        sdf = {k: (v * (1.0 + 0.05 * np.random.RandomState(SEED + 42).standard_normal(v.shape))).astype(np.float32)
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
        kw = dict(kwargs, network_query_fn=q, network_fn=nn_[0], network_fine=nn_[1] if ni > 0 else None, use_viewdirs=uv,
                  N_samples=ns, N_importance=ni)
        with torch.no_grad():
            net_out = q(torch.from_numpy(pts)[:, None, :], torch.from_numpy(dirs) if uv else None, nn_[0])[:, 0]
        rays = torch.stack([ro, rd], 0).clone().requires_grad_(True)
        with G.Capture(RN, RH) as cap:
            rgb, disp, acc, ex = RN.render(400, 400, O.YCBV_K, chunk=len(ro), rays=rays, **kw)
        (gr,) = torch.autograd.grad(rgb, rays, grad_outputs=cot)
        g25.update({tag + "_shape": np.array([D, Wd, L, Lv, int(uv), ns, ni] + list(skips)), tag + "_net_out": net_out.numpy(),
                    tag + "_rgb": rgb.detach().numpy(), tag + "_disp": disp.detach().numpy(), tag + "_acc": acc.detach().numpy(),
                    tag + "_grad_rays": gr.numpy()})
        if ni > 0:
            assert cap.log[0]["weights"].shape == (len(ro), ns - 2) and cap.log[0]["samples"].shape == (len(ro), ni)
            g25.update({tag + "_rgb0": ex["rgb0"].detach().numpy(), tag + "_acc0": ex["acc0"].detach().numpy(),
                        tag + "_disp0": ex["disp0"].detach().numpy(), tag + "_z_std": ex["z_std"].detach().numpy(),
                        tag + "_z_samples": cap.log[0]["samples"], tag + "_inds": cap.log[0]["inds"].astype(np.int16),
                        tag + "_pdf_weights": cap.log[0]["weights"], tag + "_cdf": cap.log[0]["cdf"]})
    G.save("g25_wide_networks", **g25)


# tag -> (D, W, multires, multires_views, skips, use_viewdirs, N_samples, N_importance)
WIDE_CASES = {
    "a": (10, 384, 10, 4, [4], True, 64, 128),      # wider and deeper than the fused kernels' 8 x 256
    "b": (6, 300, 6, 2, [1, 3], True, 48, 100),     # two skips, a width that is no multiple of 32, odd sample counts
    "c": (9, 272, 10, 4, [5], False, 24, 40),       # no view directions (output_linear, 5 rows), a late skip
    "d": (3, 512, 4, 1, [], True, 20, 0),           # coarse only, no skip
}


if __name__ == "__main__":
    main()
