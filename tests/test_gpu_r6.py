"""GPU tests added in round 6 (run with -m gpu on an MI355X): parity on a TRAINED pair of networks.

Every other fixture uses one synthetic weight family (nn.Linear-uniform init, scaled).  tests/golden/g26_trained.npz holds a
pair trained by the reference's OWN code on an analytic textured box (oracle/train_g26.py) and the reference's render of it:
densities of several hundred, four fifths of the samples in empty space, opaque rays whose empty resampling bins sit at the
1e-5 denominator switch (RH:238-239), hidden activations an order of magnitude beyond the synthetic family's -- what drives the
three discontinuities of the path and f16x2's range.  tests/test_oracle_golden.py pins the oracle to the reference on it."""
import os
import sys

import numpy as np
import pytest

from conftest import assert_close, census_ref, load_golden, trained_pair

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

KERNELS = ["f16x2", "bf16x3", "fp32", "x32", "layered-bf16x3", "layered-fp32", "layered-f16x2"]


def cpu(t):
    return t.detach().cpu().numpy()


def _census_mod():
    p = os.path.join(ROOT, "oracle")
    if p not in sys.path:
        sys.path.insert(0, p)
    import census
    return census


def _model(kernel, sd_c, sd_f):
    if kernel.startswith("layered-"):
        from neural_sim_nerf_amd.wide import WideModel
        return WideModel(sd_c, sd_f, mlp=kernel[len("layered-"):])
    from neural_sim_nerf_amd.engine import NsrModel
    return NsrModel(sd_c, sd_f, variant=32) if kernel == "x32" else NsrModel(sd_c, sd_f, mlp=kernel)


def _rel_rows(a, b):
    return np.linalg.norm(a - b, axis=1) / (np.linalg.norm(b, axis=1) + 1e-12)


@pytest.mark.parametrize("kernel", KERNELS)
def test_trained_network_stagewise_census_and_gradient(kernel, oracle):
    """The 40 x 40 view of g26 on every arithmetic of the fused kernels and of the layered renderer: (1) stage by stage against
    the oracle on the kernel's own intermediates -- network outputs relative to their size (sigma reaches several hundred),
    coarse weights 2e-6, cdf -> indices -> samples and the merged depths BIT FOR BIT, pixels 3e-6; (2) END TO END against the
    reference's own render: every ray beyond 1e-4 attributed (oracle/census.py), PSNR-delta <= 0.1 dB, and the counts printed
    next to the synthetic family's; (3) the input gradient at the reference's own depths against the reference's autograd;
    (4) f16x2: how many points left the fp16 range (they are re-rendered on bf16 MFMAs inside the call) and the pack-time
    head-room of this network."""
    C = _census_mod()
    g = load_golden("g26_trained")
    sd_c, sd_f = trained_pair(g)
    near, far = oracle.YCBV_NEAR, oracle.YCBV_FAR
    K = g["K40"].tolist()
    ro, rd = oracle.get_rays(40, 40, K, g["c2w"][:3, :4])
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    n = len(ro)
    vd = oracle.normalize_dirs(rd)
    m = _model(kernel, sd_c, sd_f)
    r = m.render_views(g["c2w"], 40, 40, K, near, far, debug=True)
    # ---- (1) stage by stage on the kernel's own intermediates
    z = oracle.coarse_z(np.full(n, near, np.float32), np.full(n, far, np.float32))
    raw0 = oracle.run_network(sd_c, (ro[:, None] + rd[:, None] * z[..., None]).astype(np.float32), vd)
    k_raw0 = cpu(r["raw0"])
    # fp32-grade: 5e-5 absolute on outputs of order 1 (the synthetic family's bound), relative on the large densities
    assert_close(k_raw0, raw0, atol=5e-5, rtol=2e-5, what="coarse raw")
    rgb0, _, acc0, w0, _ = oracle.raw2outputs(k_raw0, z, rd)
    assert_close(cpu(r["weights0"]), w0, atol=2e-6, what="weights0 | own raw")
    assert_close(cpu(r["rgb0"]), rgb0, atol=3e-6, what="rgb0 | own raw")
    z_mid = (np.float32(0.5) * (z[:, 1:] + z[:, :-1])).astype(np.float32)
    s, inds, _ = oracle.sample_pdf(z_mid, cpu(r["weights0"])[:, 1:-1])
    assert np.array_equal(cpu(r["inds"]), inds) and np.array_equal(cpu(r["z_samples"]), s)       # bit for bit
    zf = np.sort(np.concatenate([z, s], -1), -1)
    assert np.array_equal(cpu(r["z_fine"]), zf)
    raw = oracle.run_network(sd_f, (ro[:, None] + rd[:, None] * zf[..., None]).astype(np.float32), vd)
    k_raw = cpu(r["raw"])
    assert_close(k_raw, raw, atol=5e-5, rtol=2e-5, what="fine raw | own z")
    rgb, _, acc, _, _ = oracle.raw2outputs(k_raw, zf, rd)
    assert_close(cpu(r["rgb_map"]), rgb, atol=3e-6, what="rgb | own raw")
    assert_close(cpu(r["acc_map"]), acc, atol=3e-6, what="acc | own raw")
    # ---- (2) end to end against the REFERENCE's render of these weights
    taps = ("rgb_map", "acc_map", "disp_map", "rgb0", "acc0", "raw0", "weights0", "inds", "z_samples", "z_fine", "raw")
    c = C.census((sd_c, sd_f), ro, rd, near, far, {k: cpu(r[k]) for k in taps}, census_ref(g))
    inds_eq = float((cpu(r["inds"]) == g["inds"].astype(np.int64)).mean())
    print("%s on the TRAINED pair vs the reference: %s, inds equal to the reference's end to end: %.6f, max |rgb| %.2e |acc| %.2e"
          % (kernel, {k: c[k] for k in ("rays", "rays_above_tol", "cliff_rays", "index_flip_rays", "denom_switch_rays",
                                        "illconditioned_shift_rays", "unattributed", "psnr_delta_db")}, inds_eq,
             np.abs(cpu(r["rgb_map"]) - g["rgb"]).max(), np.abs(cpu(r["acc_map"]) - g["acc"]).max()))
    assert C.passes(c) and c["unattributed"] == 0 and c["psnr_delta_db"] <= 0.1, c
    assert c["rays_above_tol"] <= 0.05 * c["rays"], c
    assert C.psnr_delta(cpu(r["rgb_map"]), g["rgb"]) <= 0.1
    assert_close(cpu(r["rgb0"]), g["rgb0"], atol=1e-5, what="coarse image vs the reference")
    # ---- (3) the input gradient AT the reference's own depths (z_samples is detached, RN:475) against its autograd
    gro, grd = g["grad_rays_in"]
    zg = oracle.coarse_z(np.full(len(gro), near, np.float32), np.full(len(gro), far, np.float32))
    zf_ref = np.sort(np.concatenate([zg, g["grad_z_samples"]], -1), -1)
    out = m.render_rays_vjp(gro, grd, near, far, g["cot"], z_fine=zf_ref)
    for a, b, what in ((cpu(out[0]), g["grad_rays"][0], "grad_o"), (cpu(out[1]), g["grad_rays"][1], "grad_d")):
        e = _rel_rows(a, b)
        print("%s %s vs the reference's autograd: median %.2e  90 %% %.2e  max %.2e" % (kernel, what, np.median(e), np.percentile(e, 90), e.max()))
        assert np.isfinite(a).all() and np.percentile(e, 90) < 3e-4 and np.linalg.norm(a - b) / np.linalg.norm(b) < 2e-2, \
            (what, np.percentile(e, 90), e.max())
    # ---- (4) the range of f16x2 on a trained network
    st = m.range_status()
    if kernel == "f16x2":
        from neural_sim_nerf_amd import pack
        for name, sd in (("coarse", sd_c), ("fine", sd_f)):
            rows = pack.h2_report(sd)
            print("f16x2 pack-time head-room of the trained %s network (bits, typical / worst case): %s" % (
                name, [(round(x["headroom_typical_bits"], 1), round(x["headroom_worst_bits"], 1)) for x in rows]))
            assert min(x["headroom_typical_bits"] for x in rows) > 0
        print("f16x2 range status on the trained pair (points / rays re-rendered on bf16 MFMAs, dropped):", st)
        assert st["dropped_items"] == 0
    else:
        assert st["points"] == 0 and st["dropped_items"] == 0 and st.get("passes_rerun", 0) == 0, st
    m.close()


def test_trained_network_through_the_dropin_api(oracle):
    """The same pair loaded the way the reference's consumers load a checkpoint (load_state_dict on NeRF modules, RN:296-314)
    and rendered through render(c2w=...) / render(rays=...) + autograd: the reference's pixels within the end-to-end rule, its
    gradient direction, no range warning that drops anything."""
    import torch
    import neural_sim_nerf_amd.run_nerf_noscale as R
    C = _census_mod()
    g = load_golden("g26_trained")
    sd_c, sd_f = trained_pair(g)
    nets = []
    for sd in (sd_c, sd_f):
        net = R.NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        nets.append(net.to(R.device))
    kw = dict(network_query_fn=None, perturb=False, N_importance=128, network_fine=nets[1], N_samples=64, network_fn=nets[0],
              use_viewdirs=True, white_bkgd=False, raw_noise_std=0., ndc=False, lindisp=False, near=oracle.YCBV_NEAR, far=oracle.YCBV_FAR)
    with torch.no_grad():
        rgb, disp, acc, ex = R.render(40, 40, g["K40"].tolist(), chunk=512, c2w=torch.from_numpy(g["c2w"][:3, :4]), **kw)
    d = np.abs(cpu(rgb).reshape(-1, 3) - g["rgb"]).max(-1)
    print("drop-in render of the trained pair vs the reference: %d of 1600 rays beyond 1e-4, PSNR-delta %.4f dB"
          % ((d > 1e-4).sum(), C.psnr_delta(cpu(rgb).reshape(-1, 3), g["rgb"])))
    assert C.psnr_delta(cpu(rgb).reshape(-1, 3), g["rgb"]) <= 0.1 and (d > 1e-4).mean() <= 0.05
    assert_close(cpu(ex["rgb0"]).reshape(-1, 3), g["rgb0"], atol=1e-5, what="rgb0 vs reference")
    rays = torch.from_numpy(g["grad_rays_in"]).to(R.device).requires_grad_(True)
    rgb_p = R.render(40, 40, g["K40"].tolist(), chunk=128, rays=rays, **kw)[0]
    (gr,) = torch.autograd.grad(rgb_p, rays, grad_outputs=torch.from_numpy(g["cot"]).to(R.device))
    assert C.psnr_delta(cpu(rgb_p), g["grad_rgb"]) <= 0.1
    rel = np.linalg.norm(cpu(gr) - g["grad_rays"]) / np.linalg.norm(g["grad_rays"])
    print("drop-in autograd vs the reference's (own depths on both sides): relative difference %.3e" % rel)
    assert rel < 0.2          # (the depths are each side's own: a flipped resampling index moves a ray's gradient; direction and size agree)
    for n_ in nets:
        n_.invalidate()


def test_module_level_sample_pdf_any_counts_and_both_branches(oracle):
    """run_nerf_helpers.sample_pdf (RH:199-243) as a function of its own: any bin / sample count, det=True, det=False (torch.rand of
    the reference's shape, drawn by the wrapper) and pytest=True (numpy's generator reseeded with 0, RH:214-222).  r05 served
    63 bins / 128 samples / det=True only.  Against the oracle (pinned to the reference's sample_pdf: g5, g6, g14, g22-g26) BIT FOR BIT."""
    import torch
    import neural_sim_nerf_amd.run_nerf_helpers as H
    from neural_sim_nerf_amd import wide
    rng = np.random.RandomState(5)
    for n_bins, n_smp, rows in ((63, 128, 300), (47, 100, 65), (2, 7, 5), (200, 512, 33), (511, 1, 4)):
        bins = np.sort(rng.uniform(0.3, 1.9, (rows, n_bins)).astype(np.float32), -1)
        w = rng.uniform(0, 1, (rows, n_bins - 1)).astype(np.float32) ** 4
        w[0] = 0.0                                                   # (all-zero weights: uniform pdf, RH:201)
        if n_bins > 8:
            w[1, :] = 0.0
            w[1, 3] = 1.0                                            # a spike: most bins empty, denominators at the 1e-5 switch
        want, want_i, _ = oracle.sample_pdf(bins, w, n_smp)
        got = H.sample_pdf(torch.tensor(bins), torch.tensor(w), n_smp, det=True)
        assert got.is_cuda and np.array_equal(cpu(got), want), (n_bins, n_smp)
        s2, i2 = wide.sample_pdf(bins, w, oracle.torch_linspace01(n_smp))
        assert np.array_equal(cpu(i2), want_i) and np.array_equal(cpu(s2), want)
        # det=False: the wrapper draws torch.rand(rows, n_smp) on the bins' device exactly once
        torch.manual_seed(11)
        got = H.sample_pdf(torch.tensor(bins, device="cuda"), torch.tensor(w, device="cuda"), n_smp, det=False)
        torch.manual_seed(11)
        u = torch.rand([rows, n_smp], device="cuda")
        want_u, _, _ = oracle.sample_pdf(bins, w, n_smp, u=cpu(u))
        assert np.array_equal(cpu(got), want_u), (n_bins, n_smp)
        # pytest=True: numpy's draws
        got = H.sample_pdf(torch.tensor(bins), torch.tensor(w), n_smp, det=False, pytest=True)
        np.random.seed(0)
        want_p, _, _ = oracle.sample_pdf(bins, w, n_smp, u=np.random.rand(rows, n_smp).astype(np.float32))
        assert np.array_equal(cpu(got), want_p)
    # leading batch dimensions like the reference's [..., n_bins]
    bins = np.sort(rng.uniform(0.3, 1.9, (3, 5, 20)).astype(np.float32), -1)
    w = rng.uniform(0, 1, (3, 5, 19)).astype(np.float32)
    got = H.sample_pdf(torch.tensor(bins), torch.tensor(w), 9, det=True)
    assert tuple(got.shape) == (3, 5, 9)
    assert np.array_equal(cpu(got).reshape(15, 9), oracle.sample_pdf(bins.reshape(15, 20), w.reshape(15, 19), 9)[0])
    with pytest.raises(NotImplementedError):
        H.sample_pdf(torch.zeros(2, 600), torch.zeros(2, 599), 8, det=True)


def test_embedder_is_differentiable(oracle):
    """Embedder.embed (RH:39-48) with requires_grad input: forward = the native encoding, backward = nsrw_embed_vjp; against
    autograd through a float64 torch restatement of the same function (r05 refused a differentiable embed)."""
    import torch
    import neural_sim_nerf_amd.run_nerf_helpers as H
    rng = np.random.RandomState(3)
    for L in (10, 4, 0):
        fn, out_dim = H.get_embedder(L, 0) if L else (None, 3)
        if fn is None:
            continue
        x = torch.tensor((rng.rand(257, 3).astype(np.float32) - 0.5) * 3.0, device="cuda", requires_grad=True)
        g = torch.tensor(rng.standard_normal((257, out_dim)).astype(np.float32), device="cuda")
        y = fn(x)
        assert y.requires_grad and tuple(y.shape) == (257, out_dim)
        assert_close(cpu(y), oracle.embed(cpu(x), L), atol=2.5e-7, what="embed forward")
        (gx,) = torch.autograd.grad(y, x, grad_outputs=g)
        x64 = x.detach().double().cpu().requires_grad_(True)
        parts = [x64]
        for l in range(L):
            a = (x.detach() * np.float32(2.0 ** l)).double().cpu()                # the forward's own fp32 argument
            a = a + (x64 - x64.detach()) * (2.0 ** l)                               # ... differentiable in x
            parts += [torch.sin(a), torch.cos(a)]
        (want,) = torch.autograd.grad(torch.cat(parts, -1), x64, grad_outputs=g.double().cpu())
        scale = np.abs(cpu(want)).max()
        assert np.abs(cpu(gx) - cpu(want)).max() <= 1e-6 * scale, (L, np.abs(cpu(gx) - cpu(want)).max(), scale)
        # a CPU input gets a CPU gradient back; no_grad / detached inputs take the plain forward
        xc = x.detach().cpu().requires_grad_(True)
        (gc,) = torch.autograd.grad(fn(xc), xc, grad_outputs=g)
        assert not gc.is_cuda and np.array_equal(cpu(gc), cpu(gx))
        with torch.no_grad():
            assert not fn(x).requires_grad


def test_a_module_of_the_references_own_class_layout_is_adopted(oracle):
    """network_fn / network_fine that are NOT neural_sim_nerf_amd modules but have the reference's layout (RH:70-97) -- what a
    caller holds who built the networks with the reference's own class -- are served through a wrapper SHARING their parameters
    (NeRF.adopt; r05 refused them): same pixels as the drop-in module, and a weight update on the foreign module is seen."""
    import torch
    import torch.nn as nn
    import neural_sim_nerf_amd.run_nerf_noscale as R

    class Foreign(nn.Module):                                        # the attribute / layer names of RH:70-97, nothing else
        def __init__(self, D=8, W=256, input_ch=63, input_ch_views=27, skips=(4,)):
            super().__init__()
            self.D, self.W, self.input_ch, self.input_ch_views, self.skips, self.use_viewdirs = D, W, input_ch, input_ch_views, list(skips), True
            self.pts_linears = nn.ModuleList([nn.Linear(input_ch, W)] + [nn.Linear(W + input_ch if i in skips else W, W) for i in range(D - 1)])
            self.views_linears = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
            self.feature_linear, self.alpha_linear, self.rgb_linear = nn.Linear(W, W), nn.Linear(W, 1), nn.Linear(W // 2, 3)

    sd_c = oracle.synth_weights(0)
    sd_f = oracle.synth_weights(1000, fine_of=sd_c)
    near, far = oracle.YCBV_NEAR, oracle.YCBV_FAR
    K = oracle.scaled_K(20.0)
    pose = torch.as_tensor(np.asarray(oracle.sweep_poses(1, seed=4))[0][:3, :4])
    foreign, ours = [], []
    for sd in (sd_c, sd_f):
        f = Foreign()
        f.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        foreign.append(f.to(R.device))
        n = R.NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
        n.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        ours.append(n.to(R.device))
    kw = dict(network_query_fn=None, perturb=False, N_importance=128, N_samples=64, use_viewdirs=True, white_bkgd=False,
              raw_noise_std=0., ndc=False, lindisp=False, near=near, far=far)
    with torch.no_grad():
        a = R.render(20, 20, K, c2w=pose, network_fn=foreign[0], network_fine=foreign[1], **kw)[0]
        b = R.render(20, 20, K, c2w=pose, network_fn=ours[0], network_fine=ours[1], **kw)[0]
    assert np.array_equal(cpu(a), cpu(b), equal_nan=True)
    ad = R.NeRF.adopt(foreign[1])
    assert ad is R.NeRF.adopt(foreign[1]) and ad.rgb_linear.weight is foreign[1].rgb_linear.weight
    with torch.no_grad():                                              # an update of the FOREIGN module's weights ...
        foreign[1].rgb_linear.bias.add_(0.5)
        ours[1].rgb_linear.bias.add_(0.5)
        a2 = R.render(20, 20, K, c2w=pose, network_fn=foreign[0], network_fine=foreign[1], **kw)[0]
        b2 = R.render(20, 20, K, c2w=pose, network_fn=ours[0], network_fine=ours[1], **kw)[0]
    assert np.array_equal(cpu(a2), cpu(b2), equal_nan=True) and not np.array_equal(cpu(a2), cpu(a))      # ... is rendered
    with pytest.raises(NotImplementedError, match="reference's NeRF layout"):
        R.render(20, 20, K, c2w=pose, network_fn=nn.Linear(3, 4).to(R.device), network_fine=None, **dict(kw, N_importance=0))
    for n_ in ours:
        n_.invalidate()


def test_random_network_shapes_that_the_fused_kernels_serve_by_re_expression(oracle):
    """Twelve seeded random shapes INSIDE fits_kernel (depth 1..8, even width 8..256, 0..10 / 0..4 frequencies, no skip or one skip
    the kernels' skip can absorb, with and without view directions, 4 or 5 output rows): each is served by the fused 8 x 256 kernels
    through the host-side re-expression of its weights (run_nerf_helpers.as_kernel_network: zero padding, identity layers, the
    +y / -y view layer).  NeRF.evaluate against the oracle's evaluation of the ORIGINAL network on random points, a 12 x 12 render
    against the oracle's render of the original network, the input gradient finite and close.  What g15 / g16 pin on two shapes."""
    import torch
    import neural_sim_nerf_amd.run_nerf_noscale as R
    from neural_sim_nerf_amd.run_nerf_helpers import fits_kernel
    near, far = oracle.YCBV_NEAR, oracle.YCBV_FAR
    rng = np.random.RandomState(77)
    K = oracle.scaled_K(400.0 / 12)
    pose = np.asarray(oracle.sweep_poses(1, seed=6))[0]
    ro, rd = (a.reshape(-1, 3) for a in oracle.get_rays(12, 12, K, pose[:3, :4]))
    vd = oracle.normalize_dirs(rd)
    done = 0
    while done < 12:
        D = int(rng.randint(1, 9))
        W = int(2 * rng.randint(4, 129))
        L, Lv = int(rng.randint(0, 11)), int(rng.randint(0, 5))
        uv = bool(rng.randint(0, 2))
        oc = int(rng.choice([4, 5]))
        skips = [int(rng.randint(0, max(D - 1, 1)))] if (D > 1 and rng.randint(0, 2)) else []
        in_ch, in_v = 3 + 6 * L, (3 + 6 * Lv if uv else 0)
        if fits_kernel(D, W, in_ch, in_v, skips, uv, oc):
            continue
        done += 1
        tag = "%d x %d skips %s L %d Lv %d viewdirs %s oc %d" % (D, W, skips, L, Lv, uv, oc)
        sd_c = oracle.synth_weights_shape(300 + done, D, W, L, Lv, skips, uv, oc)
        sd_f = oracle.synth_weights_shape(400 + done, D, W, L, Lv, skips, uv, oc)
        nets = []
        for sd in (sd_c, sd_f):
            n_ = R.NeRF(D=D, W=W, input_ch=in_ch, output_ch=oc, skips=skips, input_ch_views=in_v, use_viewdirs=uv)
            assert n_.fused_why_not is None, (tag, n_.fused_why_not)
            n_.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
            nets.append(n_.to(R.device))
        pts = (rng.rand(200, 3).astype(np.float32) - 0.5) * 0.5
        dirs = oracle.normalize_dirs(rng.standard_normal((200, 3)).astype(np.float32))
        want = oracle.run_network(sd_f, pts[:, None], dirs)[:, 0]
        got = cpu(nets[1].evaluate(torch.tensor(pts, device=R.device), torch.tensor(dirs, device=R.device)))
        assert_close(got[:, :4], want[:, :4], atol=5e-5 * max(1.0, float(np.abs(want).max())), rtol=5e-5, what=tag + ": evaluate")
        kw = dict(network_query_fn=None, perturb=False, N_importance=128, network_fine=nets[1], N_samples=64, network_fn=nets[0],
                  use_viewdirs=uv, white_bkgd=False, raw_noise_std=0., ndc=False, lindisp=False, near=near, far=far)
        rays = torch.tensor(np.stack([ro, rd]), device=R.device, requires_grad=True)
        rgb, _, acc, ex = R.render(12, 12, K, rays=rays, **kw)
        assert R._model_for(nets[0], nets[1], 128, kw).mlp == "f16x2", tag
        ref = oracle.render_rays(sd_c, sd_f, ro, rd, vd, near, far)
        d = np.abs(cpu(rgb) - ref["rgb_map"]).max(-1)
        assert_close(cpu(ex["rgb0"]), ref["rgb0"], atol=1e-5, what=tag + ": coarse image")
        assert (d > 1e-4).mean() <= 0.1 and oracle.psnr(cpu(rgb), ref["rgb_map"]) > 50.0, (tag, (d > 1e-4).sum(), d.max())
        (gr,) = torch.autograd.grad(rgb, rays, grad_outputs=torch.ones_like(rgb))
        assert np.isfinite(cpu(gr)).all(), tag
        for n_ in nets:
            n_.invalidate()

