"""GPU tests of the LAYERED renderer (include/nsr_wide.h, neural_sim_nerf_amd/wide.py; run with -m gpu on an MI355X): the same
path as the fused kernels -- render_rays and its input-side VJP -- for the networks and sample counts they are not built for.
Parity is the oracle's (pinned to the reference on exactly these cases by tests/test_oracle_golden.py: g25), stage by stage
on the renderer's own intermediates, end to end through the census and PSNR-delta against the reference's own pixels, and
the gradient against the oracle's fp64 backprop and the reference's autograd."""
import os
import sys

import numpy as np
import pytest

from conftest import assert_close, load_golden
from test_oracle_golden import wide_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def cpu(t):
    return t.detach().cpu().numpy()


def _census():
    p = os.path.join(ROOT, "oracle")
    if p not in sys.path:
        sys.path.insert(0, p)
    import census
    return census


def _rel_rows(a, b):
    return np.linalg.norm(a - b, axis=1) / (np.linalg.norm(b, axis=1) + 1e-12)


MLPS = ["bf16x3", "fp32"]          # the two arithmetics of the layer GEMMs (wide.WideModel(mlp=...)): every parity bound is the same
MLPS3 = MLPS + ["f16x2"]           # ... and the forward passes on fp16 MFMAs with two-piece operands (r06), same bounds again


def _wide(mlp):
    """WideModel with the arithmetic fixed (the tests must not depend on $NSR_WIDE_MLP / the default)"""
    from neural_sim_nerf_amd.wide import WideModel
    return lambda *a, **k: WideModel(*a, mlp=mlp, **k)


@pytest.mark.parametrize("mlp", MLPS3)
@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_layered_renderer_stagewise_census_and_gradient(tag, mlp, oracle):
    """g25 a-d: 10 x 384 at (64, 128); 6 x 300 with two skips at (48, 100); 9 x 272 without view directions at (24, 40); a
    coarse-only 3 x 512 at (20, 0) -- on fp32 MFMAs and on bf16 MFMAs with three-way split operands, same bounds."""
    WideModel = _wide(mlp)
    C = _census()
    g = load_golden("g25_wide_networks")
    sd_c, sd_f, ns, ni = wide_case(oracle, g, tag)
    fine = ni > 0
    near, far = oracle.YCBV_NEAR, oracle.YCBV_FAR
    ro, rd = g["rays_o"], g["rays_d"]
    n = len(ro)
    vd = oracle.normalize_dirs(rd)
    m = WideModel(sd_c, sd_f if fine else None, n_samples=ns, n_importance=ni)
    # run_network on given points against the reference's own outputs
    want = g[tag + "_net_out"]
    got = cpu(m.run_network(g["pts"], g["dirs"], 0))
    assert got.shape == want.shape or got.shape == (len(want), 4)
    assert_close(got[:, :want.shape[1]], want[:, :got.shape[1]], atol=1e-5 + 2e-6 * np.abs(want).max(), rtol=2e-6, what="run_network")
    r = m.render_rays(ro, rd, near, far, debug=True)
    # ---- stage by stage on the renderer's own intermediates
    z = oracle.coarse_z(np.full(n, near, np.float32), np.full(n, far, np.float32), n=ns)
    raw0 = oracle.run_network(sd_c, (ro[:, None] + rd[:, None] * z[..., None]).astype(np.float32), vd)
    k_raw0 = cpu(r["raw0"])[..., :4]
    assert_close(k_raw0, raw0, atol=5e-5, rtol=5e-5, what="coarse raw")
    rgb0, disp0, acc0, w0, _ = oracle.raw2outputs(k_raw0, z, rd)
    assert_close(cpu(r["weights0"]), w0, atol=2e-6, what="weights0 | own raw")
    assert_close(cpu(r["rgb0"] if fine else r["rgb_map"]), rgb0, atol=3e-6, what="coarse rgb | own raw")
    assert_close(cpu(r["acc0"] if fine else r["acc_map"]), acc0, atol=3e-6, what="coarse acc | own raw")
    if not fine:
        assert np.array_equal(cpu(r["z_coarse"]), z)
        assert_close(cpu(r["rgb_map"]), g[tag + "_rgb"], atol=1e-5, what="rgb vs reference")
        assert_close(cpu(r["acc_map"]), g[tag + "_acc"], atol=1e-5, what="acc vs reference")
        zf = z
    else:
        z_mid = (np.float32(0.5) * (z[:, 1:] + z[:, :-1])).astype(np.float32)
        s, inds, _ = oracle.sample_pdf(z_mid, cpu(r["weights0"])[:, 1:-1], ni)
        assert np.array_equal(cpu(r["inds"]), inds) and np.array_equal(cpu(r["z_samples"]), s)       # bit for bit
        zf = np.sort(np.concatenate([z, s], -1), -1)
        assert np.array_equal(cpu(r["z_fine"]), zf)
        raw = oracle.run_network(sd_f, (ro[:, None] + rd[:, None] * zf[..., None]).astype(np.float32), vd)
        k_raw = cpu(r["raw"])[..., :4]
        assert_close(k_raw, raw, atol=5e-5, rtol=5e-5, what="fine raw | own z")
        rgb, disp, acc, _, _ = oracle.raw2outputs(k_raw, zf, rd)
        assert_close(cpu(r["rgb_map"]), rgb, atol=3e-6, what="rgb | own raw")
        assert_close(cpu(r["acc_map"]), acc, atol=3e-6, what="acc | own raw")
        with np.errstate(invalid="ignore", divide="ignore"):       # disp = acc / depth (RN:381): a 3e-6 change of acc is 3e-6 / acc of it
            dd = np.abs(cpu(r["disp_map"]) - disp) / np.abs(disp) * np.clip(acc, 0.0, 1.0)
        assert np.array_equal(np.isnan(cpu(r["disp_map"])), np.isnan(disp)) and np.nanmax(dd) < 2e-5, np.nanmax(dd)
        assert_close(cpu(r["z_std"]), np.std(s.astype(np.float64), -1), atol=1e-6, what="z_std")
        # ---- end to end: coarse image against the reference, census against the oracle's render, PSNR-delta against the
        # reference's pixels
        assert_close(cpu(r["rgb0"]), g[tag + "_rgb0"], atol=1e-5, what="rgb0 vs reference")
        ref = oracle.render_rays(sd_c, sd_f, ro, rd, vd, near, far, n_samples=ns, n_importance=ni, extras=True)
        taps = ("rgb_map", "acc_map", "disp_map", "rgb0", "acc0", "weights0", "inds", "z_samples", "z_fine")
        got = {k: cpu(r[k]) for k in taps}
        got.update(raw0=k_raw0, raw=k_raw)
        c = C.census((sd_c, sd_f), ro, rd, near, far, got, ref, n_importance=ni, n_samples=ns)
        print(tag, "census:", {k: c[k] for k in ("rays", "rays_above_tol", "unattributed", "psnr_delta_db")})
        assert C.passes(c) and c["rays_above_tol"] <= 4 and c["psnr_delta_db"] <= 0.1, c
        assert C.psnr_delta(cpu(r["rgb_map"]), g[tag + "_rgb"]) <= 0.1
    # ---- input gradient: against the oracle's fp64 backprop at the renderer's own depths, and against the reference's autograd
    go, gd, fwd = m.render_rays_vjp(ro, rd, near, far, g["cot"], with_forward=True)
    assert np.array_equal(cpu(fwd["rgb_map"]), cpu(r["rgb_map"]), equal_nan=True)
    wo, wd, _ = oracle.render_rays_vjp(sd_c, sd_f if fine else None, ro, rd, near, far, g["cot"], n_samples=ns, n_importance=ni, z_fine=zf)
    for a, b, what in ((cpu(go), wo, "grad_o"), (cpu(gd), wd, "grad_d")):
        e = _rel_rows(a, b)
        print(tag, what, "rel. error per ray: median %.2e  90 %% %.2e  max %.2e" % (np.median(e), np.percentile(e, 90), e.max()))
        assert np.percentile(e, 90) < 1e-4 and np.linalg.norm(a - b) / np.linalg.norm(b) < 1e-2, (what, np.percentile(e, 90), e.max())
    for a, b in ((cpu(go), g[tag + "_grad_rays"][0]), (cpu(gd), g[tag + "_grad_rays"][1])):
        assert np.linalg.norm(a - b) / np.linalg.norm(b) < 5e-2        # (the reference's depths differ by its own rounding)
    st = m.range_status()
    assert st["passes_rerun"] == 0 and st["dropped_items"] == 0 and (st["passes"] > 0) == (mlp == "f16x2"), st
    if fine:         # ... and AT the reference's own depths (z_samples is detached, RN:475): autograd's numbers, ray by ray
        zf_ref = np.sort(np.concatenate([z, g[tag + "_z_samples"]], -1), -1)
        ro_, rd_ = m.render_rays_vjp(ro, rd, near, far, g["cot"], z_fine=zf_ref)
        for a, b, what in ((cpu(ro_), g[tag + "_grad_rays"][0], "grad_o"), (cpu(rd_), g[tag + "_grad_rays"][1], "grad_d")):
            e = _rel_rows(a, b)
            print(tag, what, "vs the reference's autograd at its depths: median %.2e  90 %% %.2e  max %.2e" % (np.median(e), np.percentile(e, 90), e.max()))
            assert np.percentile(e, 90) < 3e-4 and np.linalg.norm(a - b) / np.linalg.norm(b) < 2e-2, (what, np.percentile(e, 90), e.max())
    # ---- chunking never changes a result: the smallest workspace the library accepts (64-ray chunks), ragged ray counts
    os.environ["NSR_WIDE_WORKSPACE_GB"] = "0.0001"
    try:
        m2 = WideModel(sd_c, sd_f if fine else None, n_samples=ns, n_importance=ni)
        for k in (n, 1, 33):
            q = m2.render_rays(np.tile(ro, (3, 1))[:k + 96 if k == n else k], np.tile(rd, (3, 1))[:k + 96 if k == n else k], near, far)
            kk = min(k, n)
            for key in ("rgb_map", "disp_map", "acc_map") + (("rgb0", "z_std") if fine else ()):
                assert np.array_equal(cpu(q[key])[:kk], cpu(r[key])[:kk], equal_nan=True), (key, k)
        q = m2.render_rays(np.tile(ro, (3, 1)), np.tile(rd, (3, 1)), near, far)
        assert m2.last_kernel_ms()[1] == 3 and np.array_equal(cpu(q["rgb_map"])[2 * n:], cpu(r["rgb_map"]), equal_nan=True)
        g2o, g2d = m2.render_rays_vjp(np.tile(ro, (3, 1)), np.tile(rd, (3, 1)), near, far, np.tile(g["cot"], (3, 1)))
        assert np.array_equal(cpu(g2o)[n:2 * n], cpu(go)) and np.array_equal(cpu(g2d)[2 * n:], cpu(gd))
    finally:
        del os.environ["NSR_WIDE_WORKSPACE_GB"]


@pytest.mark.parametrize("mlp", ["bf16x3"])
def test_layered_renderer_chunking_at_scale(mlp, oracle):
    """20 000 rays of the 10 x 384 network (g25 a's weights) with a 16 GiB workspace (one or two chunks) and with a 0.5 GiB one
    (dozens of chunks of other sizes, a ragged last one): forward and input gradient bit for bit -- a ray's result depends on
    nothing but the ray, whatever the rows of the GEMMs it shares a launch with."""
    from neural_sim_nerf_amd.wide import WideModel
    g = load_golden("g25_wide_networks")
    sd_c, sd_f, ns, ni = wide_case(oracle, g, "a")
    near, far = oracle.YCBV_NEAR, oracle.YCBV_FAR
    K = oracle.scaled_K(400.0 / 142)
    pose = np.asarray(oracle.sweep_poses(1, seed=4))[0]
    ro, rd = (a.reshape(-1, 3)[:20000] for a in oracle.get_rays(142, 142, K, pose[:3, :4]))
    cot = np.random.RandomState(0).standard_normal((len(ro), 3)).astype(np.float32)
    res = []
    for gb in ("16", "0.5"):
        os.environ["NSR_WIDE_WORKSPACE_GB"] = gb
        try:
            m = WideModel(sd_c, sd_f, n_samples=ns, n_importance=ni)
            r = m.render_rays(ro, rd, near, far)
            chunks_f = m.last_kernel_ms()[1]
            go, gd = m.render_rays_vjp(ro, rd, near, far, cot)
            res.append((cpu(r["rgb_map"]), cpu(r["disp_map"]), cpu(r["z_std"]), cpu(go), cpu(gd), chunks_f, m.last_kernel_ms()[1]))
            m.close()
        finally:
            del os.environ["NSR_WIDE_WORKSPACE_GB"]
    a, b = res
    print("chunks forward / gradient: %d / %d with 16 GiB, %d / %d with 0.5 GiB" % (a[5], a[6], b[5], b[6]))
    assert b[5] > 4 * a[5] and b[6] > b[5]
    for x, y in zip(a[:5], b[:5]):
        assert np.array_equal(x, y, equal_nan=True)
    assert np.isfinite(a[0]).all() and np.isfinite(a[3]).all() and np.abs(a[3]).max() > 0


@pytest.mark.parametrize("mlp", MLPS3)
def test_layered_renderer_nan_rays_stay_nan_and_alone(mlp, oracle):
    """A ray with a NaN origin or an infinite direction renders to NaN (pts -> encodings -> every layer -> relu(sigma), F.relu
    keeps NaN, RN:356 -> alpha -> weights -> pixel, and disp through torch.max, RN:381), as it does in the reference -- and it
    is the ONLY ray that does: its neighbours in the same GEMM tiles keep their bits.  (bf16x3: a NaN splits into NaN pieces, an
    infinity into inf + NaN: the row is NaN like on the fp32 MFMAs; rows do not mix in a GEMM.)"""
    WideModel = _wide(mlp)
    g = load_golden("g25_wide_networks")
    sd_c, sd_f, ns, ni = wide_case(oracle, g, "b")
    near, far = oracle.YCBV_NEAR, oracle.YCBV_FAR
    ro, rd = g["rays_o"].copy(), g["rays_d"].copy()
    m = WideModel(sd_c, sd_f, n_samples=ns, n_importance=ni)
    clean = m.render_rays(ro, rd, near, far)
    ro[5, 1] = np.nan
    rd[17, 0] = np.inf
    bad = m.render_rays(ro, rd, near, far)
    go, gd = m.render_rays_vjp(ro, rd, near, far, g["cot"])
    ok = np.ones(len(ro), bool)
    ok[[5, 17]] = False
    for k in ("rgb_map", "disp_map", "acc_map", "rgb0", "acc0"):
        a, b = cpu(bad[k]), cpu(clean[k])
        assert np.isnan(a[~ok]).all(), k
        if mlp == "f16x2":      # the infinite ray leaves fp16's range: its passes are run again on bf16x3, whose bits the neighbours then carry
            fin = np.isfinite(b[ok])                       # (disp of an empty ray is 0 / 0 in the reference too, RN:381)
            assert np.array_equal(np.isfinite(a[ok]), fin), k
            dlt = np.abs(a[ok][fin] - b[ok][fin])
            assert (dlt > 1e-5).mean() < 0.02 and dlt.max() < 5e-3, (k, dlt.max())
        else:
            assert np.array_equal(a[ok], b[ok], equal_nan=True), k
    if mlp == "f16x2":
        st = m.range_status()
        assert st["passes_rerun"] >= 1 and st["passes"] > st["passes_rerun"], st
    assert np.isnan(cpu(go)[~ok]).all() and np.isfinite(cpu(go)[ok]).all() and np.isfinite(cpu(gd)[ok]).all()
    m.close()


@pytest.mark.parametrize("mlp", MLPS3)
def test_layered_renderer_options(mlp, oracle):
    """The per-ray options on the layered renderer (g25 b: two skips, (48, 100) samples): stratified depths, random uniforms,
    density noise, per-ray bounds, given view directions, white background + lindisp -- stage-wise against the oracle on the
    renderer's own intermediates, the gradient (incl. dL/d viewdirs) against the oracle's."""
    WideModel = _wide(mlp)
    g = load_golden("g25_wide_networks")
    sd_c, sd_f, ns, ni = wide_case(oracle, g, "b")
    ro, rd = g["rays_o"], g["rays_d"]
    n = len(ro)
    rng = np.random.RandomState(5)
    near = (oracle.YCBV_NEAR * (1.0 + 0.1 * rng.rand(n))).astype(np.float32)
    far = (oracle.YCBV_FAR * (1.0 - 0.1 * rng.rand(n))).astype(np.float32)
    vdirs = rng.standard_normal((n, 3)).astype(np.float32)
    vdirs /= np.linalg.norm(vdirs, axis=-1, keepdims=True)
    rnd = dict(t_rand=rng.rand(n, ns).astype(np.float32), u=rng.rand(n, ni).astype(np.float32),
               noise0=(0.3 * rng.standard_normal((n, ns))).astype(np.float32),
               noise1=(0.3 * rng.standard_normal((n, ns + ni))).astype(np.float32))
    for white, lindisp in ((False, False), (True, True)):
        m = WideModel(sd_c, sd_f, n_samples=ns, n_importance=ni, white_bkgd=white, lindisp=lindisp)
        q = m.render_rays(ro, rd, 0.0, 0.0, debug=True, extras=dict(rnd, near=near, far=far, viewdirs=vdirs))
        z = oracle.perturb_z(oracle.coarse_z(near, far, n=ns, lindisp=lindisp), rnd["t_rand"])
        raw0 = oracle.run_network(sd_c, (ro[:, None] + rd[:, None] * z[..., None]).astype(np.float32), vdirs)
        assert_close(cpu(q["raw0"]), raw0, atol=5e-5, rtol=5e-5, what="coarse raw with options")
        rgb0, _, _, w0, _ = oracle.raw2outputs(cpu(q["raw0"]), z, rd, white, rnd["noise0"])
        assert_close(cpu(q["weights0"]), w0, atol=2e-6, what="weights0 with options")
        assert_close(cpu(q["rgb0"]), rgb0, atol=3e-6, what="rgb0 with options")
        z_mid = (np.float32(0.5) * (z[:, 1:] + z[:, :-1])).astype(np.float32)
        s, inds, _ = oracle.sample_pdf(z_mid, cpu(q["weights0"])[:, 1:-1], ni, u=rnd["u"])
        assert np.array_equal(cpu(q["inds"]), inds) and np.array_equal(cpu(q["z_samples"]), s)
        zf = np.sort(np.concatenate([z, s], -1), -1)            # the random uniforms leave z_samples unsorted: a real sort
        assert np.array_equal(cpu(q["z_fine"]), zf)
        raw = oracle.run_network(sd_f, (ro[:, None] + rd[:, None] * zf[..., None]).astype(np.float32), vdirs)
        assert_close(cpu(q["raw"]), raw, atol=5e-5, rtol=5e-5, what="fine raw with options")
        rgb, _, acc, _, _ = oracle.raw2outputs(cpu(q["raw"]), zf, rd, white, rnd["noise1"])
        assert_close(cpu(q["rgb_map"]), rgb, atol=3e-6, what="rgb with options")
        assert_close(cpu(q["acc_map"]), acc, atol=3e-6, what="acc with options")
        go, gd, gv = m.render_rays_vjp(ro, rd, 0.0, 0.0, g["cot"], extras=dict(rnd, near=near, far=far, viewdirs=vdirs))
        wo, wd, _, wv = oracle.render_rays_vjp(sd_c, sd_f, ro, rd, near, far, g["cot"], n_samples=ns, n_importance=ni, z_fine=zf,
                                                white_bkgd=white, lindisp=lindisp, viewdirs=vdirs, noise1=rnd["noise1"])
        for a, b, what in ((cpu(go), wo, "grad_o"), (cpu(gd), wd, "grad_d"), (cpu(gv), wv, "grad_viewdirs")):
            e = _rel_rows(a, b)
            assert np.percentile(e, 90) < 1e-4 and np.linalg.norm(a - b) / np.linalg.norm(b) < 1e-2, (what, np.percentile(e, 90), e.max())


@pytest.mark.parametrize("mlp", MLPS3)
def test_layered_and_fused_renderers_agree_on_the_fused_kernels_network(mlp, synth_nets, oracle):
    """The YCB-V network (8 x 256, 64 + 128 samples) through BOTH renderers: two independent implementations of the same path
    -- fused f16x2 kernels with the network in registers / LDS, and GEMMs layer by layer through HBM (fp32 MFMAs / bf16x3) -- agree
    on every ray within the end-to-end rule (census against the oracle for each; between them: PSNR > 60 dB, at most 2 % of the rays
    beyond 1e-4 -- two renders that each sit on their own side of a resampling discontinuity), and on the input gradient."""
    from neural_sim_nerf_amd.engine import NsrModel
    WideModel = _wide(mlp)
    C = _census()
    near, far = oracle.YCBV_NEAR, oracle.YCBV_FAR
    K = oracle.scaled_K(40.0)
    pose = np.asarray(oracle.sweep_poses(1, seed=2))[0]
    ro, rd = (a.reshape(-1, 3) for a in oracle.get_rays(40, 40, K, pose[:3, :4]))
    f = NsrModel(synth_nets[0], synth_nets[1])
    w = WideModel(synth_nets[0], synth_nets[1])
    a = f.render_views(pose, 40, 40, K, near, far)
    b = w.render_views(pose, 40, 40, K, near, far, debug=True)
    d = np.abs(cpu(a["rgb_map"]) - cpu(b["rgb_map"])).max(-1)
    print("fused vs layered: rays beyond 1e-4: %d of %d, PSNR %.1f dB" % ((d > 1e-4).sum(), d.size, oracle.psnr(cpu(a["rgb_map"]), cpu(b["rgb_map"]))))
    assert (d > 1e-4).mean() < 0.02 and oracle.psnr(cpu(a["rgb_map"]), cpu(b["rgb_map"])) > 60.0
    vd = oracle.normalize_dirs(rd)
    ref = oracle.render_rays(synth_nets[0], synth_nets[1], ro, rd, vd, near, far, extras=True)
    taps = ("rgb_map", "acc_map", "disp_map", "rgb0", "acc0", "raw0", "weights0", "inds", "z_samples", "z_fine", "raw")
    c = C.census(synth_nets, ro, rd, near, far, {k: cpu(b[k]) for k in taps}, ref)
    print("layered census:", {k: c[k] for k in ("rays", "rays_above_tol", "unattributed", "psnr_delta_db")})
    assert C.passes(c) and c["psnr_delta_db"] <= 0.1, c
    cot = np.random.RandomState(3).standard_normal((len(ro), 3)).astype(np.float32)
    fo, fd = f.render_rays_vjp(ro, rd, near, far, cot)
    wo, wd = w.render_rays_vjp(ro, rd, near, far, cot)
    for x, y in ((cpu(fo), cpu(wo)), (cpu(fd), cpu(wd))):
        assert np.median(_rel_rows(x, y)) < 2e-4 and np.linalg.norm(x - y) / np.linalg.norm(y) < 5e-2


def test_dropin_api_serves_networks_beyond_the_fused_kernels(oracle, tmp_path):
    """create_nerf-style modules of a shape the fused kernels cannot hold (10 x 384) and sample counts they are not built for
    go through render / render_path / render_path_grad unchanged: which renderer took the call, render() against the
    reference's pixels (g25 a) and its autograd against the reference's, render_path == render per pose, render_path_grad's
    per-patch dL/dpsi == autograd through render(rays=...) patch by patch (RN:168-181), retraw in the reference's shape,
    NeRF.forward on the reference's embedded input."""
    import torch
    import neural_sim_nerf_amd.run_nerf_noscale as R
    C = _census()
    g = load_golden("g25_wide_networks")
    sd_c, sd_f, ns, ni = wide_case(oracle, g, "a")
    near, far = oracle.YCBV_NEAR, oracle.YCBV_FAR
    D, W = oracle.net_shape(sd_c)[:2]
    nets = []
    for sd in (sd_c, sd_f):
        net = R.NeRF(D=D, W=W, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        nets.append(net.to(R.device))
    kw = dict(network_query_fn=None, perturb=False, N_importance=ni, network_fine=nets[1], N_samples=ns, network_fn=nets[0],
              use_viewdirs=True, white_bkgd=False, raw_noise_std=0., ndc=False, lindisp=False, near=near, far=far)
    m = R._model_for(nets[0], nets[1], ni, kw)
    assert m.mlp.startswith("layered-") and ("netdepth" in m.why_layered or "netwidth" in m.why_layered)
    rays = torch.stack([torch.from_numpy(g["rays_o"]), torch.from_numpy(g["rays_d"])], 0).to(R.device).requires_grad_(True)
    rgb, disp, acc, ex = R.render(400, 400, oracle.YCBV_K, chunk=512, rays=rays, retraw=True, **kw)
    assert ex["raw"].shape == (len(g["rays_o"]), ns + ni, 4) and set(ex) == {"rgb0", "disp0", "acc0", "z_std", "raw"}
    assert C.psnr_delta(cpu(rgb), g["a_rgb"]) <= 0.1 and (np.abs(cpu(rgb) - g["a_rgb"]).max(-1) > 1e-4).mean() <= 0.1
    assert_close(cpu(ex["rgb0"]), g["a_rgb0"], atol=1e-5, what="rgb0 vs reference")
    (gr,) = torch.autograd.grad(rgb, rays, grad_outputs=torch.from_numpy(g["cot"]).to(R.device))
    assert np.linalg.norm(cpu(gr) - g["a_grad_rays"]) / np.linalg.norm(g["a_grad_rays"]) < 5e-2
    # NeRF.forward on the reference's [P, 63 + 27] embedded tensor (only the raw coordinates are read)
    x = np.concatenate([oracle.embed(g["pts"], 10), oracle.embed(g["dirs"], 4)], -1)
    out = nets[0](torch.from_numpy(x).to(R.device))
    assert_close(cpu(out), g["a_net_out"][:, :4], atol=1e-5 + 2e-6 * np.abs(g["a_net_out"]).max(), rtol=2e-6, what="NeRF.forward")
    # render_path / render_path_grad at odd sample counts on a small image, two poses
    kw2 = dict(kw, N_samples=40, N_importance=72)
    assert "N_samples=40" in R._layered_why(nets[0], nets[1], 40, 72) or nets[0].fused_why_not
    Hs, Ks = 12, oracle.scaled_K(400.0 / 12)
    hwf = [Hs, Hs, Ks[0][0]]
    poses = torch.tensor(np.asarray(oracle.sweep_poses(2, seed=5)), dtype=torch.float32)
    rgbs, disps = R.render_path(None, poses, hwf, Ks, 32, kw2, savedir=str(tmp_path), object_id=2)
    assert rgbs.shape == (2, Hs, Hs, 3) and sorted(os.listdir(tmp_path / "2")) == ["000.png", "001.png"]
    for i in range(2):
        one = R.render(Hs, Hs, Ks, chunk=32, c2w=poses[i, :3, :4], **kw2)
        assert np.array_equal(cpu(one[0]), rgbs[i]) and np.array_equal(cpu(one[1]), disps[i], equal_nan=True)
    Dm = torch.tensor(np.random.RandomState(3).standard_normal((8, 4, 4)).astype(np.float32) * 0.05)
    Dm[:, 3] = 0
    prob = torch.full((8,), 0.125, requires_grad=True)          # a differentiable path psi -> poses (as LL:202-247 provides)
    gposes = [p + (prob[:, None, None] * Dm).sum(0) * (1.0 + 0.5 * i) for i, p in enumerate(poses)]
    rng = np.random.RandomState(1)
    gE = [{"grad_E": [torch.from_numpy(rng.standard_normal((3, Hs, Hs)).astype(np.float32))]} for _ in range(2)]
    rgbs_g, dl = R.render_path_grad(prob, gposes, hwf, Ks, 32, gE, kw2, savedir=None)
    n_patches = (Hs * Hs + 31) // 32
    assert len(dl) == 2 * n_patches and rgbs_g.shape == (2, Hs, Hs, 3)
    for i in (0, 1):
        c2w = gposes[i][:3, :4]
        ro, rdd = R.get_rays(Hs, Hs, Ks, c2w)
        cot = gE[i]["grad_E"][0].permute(1, 2, 0).reshape(-1, 3)
        for p in (0, n_patches - 1):
            sl = slice(32 * p, min(32 * (p + 1), Hs * Hs))
            batch = torch.stack([ro.reshape(-1, 3)[sl], rdd.reshape(-1, 3)[sl]], 0)
            rgb_p = R.render(Hs, Hs, Ks, chunk=32, rays=batch, **kw2)[0]
            (gb,) = torch.autograd.grad(rgb_p, batch, grad_outputs=cot[sl].to(rgb_p.device), retain_graph=True)
            (gp,) = torch.autograd.grad(batch, prob, grad_outputs=gb.to(batch.device), retain_graph=True)
            want = gp.detach().cpu().numpy()
            got = dl[i * n_patches + p].numpy()
            assert np.abs(got - want).max() <= 1e-3 * np.abs(want).max() + 1e-9, (i, p, got, want)


@pytest.mark.parametrize("mlp", MLPS3)
def test_c_host_of_the_layered_renderer(mlp, tmp_path, oracle):
    """The layered renderer's boundary is a C ABI too: examples/c_host_wide.c (plain C + the HIP runtime C API +
    include/nsr_wide.h; no Python, no torch) describes the 6 x 300 two-skip network of g25 b, uploads the parameters in the
    modules' own layout, sizes and allocates its workspace itself and renders the 48 golden rays forward and with the input
    gradient at (48, 100) samples -- every output equals the Python mirror's (wide.WideModel), bit for bit, also when its
    workspace holds 64 rays only... and when it holds all of them."""
    import shutil
    import subprocess
    from neural_sim_nerf_amd.wide import describe, FLAG_MLP_BF16X3, FLAG_MLP_F16X2
    WideModel = _wide(mlp)
    if shutil.which("gcc") is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("needs gcc and the ROCm headers")
    import torch
    csrc = os.path.join(ROOT, "neural_sim_nerf_amd", "csrc")
    exe = str(tmp_path / "c_host_wide")
    subprocess.check_call(["gcc", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_host_wide.c"), "-L" + csrc, "-lnsr", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    g = load_golden("g25_wide_networks")
    sd_c, sd_f, ns, ni = wide_case(oracle, g, "b")
    ro, rd, cot = np.tile(g["rays_o"], (3, 1)), np.tile(g["rays_d"], (3, 1)), np.tile(g["cot"], (3, 1))
    n = len(ro)
    near, far = np.float32(oracle.YCBV_NEAR), np.float32(oracle.YCBV_FAR)
    net, flat_c = describe(sd_c)
    _, flat_f = describe(sd_f)
    m = WideModel(sd_c, sd_f, n_samples=ns, n_importance=ni)
    want = m.render_rays(ro, rd, float(near), float(far))
    go, gd = m.render_rays_vjp(ro, rd, float(near), float(far), cot)
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    for ws_rays in (64, n):
        hd = np.zeros(32, np.int32)
        hd[:7] = [net.D, net.W, net.multires, net.multires_views, net.use_viewdirs, net.output_ch, net.n_skips]
        hd[7:23] = list(net.skips)
        hd[23:28] = [ns, ni, n, ws_rays, {"bf16x3": FLAG_MLP_BF16X3, "f16x2": FLAG_MLP_F16X2}.get(mlp, 0)]
        with open(str(tmp_path / "in.bin"), "wb") as f:
            f.write(hd.tobytes())
            for a in (flat_c, flat_f, torch.linspace(0., 1., ns).numpy(), torch.linspace(0., 1., ni).numpy(), ro, rd, cot,
                      np.array([near, far], np.float32)):
                f.write(np.ascontiguousarray(a, np.float32).tobytes())
        r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=120, env=env)
        assert r.returncode == 0, r.stderr
        print(r.stdout.strip())
        got = np.fromfile(str(tmp_path / "out.bin"), np.float32)
        off = 0
        for key, width in (("rgb_map", 3), ("disp_map", 1), ("acc_map", 1), ("rgb0", 3), ("disp0", 1), ("acc0", 1), ("z_std", 1)):
            assert np.array_equal(got[off:off + width * n], cpu(want[key]).reshape(-1), equal_nan=True), (ws_rays, key)
            off += width * n
        assert np.array_equal(got[off:off + 3 * n], cpu(go).reshape(-1)) and np.array_equal(got[off + 3 * n:], cpu(gd).reshape(-1)), ws_rays
    m.close()



# ---- r06: edge shapes, hipGraph capture, two streams, the bounds-checked build -------------------------
def _edge_case(oracle):
    """a 2 x 34 network (pad32 edge: 34 -> 64 columns, of which 30 are padding; W / 2 = 17), skip after layer 0, 2 + 1 frequencies,
    at N_samples = 3 (ONE interior weight: the shortest row sample_pdf accepts) and N_importance = 512 (the longest)"""
    sd_c = oracle.synth_weights_shape(11, 2, 34, 2, 1, [0], True)
    sd_f = oracle.synth_weights_shape(12, 2, 34, 2, 1, [0], True)
    K = oracle.scaled_K(50.0)
    pose = np.asarray(oracle.sweep_poses(1, seed=2))[0]
    ro, rd = (a.reshape(-1, 3) for a in oracle.get_rays(8, 8, K, pose[:3, :4]))
    return sd_c, sd_f, np.ascontiguousarray(ro[:61]), np.ascontiguousarray(rd[:61])


@pytest.mark.parametrize("mlp", MLPS3)
def test_layered_renderer_edge_shapes(mlp, oracle):
    """W = 34, N_samples = 3, N_importance = 512, 61 rays (a ragged last GEMM tile at every tile height), and run_network on
    1 / 127 / 129 / 257 points of two networks: stage by stage against the oracle, indices and samples bit for bit, the input
    gradient against the oracle's fp64 backprop."""
    WideModel = _wide(mlp)
    sd_c, sd_f, ro, rd = _edge_case(oracle)
    n = len(ro)
    near, far = oracle.YCBV_NEAR, oracle.YCBV_FAR
    vd = oracle.normalize_dirs(rd)
    m = WideModel(sd_c, sd_f, n_samples=3, n_importance=512)
    r = m.render_rays(ro, rd, near, far, debug=True)
    z = oracle.coarse_z(np.full(n, near, np.float32), np.full(n, far, np.float32), n=3)
    raw0 = oracle.run_network(sd_c, (ro[:, None] + rd[:, None] * z[..., None]).astype(np.float32), vd)
    assert_close(cpu(r["raw0"]), raw0, atol=5e-5, rtol=5e-5, what="coarse raw")
    _, _, _, w0, _ = oracle.raw2outputs(cpu(r["raw0"]), z, rd)
    assert_close(cpu(r["weights0"]), w0, atol=2e-6, what="weights0 | own raw")
    z_mid = (np.float32(0.5) * (z[:, 1:] + z[:, :-1])).astype(np.float32)
    s, inds, _ = oracle.sample_pdf(z_mid, cpu(r["weights0"])[:, 1:-1], 512)
    assert np.array_equal(cpu(r["inds"]), inds) and np.array_equal(cpu(r["z_samples"]), s)
    zf = np.sort(np.concatenate([z, s], -1), -1)
    assert np.array_equal(cpu(r["z_fine"]), zf)
    raw = oracle.run_network(sd_f, (ro[:, None] + rd[:, None] * zf[..., None]).astype(np.float32), vd)
    assert_close(cpu(r["raw"]), raw, atol=5e-5, rtol=5e-5, what="fine raw | own z")
    rgb, _, acc, _, _ = oracle.raw2outputs(cpu(r["raw"]), zf, rd)
    assert_close(cpu(r["rgb_map"]), rgb, atol=3e-6, what="rgb | own raw")
    assert_close(cpu(r["acc_map"]), acc, atol=3e-6, what="acc | own raw")
    cot = np.random.RandomState(2).standard_normal((n, 3)).astype(np.float32)
    go, gd = m.render_rays_vjp(ro, rd, near, far, cot)
    wo, wd, _ = oracle.render_rays_vjp(sd_c, sd_f, ro, rd, near, far, cot, n_samples=3, n_importance=512, z_fine=zf)
    for a, b, what in ((cpu(go), wo, "grad_o"), (cpu(gd), wd, "grad_d")):
        e = _rel_rows(a, b)
        assert np.percentile(e, 90) < 1e-4 and np.linalg.norm(a - b) / np.linalg.norm(b) < 1e-2, (what, np.percentile(e, 90), e.max())
    # ragged M of the GEMMs themselves: P points, one row more / less than a tile
    g = load_golden("g25_wide_networks")
    big_c, _, ns, ni = wide_case(oracle, g, "a")                         # 10 x 384: 256- + 128-column tiles
    mb = WideModel(big_c, None, n_samples=ns, n_importance=0)
    rng = np.random.RandomState(7)
    for P in (1, 127, 129, 257):
        pts = (rng.rand(P, 3).astype(np.float32) - 0.5) * 0.4
        dirs = oracle.normalize_dirs(rng.standard_normal((P, 3)).astype(np.float32))
        for mm, sd in ((m, sd_c), (mb, big_c)):
            want = oracle.run_network(sd, pts[:, None], dirs)[:, 0]
            got = cpu(mm.run_network(pts, dirs, 0))
            assert_close(got, want, atol=5e-5, rtol=5e-5, what="run_network on %d points" % P)
    m.close()
    mb.close()


@pytest.mark.parametrize("mlp", MLPS3)
def test_layered_launch_is_graph_capturable_and_replays_bit_identically(mlp, oracle):
    """include/nsr_wide.h: the launch calls only enqueue work (kernels, memset / memcpy nodes; the timing events are skipped
    under capture), so nsrw_render_rays and nsrw_render_rays_vjp can be captured into a hipGraph: the replay equals the eager
    launch bit for bit, also after the ray buffers the graph reads have been rewritten in place."""
    import torch
    WideModel = _wide(mlp)
    g = load_golden("g25_wide_networks")
    sd_c, sd_f, ns, ni = wide_case(oracle, g, "b")
    near, far = oracle.YCBV_NEAR, oracle.YCBV_FAR
    m = WideModel(sd_c, sd_f, n_samples=ns, n_importance=ni)
    dev = m.device
    ro, rd, cot = (torch.as_tensor(np.ascontiguousarray(x), device=dev) for x in (g["rays_o"], g["rays_d"], g["cot"]))
    ro2, rd2 = ro.flip(0).contiguous(), (rd.flip(0) * 1.01).contiguous()
    eager = []
    for a, b in ((ro, rd), (ro2, rd2)):
        r = m.render_rays(a, b, near, far)
        go, gd = m.render_rays_vjp(a, b, near, far, cot)              # (also sizes the shared workspace for the gradient)
        eager.append([cpu(r[k]) for k in ("rgb_map", "disp_map", "acc_map", "z_std")] + [cpu(go), cpu(gd)])
    bo, bd = ro.clone(), rd.clone()
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        m.render_rays_vjp(bo, bd, near, far, cot)                      # this stream's workspace exists before the capture
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            r = m.render_rays(bo, bd, near, far)
            go, gd = m.render_rays_vjp(bo, bd, near, far, cot)
    with pytest.raises(Exception, match="not timed|no timed launch"):
        m.last_kernel_ms()
    for it, (a, b) in enumerate(((ro, rd), (ro2, rd2), (ro, rd))):
        bo.copy_(a)
        bd.copy_(b)
        graph.replay()
        torch.cuda.synchronize(dev)
        got = [cpu(r[k]) for k in ("rgb_map", "disp_map", "acc_map", "z_std")] + [cpu(go), cpu(gd)]
        for x, y in zip(got, eager[it % 2]):
            assert np.array_equal(x, y, equal_nan=True), it
    m.close()


def test_layered_f16x2_pass_that_leaves_the_fp16_range_is_rerun_on_bf16x3(oracle):
    """NSRW_FLAG_MLP_F16X2's safety net.  A coarse network whose layer-2 bias sends one hidden unit to 7e4 on every point: the
    activations entering layer 3 leave fp16's range (65504), kw_gemm_h2 raises the flag, and the whole coarse pass runs again on
    bf16x3 inside the same call -- the coarse outputs, the resampling indices and the fine depths are then the bf16x3 handle's BIT
    FOR BIT; the fine pass (in range) stays on fp16 MFMAs.  Counted per pass: one re-run per chunk of rays, none dropped; and the
    call stays capturable (no host round trip: the re-run's launches are conditional on a device flag)."""
    import torch
    g = load_golden("g25_wide_networks")
    sd_c, sd_f, ns, ni = wide_case(oracle, g, "b")
    sd_c = {k: v.copy() for k, v in sd_c.items()}
    sd_c["pts_linears.2.bias"][5] += 7.0e4
    near, far = oracle.YCBV_NEAR, oracle.YCBV_FAR
    ro, rd, cot = np.tile(g["rays_o"], (4, 1)), np.tile(g["rays_d"], (4, 1)), np.tile(g["cot"], (4, 1))
    os.environ["NSR_WIDE_WORKSPACE_GB"] = "0.0001"        # the smallest workspace the library accepts: chunks of 64 rays
    try:
        h2, b3 = (_wide(mlp)(sd_c, sd_f, n_samples=ns, n_importance=ni) for mlp in ("f16x2", "bf16x3"))
        a, b = h2.render_rays(ro, rd, near, far, debug=True), b3.render_rays(ro, rd, near, far, debug=True)
        chunks = h2.last_kernel_ms()[1]
        st = h2.range_status()
        assert chunks >= 2 and st["passes"] == 2 * chunks and st["passes_rerun"] == chunks and st["dropped_items"] == 0, (chunks, st)
        for k in ("rgb0", "acc0", "weights0", "raw0", "inds", "z_samples", "z_fine"):
            assert np.array_equal(cpu(a[k]), cpu(b[k]), equal_nan=True), k
        assert np.isfinite(cpu(a["rgb_map"])).all() and not np.array_equal(cpu(a["raw"]), cpu(b["raw"]))      # the fine pass ran on fp16 MFMAs
        assert_close(cpu(a["raw"]), cpu(b["raw"]), atol=5e-5, rtol=5e-5, what="fine raw: f16x2 vs bf16x3 at the same depths")
        assert_close(cpu(a["rgb_map"]), cpu(b["rgb_map"]), atol=3e-6, what="rgb: f16x2 vs bf16x3 at the same depths")
        # against the oracle, like any other network
        vd = oracle.normalize_dirs(rd)
        z = oracle.coarse_z(np.full(len(ro), near, np.float32), np.full(len(ro), far, np.float32), n=ns)
        raw0 = oracle.run_network(sd_c, (ro[:, None] + rd[:, None] * z[..., None]).astype(np.float32), vd)
        assert_close(cpu(a["raw0"])[..., :4], raw0, atol=5e-5, rtol=5e-5, what="coarse raw of the re-run pass")
        # the gradient call: forward re-run as well, gradient GEMMs on bf16x3 -> the bf16x3 handle's numbers at the same depths
        go, gd = h2.render_rays_vjp(ro, rd, near, far, cot)
        wo, wd = b3.render_rays_vjp(ro, rd, near, far, cot)
        for x, y in ((cpu(go), cpu(wo)), (cpu(gd), cpu(wd))):
            assert np.isfinite(x).all() and np.percentile(_rel_rows(x, y), 90) < 1e-4
        # under capture
        dev = h2.device
        to, td = (torch.as_tensor(np.ascontiguousarray(x), device=dev) for x in (ro, rd))
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            h2.render_rays(to, td, near, far)
            torch.cuda.synchronize(dev)
            before = h2.range_status()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=s):
                r = h2.render_rays(to, td, near, far)
        graph.replay()
        torch.cuda.synchronize(dev)
        after = h2.range_status()
        assert after["passes_rerun"] - before["passes_rerun"] == chunks, (before, after)
        assert np.array_equal(cpu(r["rgb_map"]), cpu(a["rgb_map"]), equal_nan=True)
        h2.close()
        b3.close()
    finally:
        del os.environ["NSR_WIDE_WORKSPACE_GB"]


def test_layered_f16x2_gradient_over_twelve_orders_of_magnitude(oracle):
    """The gradient GEMMs of an f16x2 handle run on per-point NORMALISED gradients (kw_composite_bwd scales a point's dL/d raw to
    [2^7, 2^8), kw_embed_bwd undoes it: the chain is linear).  Cotangents between 1e-6 and 1e+6 per ray in ONE launch: per-ray
    error against the oracle's fp64 backprop at the same depths like the bf16x3 handle's; a zero cotangent gives exactly zero; a
    power-of-two multiple of the cotangent scales the gradient exactly; nothing is re-run."""
    g = load_golden("g25_wide_networks")
    sd_c, sd_f, ns, ni = wide_case(oracle, g, "b")
    near, far = oracle.YCBV_NEAR, oracle.YCBV_FAR
    ro, rd = g["rays_o"], g["rays_d"]
    rng = np.random.RandomState(12)
    amp = np.exp(rng.uniform(np.log(1e-6), np.log(1e6), (len(ro), 1))).astype(np.float32)
    cot = (g["cot"] * amp).astype(np.float32)
    cot[7] = 0.0
    h2, b3 = (_wide(mlp)(sd_c, sd_f, n_samples=ns, n_importance=ni) for mlp in ("f16x2", "bf16x3"))
    zf = cpu(b3.render_rays(ro, rd, near, far, debug=True)["z_fine"])
    wo, wd, _ = oracle.render_rays_vjp(sd_c, sd_f, ro, rd, near, far, cot, n_samples=ns, n_importance=ni, z_fine=zf)
    want = np.concatenate([wo, wd], 1)
    errs = {}
    for name, m in (("f16x2", h2), ("bf16x3", b3)):
        go, gd = m.render_rays_vjp(ro, rd, near, far, cot, z_fine=zf)
        got = np.concatenate([cpu(go), cpu(gd)], 1)
        assert np.isfinite(got).all() and not got[7].any(), name
        ok = np.arange(len(ro)) != 7
        errs[name] = _rel_rows(got[ok], want[ok])
        print("layered-%s, cotangents 1e-6 .. 1e+6: per-ray error median %.2e  90 %% %.2e  max %.2e" % (name, np.median(errs[name]), np.percentile(errs[name], 90), errs[name].max()))
    assert np.median(errs["f16x2"]) <= max(3.0 * np.median(errs["bf16x3"]), 2e-6) and np.percentile(errs["f16x2"], 90) < 1e-4, errs
    a1 = np.concatenate([cpu(x) for x in h2.render_rays_vjp(ro, rd, near, far, g["cot"], z_fine=zf)], 1)
    a2 = np.concatenate([cpu(x) for x in h2.render_rays_vjp(ro, rd, near, far, (g["cot"] * np.float32(2.0 ** -9)).astype(np.float32), z_fine=zf)], 1)
    assert np.array_equal(a2, a1 * np.float32(2.0 ** -9))
    assert h2.range_status()["passes_rerun"] == 0
    h2.close()
    b3.close()


@pytest.mark.parametrize("mlp", MLPS3)
def test_layered_renderer_on_random_network_shapes(mlp, oracle):
    """Sixteen seeded random shapes -- depth 1..11, width 34..600 (every mix of 256- / 128- / 64-column GEMM tiles and padded tails),
    random skip lists, 0..10 / 0..4 encoding frequencies, with and without view directions (4 or 5 output rows), random sample
    counts -- each rendered on 33 rays: network outputs against the oracle, compositing on the renderer's own raw, resampling
    indices / samples / merged depths BIT FOR BIT on its own coarse weights, the input gradient against the oracle's fp64
    backprop.  What the fixed cases (g25, the edge shapes) cannot enumerate."""
    WideModel = _wide(mlp)
    near, far = oracle.YCBV_NEAR, oracle.YCBV_FAR
    rng = np.random.RandomState(2026)
    K = oracle.scaled_K(40.0)
    pose = np.asarray(oracle.sweep_poses(1, seed=9))[0]
    ro_all, rd_all = (a.reshape(-1, 3) for a in oracle.get_rays(40, 40, K, pose[:3, :4]))
    for case in range(16):
        D = int(rng.randint(1, 12))
        W = int(rng.choice([34, 40, 72, 100, 136, 200, 264, 300, 392, 520, 600]))
        skips = sorted(int(x) for x in rng.choice(np.arange(max(D - 1, 1)), size=min(int(rng.randint(0, 3)), max(D - 1, 0)), replace=False)) if D > 1 else []
        L, Lv = int(rng.randint(0, 11)), int(rng.randint(0, 5))
        uv = bool(rng.randint(0, 2))
        oc = int(rng.choice([4, 5]))
        ns, ni = int(rng.randint(3, 41)), int(rng.choice([0, 1, 1]) * rng.randint(1, 41))
        sd_c = oracle.synth_weights_shape(100 + case, D, W, L, Lv, skips, uv, oc)
        sd_f = oracle.synth_weights_shape(200 + case, D, W, L, Lv, skips, uv, oc)
        tag = "case %d: %d x %d skips %s L %d Lv %d viewdirs %s oc %d samples %d + %d" % (case, D, W, skips, L, Lv, uv, oc, ns, ni)
        sel = rng.choice(len(ro_all), 33, replace=False)
        ro, rd = ro_all[sel], rd_all[sel]
        n = len(ro)
        vd = oracle.normalize_dirs(rd)
        m = WideModel(sd_c, sd_f if ni else None, n_samples=ns, n_importance=ni)
        r = m.render_rays(ro, rd, near, far, debug=True)
        z = oracle.coarse_z(np.full(n, near, np.float32), np.full(n, far, np.float32), n=ns)
        raw0 = oracle.run_network(sd_c, (ro[:, None] + rd[:, None] * z[..., None]).astype(np.float32), vd)
        k_raw0 = cpu(r["raw0"])                     # (coarse only: the last pass IS the coarse pass, same tap)
        scale = max(1.0, float(np.abs(raw0).max()))
        assert_close(k_raw0[..., :4], raw0[..., :4], atol=5e-5 * scale, rtol=5e-5, what=tag + ": coarse raw")
        _, _, _, w0, _ = oracle.raw2outputs(k_raw0[..., :4], z, rd)
        if ni:
            assert_close(cpu(r["weights0"]), w0, atol=2e-6, what=tag + ": weights0 | own raw")
            z_mid = (np.float32(0.5) * (z[:, 1:] + z[:, :-1])).astype(np.float32)
            s_, inds, _ = oracle.sample_pdf(z_mid, cpu(r["weights0"])[:, 1:-1], ni)
            assert np.array_equal(cpu(r["inds"]), inds) and np.array_equal(cpu(r["z_samples"]), s_), tag
            zf = np.sort(np.concatenate([z, s_], -1), -1)
            assert np.array_equal(cpu(r["z_fine"]), zf), tag
            raw = oracle.run_network(sd_f, (ro[:, None] + rd[:, None] * zf[..., None]).astype(np.float32), vd)
            k_raw = cpu(r["raw"])
            assert_close(k_raw[..., :4], raw[..., :4], atol=5e-5 * max(1.0, float(np.abs(raw).max())), rtol=5e-5, what=tag + ": fine raw | own z")
        else:
            zf, k_raw = z, k_raw0
        rgb, _, acc, _, _ = oracle.raw2outputs(k_raw[..., :4], zf, rd)
        assert_close(cpu(r["rgb_map"]), rgb, atol=3e-6, what=tag + ": rgb | own raw")
        assert_close(cpu(r["acc_map"]), acc, atol=3e-6, what=tag + ": acc | own raw")
        cot = rng.standard_normal((n, 3)).astype(np.float32)
        go, gd = m.render_rays_vjp(ro, rd, near, far, cot, z_fine=zf if ni else None)
        wo, wd, _ = oracle.render_rays_vjp(sd_c, sd_f if ni else None, ro, rd, near, far, cot, n_samples=ns, n_importance=ni, z_fine=zf)
        for a_, b_, what in ((cpu(go), wo, "grad_o"), (cpu(gd), wd, "grad_d")):
            e = _rel_rows(a_, b_)
            assert np.isfinite(a_).all() and np.percentile(e, 90) < 2e-4 and np.linalg.norm(a_ - b_) / max(np.linalg.norm(b_), 1e-20) < 1e-2, \
                (tag, what, np.percentile(e, 90), e.max())
        assert m.range_status()["passes_rerun"] == 0, tag
        m.close()


def test_two_layered_handles_on_two_streams(oracle):
    """One handle per (model, stream): two handles -- one per arithmetic -- launched back to back on two streams, each with
    its stream's own workspace, give what each gives alone."""
    import torch
    g = load_golden("g25_wide_networks")
    sd_c, sd_f, ns, ni = wide_case(oracle, g, "b")
    near, far = oracle.YCBV_NEAR, oracle.YCBV_FAR
    ro, rd = np.tile(g["rays_o"], (40, 1)), np.tile(g["rays_d"], (40, 1))
    ms = [_wide(mlp)(sd_c, sd_f, n_samples=ns, n_importance=ni) for mlp in MLPS]
    dev = ms[0].device
    alone = [cpu(m.render_rays(ro, rd, near, far)["rgb_map"]) for m in ms]
    torch.cuda.synchronize(dev)
    streams = [torch.cuda.Stream(device=dev) for _ in ms]
    outs = []
    for it in range(3):
        for m, st in zip(ms, streams):
            with torch.cuda.stream(st):
                outs.append(m.render_rays(ro, rd, near, far)["rgb_map"])
    torch.cuda.synchronize(dev)
    for i, o in enumerate(outs):
        assert np.array_equal(cpu(o), alone[i % 2], equal_nan=True), i
    from neural_sim_nerf_amd import wide
    assert len({k for k in wide._WORKSPACES if k[0] == dev.index}) >= 3          # the default stream's and the two streams'
    for m in ms:
        m.close()


def test_layered_debug_bounds_build_is_clean(tmp_path):
    """`make debug` compiles nsr_wide.hip with -DNSR_DEBUG_BOUNDS too (r06): every global / LDS index of the GEMM body (kw_gemm_h2 /
    _b3 / _f32) and of the per-ray kernels is checked against its extent.  A fresh process runs, on ALL THREE arithmetics, the edge shapes (W = 34,
    3 + 512 samples, 61 rays), ragged run_network calls, NaN / zero / infinite rays, a 64-ray-chunk workspace and the
    gradient; no check may trip, and every output equals the release build's."""
    import subprocess
    dbg = os.path.join(ROOT, "neural_sim_nerf_amd", "csrc", "libnsr_debug.so")
    if not os.path.exists(dbg):
        pytest.skip("libnsr_debug.so not built (make -C neural_sim_nerf_amd/csrc debug)")
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import nerf_oracle as O
from neural_sim_nerf_amd.wide import WideModel
from test_gpu_wide import _edge_case
from test_oracle_golden import wide_case
g = np.load(%r)
out = {}
res = []
for mlp in ("bf16x3", "fp32", "f16x2"):
    sd_c, sd_f, ro, rd = _edge_case(O)
    ro, rd = ro.copy(), rd.copy()
    ro[3] = np.nan; rd[5] = 0.0; rd[7] = np.inf
    cases = [(sd_c, sd_f, 3, 512, ro, rd)]
    for tag in "bcd":
        c, f, ns, ni = wide_case(O, g, tag)
        cases.append((c, f if ni else None, ns, ni, g["rays_o"], g["rays_d"]))
    for i, (c, f, ns, ni, o, d) in enumerate(cases):
        for gb in ("16", "0.0001"):
            os.environ["NSR_WIDE_WORKSPACE_GB"] = gb
            m = WideModel(c, f, n_samples=ns, n_importance=ni, mlp=mlp)
            r = m.render_rays(o, d, O.YCBV_NEAR, O.YCBV_FAR, debug=True)
            go, gd = m.render_rays_vjp(o, d, O.YCBV_NEAR, O.YCBV_FAR, np.ones((len(o), 3), np.float32))
            rn = m.run_network(np.zeros((129, 3), np.float32) + 0.01, np.tile(np.float32([0, 0, 1]), (129, 1)), 0)
            res += [r["rgb_map"].cpu().numpy(), r["z_fine" if ni else "z_coarse"].cpu().numpy(), go.cpu().numpy(), gd.cpu().numpy(), rn.cpu().numpy()]
            out["%%s_%%d_%%s" %% (mlp, i, gb)] = m.debug_bounds_status()
            m.close()
np.savez(sys.argv[1] + "/res.npz", *res)
print(out)
''' % (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden", "g25_wide_networks.npz"))
    res = {}
    for name, lib in (("debug", dbg), ("release", os.path.join(ROOT, "neural_sim_nerf_amd", "csrc", "libnsr.so"))):
        d = tmp_path / name
        d.mkdir()
        r = subprocess.run([sys.executable, "-c", code, str(d)], env=dict(os.environ, NSR_LIB_PATH=lib), capture_output=True, text=True,
                           timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        res[name] = eval(r.stdout.strip().splitlines()[-1])
    assert len(res["debug"]) == 24 and all(v == (True, 0) for v in res["debug"].values()), res["debug"]
    assert all(v == (False, 0) for v in res["release"].values()), res["release"]
    a, b = np.load(tmp_path / "debug" / "res.npz"), np.load(tmp_path / "release" / "res.npz")
    assert len(a.files) == len(b.files) == 120
    for k in a.files:
        assert np.array_equal(a[k], b[k], equal_nan=True), k
