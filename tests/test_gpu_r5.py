"""GPU tests added in round 5 (run with -m gpu on an MI355X), through the C ABI / the drop-in API:
  * BASELINE configs[2] in its N = 1 form: ONE 100-view render_path launch at 400x400 (16 M rays, 3.07 G samples: the first
    driver-run launch past 2^31 samples) -- views bit-equal to single-view renders, PNGs in pose order, nothing reported by
    the range safety net;
  * the 8-rank layout on one GPU: bench.py --gpus 8 --backend gloo --share-gpu for sweep100 and models21 (ranks_seen == 8);
  * render_path / render_path_grad with render_kwargs_train (perturb, raw_noise_std), pinned to the reference run with its
    own recorded draws (tests/golden/g21_path_options.npz)."""
import os

import numpy as np
import pytest

from conftest import assert_close, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def cpu(t):
    return t.detach().cpu().numpy()


def test_config3_hundred_views_in_one_launch(synth_nets, oracle, tmp_path):
    """BASELINE configs[2] on one GPU, through the drop-in render_path: 100 poses x 400 x 400 in ONE persistent launch
    (16 000 000 rays = 8 000 000 work items, 3 072 000 000 ray-samples).  Views 0, 57 and 99 equal single-view renders of
    the same poses bit for bit (a ray's result depends on nothing but the ray: RN:229's loop carries no state), the PNGs
    are 000.png .. 099.png in pose order with exactly the returned pixels, and the range safety net reports nothing."""
    import torch
    import neural_sim_nerf_amd.run_nerf_noscale as R
    from neural_sim_nerf_amd import png
    near, far = oracle.YCBV_NEAR, oracle.YCBV_FAR
    K = oracle.YCBV_K
    nets = []
    for sd in synth_nets:
        n = R.NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
        n.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        nets.append(n.to(R.device))
    kw = dict(network_query_fn=None, perturb=False, N_importance=128, network_fine=nets[1], N_samples=64,
              network_fn=nets[0], use_viewdirs=True, white_bkgd=False, raw_noise_std=0., ndc=False, lindisp=False,
              near=near, far=far)
    poses = torch.tensor(np.asarray(oracle.sweep_poses(100, seed=0)))
    rgbs, disps = R.render_path(None, poses, [400, 400, K[0][0]], K, 512, kw, savedir=str(tmp_path), object_id=2)
    assert rgbs.shape == (100, 400, 400, 3) and disps.shape == (100, 400, 400) and rgbs.dtype == np.float32
    assert np.isfinite(rgbs).all()
    m = R._model_for(nets[0], nets[1], 128, kw)
    assert m.mlp == "f16x2" and m.range_status() == dict(last_items=0, points=0, rays=0, dropped_items=0)
    for i in (0, 57, 99):
        one = m.render_views(poses[i], 400, 400, K, near, far)
        assert np.array_equal(cpu(one["rgb_map"]).reshape(400, 400, 3), rgbs[i]), i
        assert np.array_equal(cpu(one["disp_map"]).reshape(400, 400), disps[i], equal_nan=True), i
    assert not np.array_equal(rgbs[0], rgbs[57])
    files = sorted(os.listdir(tmp_path / "2"))
    assert files == ["%03d.png" % i for i in range(100)]
    for i in (0, 57, 99):
        assert np.array_equal(png.imread(str(tmp_path / "2" / ("%03d.png" % i))), R.to8b(rgbs[i])), i


def _bench_json(args, timeout=900):
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                       env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                              # ONE JSON line on rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("workload", ["sweep100", "models21"])
def test_bench_eight_ranks_share_the_gpu(workload):
    """The driver's first 8-GPU run, rehearsed on one GPU: `bench.py --gpus 8 --backend gloo --share-gpu` -- the self-launch
    under torch.distributed.run, EIGHT ranks, the 8-way sharding of views (16 views -> 2 per rank) and models (21 ->
    3,3,3,3,3,2,2,2), the collectives at the outer-loop boundary (gloo stages them through the host: RCCL refuses two ranks
    on one device) and the ONE JSON line.  Not a scaling number; `ranks_seen == 8` says every rank took part."""
    args = ["--gpus", "8", "--backend", "gloo", "--share-gpu", "--workload", workload, "--steps", "1", "--warmup", "1",
            "--no-cpu-baseline", "--no-extras"] + (["--views", "16"] if workload == "sweep100" else [])
    d = _bench_json(args)
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and d["value"] > 0 and d["unit"] == "Mray-samples/s", d
    if workload == "sweep100":
        assert d["config"]["views"] == 16 and d["config"]["views_on_busiest_rank"] == 2 and d["scaling"] == "strong"
        assert d["ideal_speedup_over_1_gpu"] == 8.0
    else:
        assert d["config"]["models"] == 21 and d["config"]["models_on_busiest_rank"] == 3
        assert d["ideal_speedup_over_1_gpu"] == 7.0


# ------------------------------------------------------------------------------------------------------------------
# render_path / render_path_grad with the options the reference forwards to render() (RN:233, RN:168)
# ------------------------------------------------------------------------------------------------------------------
def _oracle_dir_on_path():
    import sys
    p = os.path.join(ROOT, "oracle")
    if p not in sys.path:
        sys.path.insert(0, p)


def _g21_pose_draws(g, which, i):
    std = np.float32(float(g["noise_std"]))
    sl = slice(4 * i, 4 * (i + 1))
    cat = lambda k: np.concatenate(list(g[which + "_" + k][sl]), 0)
    return dict(t_rand=cat("t_rand"), u=cat("u"), noise0=(cat("randn0") * std).astype(np.float32),
                noise1=(cat("randn1") * std).astype(np.float32))


class _Feeder:
    """stands in for torch.rand / torch.randn: hands out the draws the REFERENCE recorded (g21), in its call order, checking
    that the build asks for the same shapes in the same order"""

    def __init__(self, g, which):
        n = g[which + "_t_rand"].shape[0]
        self.q = [g[which + "_" + k][c] for c in range(n) for k in ("t_rand", "randn0", "u", "randn1")]
        self.kind = ["rand", "randn", "rand", "randn"] * n
        self.i = 0

    def fn(self, kind):
        import torch

        def f(*shape, device=None, **kw):
            assert self.i < len(self.q), "more draws than the reference made"
            a = self.q[self.i]
            assert self.kind[self.i] == kind and tuple(shape) == a.shape, (self.i, kind, shape, a.shape)
            self.i += 1
            return torch.from_numpy(a.copy()).to(device)
        return f


def test_path_functions_take_the_train_kwargs_like_the_reference(oracle, synth_nets, tmp_path, monkeypatch):
    """render_path(render_kwargs_train) / render_path_grad(render_kwargs_train) -- perturb = 1, raw_noise_std
    > 0; the reference forwards **render_kwargs to render() (RN:233, RN:168) -- against what the REFERENCE returned (g21),
    with ITS draws: torch.rand / torch.randn are replaced by a feeder that replays the recorded numbers and insists on the
    reference's call order and shapes (per 16-ray chunk: t_rand, coarse noise, u, fine noise).  Engine level: the census
    against the reference's own intermediates; API level: images (+ PNGs) and per-patch psi-gradients."""
    import torch
    import neural_sim_nerf_amd.run_nerf_noscale as R
    from neural_sim_nerf_amd import pose as P
    from neural_sim_nerf_amd.engine import NsrModel
    _oracle_dir_on_path()
    import census as C
    g = load_golden("g21_path_options")
    g10 = load_golden("g10_path_grad")
    near, far = oracle.YCBV_NEAR, oracle.YCBV_FAR
    K = g["K"].tolist()
    H = W = 8
    # ---- engine level: census on the reference's intermediates, both poses in one launch -----------------------------
    m = NsrModel(synth_nets[0], synth_nets[1])
    ro, rd, rnd = [], [], []
    for i in range(2):
        o, d = oracle.get_rays(H, W, K, g["poses"][i][:3, :4])
        ro.append(o.reshape(-1, 3)); rd.append(d.reshape(-1, 3)); rnd.append(_g21_pose_draws(g, "path", i))
    ro, rd = np.concatenate(ro), np.concatenate(rd)
    rnd = {k: np.concatenate([x[k] for x in rnd]) for k in rnd[0]}
    r = m.render_rays(ro, rd, near, far, debug=True, extras=rnd)
    flat = lambda a, t: np.asarray(a).reshape((-1,) + np.asarray(a).shape[np.asarray(a).ndim - t:])
    ref = dict(rgb_map=flat(g["path_rgbs"], 1), acc_map=flat(g["path_acc"], 0), disp_map=flat(g["path_disps"], 0),
               rgb0=flat(g["path_rgb0"], 1), acc0=flat(g["path_acc0"], 0), sigma0_last=g["path_sigma0_last"],
               pdf_weights=g["path_pdf_weights"], inds=g["path_inds"], z_samples=g["path_z_samples"])
    taps = ("rgb_map", "acc_map", "disp_map", "rgb0", "acc0", "raw0", "weights0", "inds", "z_samples", "z_fine", "raw")
    c = C.census(synth_nets, ro, rd, near, far, {k: cpu(r[k]) for k in taps}, ref, rnd=rnd)
    print("g21 census:", {k: c[k] for k in ("rays", "rays_above_tol", "unattributed", "psnr_delta_db")})
    assert C.passes(c), c
    assert c["rays_above_tol"] <= max(3, 3 * (43.0 / 1600.0) * c["rays"]) and c["psnr_delta_db"] <= 0.1, c
    assert_close(cpu(r["rgb0"]), ref["rgb0"], atol=1e-5, what="rgb0 vs reference")
    m.close()
    # ---- the drop-in API with the reference's draws ----------------------------------------------------------------------
    nets = []
    for sd in synth_nets:
        n = R.NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
        n.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        nets.append(n.to(R.device))
    kw_train = dict(network_query_fn=None, perturb=1.0, N_importance=128, network_fine=nets[1], N_samples=64,
                    network_fn=nets[0], use_viewdirs=True, white_bkgd=False, raw_noise_std=float(g["noise_std"]), ndc=False,
                    lindisp=False, near=near, far=far)
    hwf = [H, W, K[0][0]]
    feed = _Feeder(g, "path")
    monkeypatch.setattr(torch, "rand", feed.fn("rand"))
    monkeypatch.setattr(torch, "randn", feed.fn("randn"))
    rgbs, disps = R.render_path(None, torch.from_numpy(g["poses"]), hwf, K, int(g["chunk"]), kw_train, savedir=str(tmp_path), object_id=2)
    assert feed.i == len(feed.q)                                          # every draw the reference made was asked for
    monkeypatch.undo()
    assert rgbs.shape == (2, H, W, 3) and disps.shape == (2, H, W)
    assert np.array_equal(rgbs.reshape(-1, 3), cpu(r["rgb_map"]))        # = the engine's render of the same rays and draws
    d = np.abs(rgbs - g["path_rgbs"]).max(-1)
    assert (d > 1e-4).sum() <= 3 and C.psnr_delta(rgbs.reshape(-1, 3), g["path_rgbs"].reshape(-1, 3)) <= 0.1, d.max()
    assert sorted(os.listdir(tmp_path / "2")) == ["000.png", "001.png"]
    # ---- render_path_grad: per-patch dL/dpsi against the reference's autograd chain -------------------------------------
    log = {"gumbel_noises": g10["gumbel"].tolist(), "uniform_noises": g10["uniform"].tolist(), "thetas": g10["thetas"].tolist()}
    prob = torch.softmax(torch.tensor(g10["psi"]) / 0.25, 0).requires_grad_()
    poses = P.sample_pose(prob, 2, 0.1, log)
    grad_E = [{"grad_E": [torch.from_numpy(x)]} for x in g["grad_E"]]
    feed = _Feeder(g, "grad")
    monkeypatch.setattr(torch, "rand", feed.fn("rand"))
    monkeypatch.setattr(torch, "randn", feed.fn("randn"))
    rgbs_g, dl = R.render_path_grad(prob, poses, hwf, K, int(g["chunk"]), grad_E, kw_train, savedir=str(tmp_path))
    assert feed.i == len(feed.q)
    monkeypatch.undo()
    assert rgbs_g.shape == (2, H, W, 3) and len(dl) == 8
    assert oracle.psnr(rgbs_g, g["grad_rgbs"]) > 50.0
    got = np.stack([x.numpy() for x in dl])
    scale = np.abs(g["dLdpsis"]).max()
    err = np.abs(got - g["dLdpsis"]).max() / scale
    print("g21 dL/dpsi: max err %.3e of the largest component" % err)
    assert err < 2e-2, err
    assert np.abs(got.mean(0) - g["dLdpsis"].mean(0)).max() < 1e-2 * scale
    assert sorted(os.listdir(tmp_path / "2" / "withgrad")) == ["000.png", "001.png"]
    # the deterministic kwargs still take the batched route and draw nothing

    def no_draw(*a, **k):
        raise AssertionError("a draw on the deterministic path")
    kw_test = dict(kw_train, perturb=False, raw_noise_std=0.)
    monkeypatch.setattr(torch, "rand", no_draw)
    monkeypatch.setattr(torch, "randn", no_draw)
    R.render_path(None, torch.from_numpy(g["poses"]), hwf, K, 16, kw_test, savedir=str(tmp_path))
    monkeypatch.undo()


def test_weight_changes_reach_the_renderer(synth_nets, oracle):
    """The packed-weight cache of render() (run_nerf_helpers: NSR_TRUST_VERSIONS, r06 default): storage identity + autograd
    versions key every call -- a parameter update that bumps the version (optimizer step, load_state_dict, copy_ under
    no_grad) is seen by the very next call of either form -- and so is a write through `.data`, which bumps nothing: per-view
    calls compare the content fingerprint before they launch, the 512-ray patch form reads it while its render runs and
    renders again when it differs (a caller that ONLY uses render(rays=...) never got the check in r05).  Also
    through autograd, and on the layered renderer."""
    import torch
    import neural_sim_nerf_amd.run_nerf_noscale as R
    import neural_sim_nerf_amd.run_nerf_helpers as RHm
    assert RHm.DEFER_PATCH_CHECK and not RHm.TRUST_PATCH_CALLS and not RHm.TRUST_VERSIONS       # (the suite runs with the default)
    near, far = oracle.YCBV_NEAR, oracle.YCBV_FAR
    K = oracle.scaled_K(50.0)
    nets = []
    for sd in synth_nets:
        n = R.NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
        n.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        nets.append(n.to(R.device))
    kw = dict(network_query_fn=None, perturb=False, N_importance=128, network_fine=nets[1], N_samples=64, network_fn=nets[0],
              use_viewdirs=True, white_bkgd=False, raw_noise_std=0., ndc=False, lindisp=False, near=near, far=far)
    pose = torch.tensor(np.asarray(oracle.sweep_poses(1, seed=3))[0][:3, :4])
    ro, rd = oracle.get_rays(8, 8, K, pose.numpy())
    rays = torch.stack([torch.from_numpy(ro.reshape(-1, 3)), torch.from_numpy(rd.reshape(-1, 3))], 0)
    for kwx in (kw, dict(kw, N_samples=40, N_importance=72)):                    # the fused kernels, the layered renderer
        with torch.no_grad():
            patch = lambda: cpu(R.render(8, 8, K, rays=rays, **kwx)[0])
            view = lambda: cpu(R.render(8, 8, K, c2w=pose, **kwx)[0]).reshape(-1, 3)
            a = patch()
            assert np.array_equal(a, view()) and np.array_equal(a, patch())
            m0 = R._model_for(nets[0], nets[1], kwx["N_importance"], kwx)
            # (1) a versioned update: the patch form sees it at once
            nets[1].rgb_linear.bias.add_(0.25)
            b = patch()
            assert np.abs(b - a).max() > 1e-3 and R._model_for(nets[0], nets[1], kwx["N_importance"], kwx) is not m0
            assert np.array_equal(b, view())
            # (2) a write through .data: no version moves; the very next PATCH call finds it by content (no view call in between)
            nets[1].rgb_linear.bias.data.add_(-0.25)
            c = patch()
            assert np.abs(c - a).max() < 1e-6 and np.array_equal(patch(), c) and np.array_equal(view(), c)
            # (3) ... and a patch-only caller keeps being checked: another .data write, two patch calls
            nets[0].alpha_linear.bias.data.add_(0.5)
            d = patch()
            assert np.abs(d - c).max() > 1e-4 and np.array_equal(patch(), d)
            nets[0].alpha_linear.bias.data.add_(-0.5)
            assert np.array_equal(patch(), c)
        # (4) the same through autograd: the gradient comes from the repacked weights as well
        r1 = rays.clone().requires_grad_(True)
        g1 = torch.autograd.grad(R.render(8, 8, K, rays=r1, **kwx)[0].sum(), r1)[0]
        saved = nets[1].rgb_linear.weight.data.clone()
        nets[1].rgb_linear.weight.data.mul_(1.5)
        r2 = rays.clone().requires_grad_(True)
        g2 = torch.autograd.grad(R.render(8, 8, K, rays=r2, **kwx)[0].sum(), r2)[0]
        nets[1].rgb_linear.weight.data.copy_(saved)
        r3 = rays.clone().requires_grad_(True)
        g3 = torch.autograd.grad(R.render(8, 8, K, rays=r3, **kwx)[0].sum(), r3)[0]
        assert (g1 - g2).abs().max() > 1e-6 and torch.equal(g1, g3)
    for n in nets:
        n.invalidate()


# ------------------------------------------------------------------------------------------------------------------
# other sample counts (RN:439 N_samples, RN:474 N_importance; NM:1258-1260): kernels specialised to (64, 96), (32, 64), (128, 128)
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["g22_counts_64_96", "g23_counts_32_64", "g24_counts_128_128"])
def test_native_sample_counts(name, oracle, synth_nets):
    """N_samples 32 / 128 and N_importance 96 on kernels of their own (f16x2 handles), against the reference's
    run at those counts (g22-g24, oracle pinned to them by tests/test_oracle_golden.py): every stage on the kernel's own
    intermediates -- cdf -> indices -> samples and the merged depths BIT FOR BIT --, the coarse image against the reference
    1e-5, the census against the oracle's render, PSNR-delta against the reference's pixels, the gradient at the reference's
    depths against the reference's autograd, ray independence, coarse-only handles, the per-ray options, and the range safety
    net's fallback at these counts."""
    import sys
    p = os.path.join(ROOT, "oracle")
    if p not in sys.path:
        sys.path.insert(0, p)
    import census as C
    from neural_sim_nerf_amd.engine import NsrModel
    g = load_golden(name)
    ns, ni = int(g["n_samples"]), int(g["n_importance"])
    near, far = oracle.YCBV_NEAR, oracle.YCBV_FAR
    ro, rd = g["rays_o"], g["rays_d"]
    n = len(ro)
    vd = oracle.normalize_dirs(rd)
    m = NsrModel(synth_nets[0], synth_nets[1], n_samples=ns, n_importance=ni)
    assert m.mlp == "f16x2" and (m.n_samples, m.ni_kernel, m.nf_kernel) == (ns, ni, ns + ni)
    r = m.render_rays(ro, rd, near, far, debug=True)
    shapes = {k: tuple(v.shape[1:]) for k, v in r.items()}
    assert shapes["weights0"] == (ns,) and shapes["raw0"] == (ns, 4) and shapes["z_samples"] == (ni,) and shapes["inds"] == (ni,)
    assert shapes["z_fine"] == (ns + ni,) and shapes["raw"] == (ns + ni, 4)
    # stage by stage on the kernel's own intermediates
    z = oracle.coarse_z(np.full(n, near, np.float32), np.full(n, far, np.float32), n=ns)
    raw0 = oracle.run_network(synth_nets[0], (ro[:, None] + rd[:, None] * z[..., None]).astype(np.float32), vd)
    assert_close(cpu(r["raw0"]), raw0, atol=5e-5, rtol=5e-5, what="coarse raw")
    rgb0, _, acc0, w0, _ = oracle.raw2outputs(cpu(r["raw0"]), z, rd)
    assert_close(cpu(r["weights0"]), w0, atol=2e-6, what="weights0 | kernel raw")
    assert_close(cpu(r["rgb0"]), rgb0, atol=3e-6, what="rgb0 | kernel raw")
    z_mid = (np.float32(0.5) * (z[:, 1:] + z[:, :-1])).astype(np.float32)
    s, inds, _ = oracle.sample_pdf(z_mid, cpu(r["weights0"])[:, 1:-1], ni)
    assert np.array_equal(cpu(r["inds"]), inds) and np.array_equal(cpu(r["z_samples"]), s)
    zf = np.sort(np.concatenate([z, s], -1), -1)
    assert np.array_equal(cpu(r["z_fine"]), zf)
    raw = oracle.run_network(synth_nets[1], (ro[:, None] + rd[:, None] * zf[..., None]).astype(np.float32), vd)
    assert_close(cpu(r["raw"]), raw, atol=5e-5, rtol=5e-5, what="fine raw | kernel z")
    rgb, disp, acc, _, _ = oracle.raw2outputs(cpu(r["raw"]), zf, rd)
    assert_close(cpu(r["rgb_map"]), rgb, atol=3e-6, what="rgb | kernel raw")
    assert_close(cpu(r["acc_map"]), acc, atol=3e-6, what="acc | kernel raw")
    assert_close(cpu(r["z_std"]), np.std(s.astype(np.float64), -1), atol=1e-6, what="z_std")
    # the kernel's inverse CDF on the REFERENCE's coarse weights: the reference's indices and samples, bit for bit (stage entry
    # points are specialised to 64 / 128, so this goes through the oracle, which the line above ties to the kernel)
    s_ref, inds_ref, _ = oracle.sample_pdf(z_mid, g["pdf_weights"], ni)
    assert np.array_equal(inds_ref, g["inds"].astype(np.int64)) and np.array_equal(s_ref, g["z_samples"])
    # against the reference: coarse image, end to end census (vs the oracle's render of these counts) and PSNR-delta
    assert_close(cpu(r["rgb0"]), g["rgb0"], atol=1e-5, what="rgb0 vs reference")
    ref = oracle.render_rays(synth_nets[0], synth_nets[1], ro, rd, vd, near, far, n_samples=ns, n_importance=ni, extras=True)
    taps = ("rgb_map", "acc_map", "disp_map", "rgb0", "acc0", "raw0", "weights0", "inds", "z_samples", "z_fine", "raw")
    c = C.census(synth_nets, ro, rd, near, far, {k: cpu(r[k]) for k in taps}, ref, n_importance=ni, n_samples=ns)
    print(name, "census:", {k: c[k] for k in ("rays", "rays_above_tol", "unattributed", "psnr_delta_db")})
    assert C.passes(c) and c["rays_above_tol"] <= 3 and c["psnr_delta_db"] <= 0.1, c
    assert C.psnr_delta(cpu(r["rgb_map"]), g["rgb"]) <= 0.1
    # gradient at the reference's depths, vs the reference's autograd; the VJP launch's forward == the forward kernel
    zf_ref = np.sort(np.concatenate([z, g["z_samples"]], -1), -1)
    go, gd = m.render_rays_vjp(ro, rd, near, far, g["cot"], z_fine=zf_ref)
    for a, b in ((cpu(go), g["grad_rays"][0]), (cpu(gd), g["grad_rays"][1])):
        e = np.linalg.norm(a - b, axis=1) / (np.linalg.norm(b, axis=1) + 1e-12)
        assert np.percentile(e, 90) < 3e-4 and np.linalg.norm(a - b) / np.linalg.norm(b) < 2e-2, (np.percentile(e, 90), e.max())
    go2, gd2, fwd = m.render_rays_vjp(ro, rd, near, far, g["cot"], with_forward=True)
    assert np.array_equal(cpu(fwd["rgb_map"]), cpu(r["rgb_map"]), equal_nan=True)
    wo, wd, _ = oracle.render_rays_vjp(synth_nets[0], synth_nets[1], ro, rd, near, far, g["cot"], n_samples=ns, n_importance=ni, z_fine=zf)
    for a, b in ((cpu(go2), wo), (cpu(gd2), wd)):
        e = np.linalg.norm(a - b, axis=1) / (np.linalg.norm(b, axis=1) + 1e-12)
        assert np.percentile(e, 90) < 3e-4, np.percentile(e, 90)
    # ray independence: odd subsets (an item is two rays) forward and backward
    for k in (1, 7, 33):
        sub = m.render_rays(ro[:k], rd[:k], near, far)
        for key in ("rgb_map", "disp_map", "acc_map", "rgb0", "z_std"):
            assert np.array_equal(cpu(sub[key]), cpu(r[key])[:k], equal_nan=True), (key, k)
        so, sd = m.render_rays_vjp(ro[:k], rd[:k], near, far, g["cot"][:k])
        assert np.array_equal(cpu(so), cpu(go2)[:k]) and np.array_equal(cpu(sd), cpu(gd2)[:k])
    # a view through render_views: rays generated in-kernel == the same rays given
    K = oracle.scaled_K(20.0)
    pose = np.asarray(oracle.sweep_poses(1, seed=3))[0]
    v = m.render_views(pose, 20, 20, K, near, far)
    vo, vdir = oracle.get_rays(20, 20, K, pose[:3, :4])
    vr = m.render_rays(vo.reshape(-1, 3), vdir.reshape(-1, 3), near, far)
    assert np.array_equal(cpu(v["rgb_map"]), cpu(vr["rgb_map"]))
    # the per-ray options at these counts: stratified depths + random uniforms + density noise, stage-wise against the oracle
    rng = np.random.RandomState(5)
    rnd = dict(t_rand=rng.rand(n, ns).astype(np.float32), u=rng.rand(n, 128).astype(np.float32),
               noise0=(0.3 * rng.standard_normal((n, ns))).astype(np.float32),
               noise1=(0.3 * rng.standard_normal((n, ns + ni))).astype(np.float32))
    q = m.render_rays(ro, rd, near, far, debug=True, extras=rnd)
    zq = oracle.perturb_z(z, rnd["t_rand"])
    _, _, _, w0q, _ = oracle.raw2outputs(cpu(q["raw0"]), zq, rd, False, rnd["noise0"])
    assert_close(cpu(q["weights0"]), w0q, atol=2e-6, what="weights0 with options")
    zq_mid = (np.float32(0.5) * (zq[:, 1:] + zq[:, :-1])).astype(np.float32)
    sq, iq, _ = oracle.sample_pdf(zq_mid, cpu(q["weights0"])[:, 1:-1], ni, rnd["u"][:, :ni])
    assert np.array_equal(cpu(q["inds"]), iq) and np.array_equal(cpu(q["z_samples"]), sq)
    assert np.array_equal(cpu(q["z_fine"]), np.sort(np.concatenate([zq, sq], -1), -1))
    rgbq, _, accq, _, _ = oracle.raw2outputs(cpu(q["raw"]), cpu(q["z_fine"]), rd, False, rnd["noise1"])
    assert_close(cpu(q["rgb_map"]), rgbq, atol=3e-6, what="rgb with options")
    m.close()
    # coarse only at this N_samples
    if ns != 64:
        mc = NsrModel(synth_nets[0], None, n_samples=ns, n_importance=0)
        rc = mc.render_rays(ro, rd, near, far, debug=True)
        assert cpu(rc["raw0"]).shape == (n, ns, 4)
        rgbc, _, accc, _, _ = oracle.raw2outputs(cpu(rc["raw0"]), z, rd)
        assert_close(cpu(rc["rgb_map"]), rgbc, atol=3e-6, what="coarse-only rgb")
        assert_close(cpu(rc["rgb_map"]), g["rgb0"], atol=1e-5, what="coarse-only rgb vs the reference's rgb0")
        # many items per workgroup (the persistent loop's pass counter across items): 40 repeats of the rays = 960 items on 256
        # workgroups, every repeat bit-equal to the first
        rep = mc.render_rays(np.tile(ro, (40, 1)), np.tile(rd, (40, 1)), near, far)
        assert np.array_equal(cpu(rep["rgb_map"]).reshape(40, n, 3), np.broadcast_to(cpu(rc["rgb_map"]), (40, n, 3)))
        mc.close()
    # ... and the same for the coarse + fine handle, forward and input gradient (several items per workgroup)
    m2 = NsrModel(synth_nets[0], synth_nets[1], n_samples=ns, n_importance=ni)
    one = m2.render_rays(ro, rd, near, far)
    rep = m2.render_rays(np.tile(ro, (40, 1)), np.tile(rd, (40, 1)), near, far)
    for key in ("rgb_map", "disp_map", "acc_map", "rgb0", "z_std"):
        v = cpu(rep[key])
        assert np.array_equal(v.reshape((40, n) + v.shape[1:]), np.broadcast_to(cpu(one[key]), (40, n) + v.shape[1:]), equal_nan=True), key
    g1o, g1d = m2.render_rays_vjp(ro, rd, near, far, g["cot"])
    gro, grd = m2.render_rays_vjp(np.tile(ro, (40, 1)), np.tile(rd, (40, 1)), near, far, np.tile(g["cot"], (40, 1)))
    assert np.array_equal(cpu(grd).reshape(40, n, 3), np.broadcast_to(cpu(g1d), (40, n, 3)))
    assert np.array_equal(cpu(gro).reshape(40, n, 3), np.broadcast_to(cpu(g1o), (40, n, 3)))
    m2.close()
    # the range safety net at these counts: every point of the coarse network overflows -> the bf16x3 kernel of the same counts
    big = [{k: np.array(v, copy=True) for k, v in sd.items()} for sd in synth_nets]
    big[0]["pts_linears.0.bias"][7] = 7.0e4
    mb = NsrModel(big[0], big[1], n_samples=ns, n_importance=ni)
    rb = mb.render_rays(ro, rd, near, far, debug=True)
    st = mb.range_status()
    assert st["rays"] == n and st["dropped_items"] == 0 and np.isfinite(cpu(rb["rgb_map"])).all(), st
    refb = oracle.render_rays(big[0], big[1], ro, rd, vd, near, far, n_samples=ns, n_importance=ni, extras=True)
    assert_close(cpu(rb["rgb0"]), refb["rgb0"], atol=1e-5, what="fallback rgb0 vs oracle")
    cb = C.census(big, ro, rd, near, far, {k: cpu(rb[k]) for k in taps}, refb, n_importance=ni, n_samples=ns)
    assert C.passes(cb), cb
    gob, gdb = mb.render_rays_vjp(ro, rd, near, far, g["cot"])
    assert np.isfinite(cpu(gdb)).all() and mb.range_status()["dropped_items"] == 0
    mb.close()


def test_dropin_api_takes_the_other_sample_counts(oracle, synth_nets):
    """render(N_samples=32, N_importance=64) through the drop-in API == the engine's render of the same rays; counts without a
    fused kernel (N_samples = 48; (32, 128); (128, 64)) go to the layered renderer; N_samples = 2 is refused."""
    import torch
    import neural_sim_nerf_amd.run_nerf_noscale as R
    from neural_sim_nerf_amd.engine import NsrModel
    g = load_golden("g23_counts_32_64")
    near, far = oracle.YCBV_NEAR, oracle.YCBV_FAR
    nets = []
    for sd in synth_nets:
        n = R.NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
        n.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        nets.append(n.to(R.device))
    kw = dict(network_query_fn=None, perturb=False, N_importance=64, network_fine=nets[1], N_samples=32, network_fn=nets[0],
              use_viewdirs=True, white_bkgd=False, raw_noise_std=0., ndc=False, lindisp=False, near=near, far=far)
    rays = (torch.tensor(g["rays_o"]), torch.tensor(g["rays_d"]))
    rgb, disp, acc, ex = R.render(400, 400, oracle.YCBV_K, rays=rays, retraw=True, **kw)
    assert tuple(ex["raw"].shape) == (len(g["rays_o"]), 96, 4)
    m = NsrModel(synth_nets[0], synth_nets[1], n_samples=32, n_importance=64)
    want = m.render_rays(g["rays_o"], g["rays_d"], near, far)
    assert np.array_equal(cpu(rgb), cpu(want["rgb_map"]))
    assert oracle.psnr(cpu(rgb), g["rgb"]) > 50.0
    m.close()
    # counts without a fused kernel: the layered renderer (include/nsr_wide.h) takes them since r05 -- against the oracle's render
    ro, rd = g["rays_o"], g["rays_d"]
    for other in (dict(N_samples=48), dict(N_samples=32, N_importance=128), dict(N_samples=128, N_importance=64)):
        kw2 = dict(kw, **other)
        ns, ni = kw2["N_samples"], kw2["N_importance"]
        assert "N_samples=%d" % ns in R._layered_why(nets[0], nets[1], ns, ni)
        got = R.render(400, 400, oracle.YCBV_K, rays=rays, **kw2)
        assert R._model_for(nets[0], nets[1], ni, kw2).mlp.startswith("layered-")
        ref = oracle.render_rays(synth_nets[0], synth_nets[1], ro, rd, oracle.normalize_dirs(rd), near, far, n_samples=ns, n_importance=ni)
        d = np.abs(cpu(got[0]) - ref["rgb_map"]).max(-1)
        assert (d > 1e-4).mean() <= 0.1 and oracle.psnr(cpu(got[0]), ref["rgb_map"]) > 50.0, (other, d.max())
        assert_close(cpu(got[3]["rgb0"]), ref["rgb0"], atol=1e-5, what="rgb0 %r" % (other,))
    with pytest.raises(NotImplementedError, match="N_samples"):
        R.render(400, 400, oracle.YCBV_K, rays=rays, **dict(kw, N_samples=2))
    for n in nets:
        n.invalidate()


def test_stage_methods_refuse_handles_of_other_counts(synth_nets):
    """NsrModel.sample_pdf is specialised to 63 bins / 128 samples and reads the handle's uniforms table; a handle
    whose kernels (and table) are specialised to other counts must not serve it with a table of padding zeros."""
    from neural_sim_nerf_amd.engine import NsrModel
    m = NsrModel(synth_nets[0], synth_nets[1], n_importance=64)
    with pytest.raises(NotImplementedError, match="specialised"):
        m.sample_pdf(np.zeros((2, 63), np.float32), np.ones((2, 62), np.float32))
    m.close()
    m = NsrModel(synth_nets[0], synth_nets[1])                 # the YCB-V handle serves it
    s, inds = m.sample_pdf(np.linspace(0.3, 1.9, 63, dtype=np.float32)[None].repeat(2, 0), np.ones((2, 62), np.float32))
    assert tuple(s.shape) == (2, 128) and (np.diff(cpu(s), axis=1) >= 0).all()
    m.close()
