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
