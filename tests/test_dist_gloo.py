"""`gloo` tests of the multi-GPU path (view sharding, image all-gather, psi-gradient all-reduce) on CPU, at world sizes
2, 3, 5 and 8 -- the 8-rank layout of BASELINE configs[2] / [4] included: 100 views -> 13 / 12 per rank, 21 models ->
3,3,3,3,3,2,2,2, and K = 5 views on 8 ranks (three ranks with nothing to render).  The renderer is replaced by a
deterministic function of the pose: what is under test is the sharding, ordering, padding and reduction logic of
neural_sim_nerf_amd/dist.py, not the kernel."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_render(poses):
    """rgb[i] = f(pose_i): lets the receiver verify which view landed where."""
    k = poses.shape[0]
    base = poses[:, :3, 3].reshape(k, 1, 1, 3)
    ramp = torch.linspace(0, 1, 4 * 5).reshape(1, 4, 5, 1)
    rgb = (base * 0.1 + ramp).to(torch.float32).expand(k, 4, 5, 3).contiguous()
    return rgb, rgb[..., 0] * 2.0


def _worker(rank, world, port, n_views, tmp, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from neural_sim_nerf_amd import dist as D
    poses = torch.eye(4).repeat(n_views, 1, 1)
    poses[:, :3, 3] = torch.arange(n_views * 3, dtype=torch.float32).reshape(n_views, 3)
    rgbs, disps = D.render_path_distributed(_fake_render, poses, savedir=tmp, object_id=7)
    want_rgb, want_disp = _fake_render(poses)
    ok = np.array_equal(rgbs, want_rgb.numpy()) and np.array_equal(disps, want_disp.numpy())
    # in-memory hand-off (uint8 images + int32 boxes) gathered in pose order
    mine = D.shard_indices(n_views, world, rank)
    ann = {"images": torch.stack([torch.full((4, 5, 3), i, dtype=torch.uint8) for i in mine]) if mine else
           torch.zeros((0, 4, 5, 3), dtype=torch.uint8),
           "bbox": torch.tensor([[i, i + 1, i + 2, i + 3] for i in mine], dtype=torch.int32).reshape(-1, 4),
           "count": torch.tensor([i + 1 for i in mine], dtype=torch.int32)}
    allv = D.gather_handoff(ann, n_views)
    ok = ok and allv["images"].dtype == torch.uint8 and allv["bbox"].dtype == torch.int32
    ok = ok and all(int(allv["images"][i, 0, 0, 0]) == i and allv["bbox"][i].tolist() == [i, i + 1, i + 2, i + 3]
                    and int(allv["count"][i]) == i + 1 for i in range(n_views))
    local = [torch.full((8,), float(i)) for i in D.shard_indices(n_views, world, rank)]   # "patch gradients"
    g = D.mean_psi_grad(local)
    ok = ok and torch.allclose(g, torch.full((8,), (n_views - 1) / 2.0))
    # a rank WITHOUT gradients (fewer poses than ranks) and a psi of another length (r03 hard-coded 8 for that rank)
    g5 = D.mean_psi_grad([torch.full((5,), 4.0), torch.full((5,), 6.0)] if rank == 0 else [])
    ok = ok and tuple(g5.shape) == (5,) and torch.allclose(g5, torch.full((5,), 5.0))
    # ... the same with n_cat supplied by the caller (no length agreement round), and ranks whose gradients disagree in
    # length: EVERY rank raises (a rank that raised alone would leave the others hanging in the next collective)
    g5 = D.mean_psi_grad([torch.full((5,), 4.0), torch.full((5,), 6.0)] if rank == world - 1 else [], n_cat=5)
    ok = ok and tuple(g5.shape) == (5,) and torch.allclose(g5, torch.full((5,), 5.0))
    try:
        D.mean_psi_grad([torch.ones(8 if rank == 0 else 7)])
        ok = False
    except ValueError as e:
        ok = ok and "entries" in str(e)
    # the sharding rule itself: every rank within one view of the others, models like views (configs[4])
    ok = ok and len(mine) in (n_views // world, -(-n_views // world))
    ok = ok and D.shard_models(21, world, rank) == list(range(rank, 21, world))
    q.put((rank, bool(ok), D.shard_indices(n_views, world, rank)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_views", [(2, 5), (2, 4), (2, 1), (8, 100), (8, 5), (5, 21), (3, 100)])
def test_view_sharding_gather_and_grad_allreduce(tmp_path, world, n_views):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_views, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    seen = sorted(i for _, _, idx in res for i in idx)
    assert seen == list(range(n_views))                       # every view rendered exactly once
    sizes = sorted((len(idx) for _, _, idx in res), reverse=True)
    if (world, n_views) == (8, 100):
        assert sizes == [13, 13, 13, 13, 12, 12, 12, 12]      # BASELINE configs[2] on one node: ideal speed-up 100 / 13 = 7.7
    if (world, n_views) == (8, 5):
        assert sizes == [1, 1, 1, 1, 1, 0, 0, 0]              # K < N: three ranks render nothing and still take part
    files = sorted(os.listdir(tmp_path / "7"))
    assert files == ["%03d.png" % i for i in range(n_views)]  # each rank wrote its own views, named by pose index


def test_single_process_paths():
    import sys
    sys.path.insert(0, ROOT)
    from neural_sim_nerf_amd import dist as D
    assert D.world_info() == (1, 0)
    assert D.shard_indices(10, 4, 1) == [1, 5, 9]
    x = torch.arange(6.).reshape(3, 2)
    assert D.gather_views(x, 3) is x
    g = D.mean_psi_grad([torch.ones(8), 3 * torch.ones(8)])
    assert torch.allclose(g, 2 * torch.ones(8))


# ------------------------------------------------------------------------------------------------------
# the drop-in API under a process group: render_path / render_path_grad shard by themselves (NM:128 / NM:184
# unchanged under torchrun), models shard for config 5
# ------------------------------------------------------------------------------------------------------
class _FakeModel:
    """Stands in for engine.NsrModel on CPU: every output is a deterministic function of the pose / cotangent."""
    device = torch.device("cpu")
    n_importance = 128

    def render_views(self, c2w, H, W, K, near, far, debug=False):
        v = c2w.shape[0]
        ramp = torch.linspace(0, 1, H * W * 3).reshape(1, H * W, 3)
        rgb = (c2w[:, :3, 3].sum(1).reshape(v, 1, 1) * 0.01 + ramp).reshape(v * H * W, 3).to(torch.float32)
        return {"rgb_map": rgb, "disp_map": rgb[:, 0] * 3.0}

    def get_rays(self, H, W, K, c2w):
        o = c2w[:3, 3].reshape(1, 1, 3).expand(H, W, 3).contiguous()
        d = (c2w[:3, :3].sum(1).reshape(1, 1, 3) + torch.arange(H * W, dtype=torch.float32).reshape(H, W, 1) * 1e-3)
        return o, d.contiguous()

    def render_rays_vjp(self, ro, rd, near, far, cot, with_forward=False, z_fine=None):
        go, gd = cot * ro.sum(1, keepdim=True), cot * 0.5 + rd
        return go, gd, {"rgb_map": (ro + rd) * 0.25}

    def pose_grad(self, go, gd, H, W, K, patch):
        n = H * W
        n_p = (n + patch - 1) // patch
        out = torch.zeros(n_p, 3, 4)
        for p in range(n_p):
            sl = slice(p * patch, min(n, (p + 1) * patch))
            out[p, :, :3] = gd[sl].sum(0).reshape(3, 1) * torch.tensor([1.0, 2.0, 3.0])
            out[p, :, 3] = go[sl].sum(0)
        return out


def _dropin_worker(rank, world, port, n_views, tmp, q, desync):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import neural_sim_nerf_amd.run_nerf_noscale as R
    from neural_sim_nerf_amd import dist as D
    R._model_for = lambda *a, **k: _FakeModel()
    H = W = 6
    hwf, K, chunk = [H, W, 100.0], [[100.0, 0, 3.0], [0, 100.0, 3.0], [0, 0, 1.0]], 8
    kw = {"network_fn": None, "network_fine": None, "N_importance": 128, "N_samples": 64, "use_viewdirs": True,
          "ndc": False, "near": 0.3, "far": 1.9, "perturb": 0.0, "raw_noise_std": 0.0}
    prob = torch.softmax(torch.arange(8.0) / 3.0, 0).requires_grad_()
    base = torch.eye(4).repeat(n_views, 1, 1)
    base[:, :3, 3] = torch.arange(n_views * 3, dtype=torch.float32).reshape(n_views, 3) * 0.1
    if desync and rank == 1:
        base[0, 0, 3] += 1.0
    wv = torch.linspace(-1, 1, n_views * 8).reshape(n_views, 8)
    poses = [base[i] + (prob * wv[i]).sum() * torch.ones(4, 4) for i in range(n_views)]     # graph to prob
    grad_E = [{"grad_E": [torch.full((3, H, W), float(i + 1))]} for i in range(n_views)]
    try:
        sharded = R.render_path(None, torch.stack(poses).detach(), hwf, K, chunk, kw, savedir=tmp, object_id=3)
        sharded_g = R.render_path_grad(prob, poses, hwf, K, chunk, grad_E, kw, savedir=tmp, object_id=3)
    except RuntimeError as e:
        q.put((rank, "error: " + str(e)[:60], None))
        dist.destroy_process_group()
        return
    os.environ["NSR_AUTO_SHARD"] = "0"                      # every rank renders everything: the reference's behaviour
    full = R.render_path(None, torch.stack(poses).detach(), hwf, K, chunk, kw)
    full_g = R.render_path_grad(prob, poses, hwf, K, chunk, grad_E, kw)
    ok = np.array_equal(sharded[0], full[0]) and np.array_equal(sharded[1], full[1])
    ok = ok and np.array_equal(sharded_g[0], full_g[0]) and len(sharded_g[1]) == len(full_g[1]) == n_views * 5
    ok = ok and all(torch.equal(a, b) for a, b in zip(sharded_g[1], full_g[1]))
    # NM:191: the mean over the stacked list == the all-reduce form over each rank's own share
    mine = D.shard_indices(n_views, world, rank)
    local = [g for i in mine for g in full_g[1][i * 5:(i + 1) * 5]]
    ok = ok and torch.allclose(D.mean_psi_grad(local), torch.stack(full_g[1]).mean(0), atol=1e-6)
    ok = ok and D.shard_models(21, world, rank) == list(range(rank, 21, world))
    q.put((rank, "ok" if ok else "mismatch", None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_views", [(2, 3), (2, 2), (8, 5), (8, 21), (3, 7)])
def test_dropin_render_path_and_grad_shard_themselves(tmp_path, world, n_views):
    """(8, 5): fewer poses than ranks -- three ranks render nothing, pad the gathers with zero views and contribute a zero
    patch count to the psi-gradient mean; (8, 21) and (3, 7): uneven shares (k_max != local count on most ranks)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dropin_worker, args=(r, world, port, n_views, str(tmp_path), q, False)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] == "ok" for r in res), res
    assert sorted(os.listdir(tmp_path / "3")) == ["%03d.png" % i for i in range(n_views)] + ["withgrad"]
    assert sorted(os.listdir(tmp_path / "3" / "withgrad")) == ["%03d.png" % i for i in range(n_views)]


def test_dropin_sharding_refuses_ranks_with_different_poses(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dropin_worker, args=(r, 2, port, 3, str(tmp_path), q, True)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1].startswith("error: view sharding") for r in res), res
