"""world_size-2 `gloo` tests of the multi-GPU path (view sharding, image all-gather, psi-gradient all-reduce) on
CPU.  The renderer is replaced by a deterministic function of the pose: what is under test is the sharding,
ordering, padding and reduction logic of neural_sim_nerf_amd/dist.py, not the kernel."""
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
    q.put((rank, bool(ok), D.shard_indices(n_views, world, rank)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_views", [5, 4, 1])
def test_view_sharding_gather_and_grad_allreduce(tmp_path, n_views):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_views, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    seen = sorted(i for _, _, idx in res for i in idx)
    assert seen == list(range(n_views))                       # every view rendered exactly once
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
