"""Multi-GPU view sharding: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  Views are independent (the reference's loop RN:229 carries no state),
so the data path has NO collective: rank r renders views r, r+world, ...; the only communication is at the
outer-loop boundary -- one all-gather of the rendered images (1.92 MB fp32 per 400x400 view) and, for the
bilevel gradient, one all-reduce of the 8-float psi-gradient sum plus its patch count (NM:191 takes the mean
over all patches of all poses, and every patch has the same weight)."""
import os
import time

import numpy as np
import torch
import torch.distributed as dist

LAST_TIMINGS = {}          # per-phase seconds of the last render_path_distributed call (max over ranks) when NSR_DIST_TIMING=1
# NSR_DIST_FORCE_COLLECTIVES=1: issue every collective also when the group has ONE rank (where each is the identity).  A
# one-GPU box can then run this module's RCCL calls for real -- dtypes, devices, contiguity -- instead of skipping them
# (tests/test_gpu_parity.py::test_collective_wrappers_over_rccl_with_one_rank).
_FORCE = os.environ.get("NSR_DIST_FORCE_COLLECTIVES", "0") == "1"


def world_info(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def shard_indices(n_items, world, rank):
    """view i -> rank i mod world (keeps every rank within one view of the others)."""
    return list(range(rank, n_items, world))


def shard_models(n_models, world, rank):
    """BASELINE config 5 (all 21 YCB-V NeRFs): model m -> rank m mod world; on a rank every model gets its own
    native handle and HIP stream.  Same rule as the views, kept separate because it is a different contract."""
    return list(range(rank, n_models, world))


def auto_shard_enabled():
    """The drop-in render_path / render_path_grad shard their poses over the default process group when one is
    initialised with more than one rank (so neural_sim_main.py runs unchanged under torchrun); NSR_AUTO_SHARD=0
    turns that off (every rank then renders everything, like the reference would)."""
    return (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
            and os.environ.get("NSR_AUTO_SHARD", "1") != "0")


def _comm_device(group=None):
    """Where collectives' buffers must live: the GPU for nccl (= RCCL), the host for gloo."""
    if dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


_DEVICES_CHECKED = set()


def check_distinct_devices(group=None):
    """One process per GPU: with the nccl (= RCCL) backend every rank of a node must sit on its OWN device.  A script
    that never calls torch.cuda.set_device(LOCAL_RANK) leaves every rank on GPU 0 -- RCCL then fails with a
    duplicate-GPU error deep inside the first collective, or the ranks silently serialise on one device.  Checked once
    per group, before the first sharded render: (hostname, device) pairs are all-gathered and must be distinct.
    NSR_ALLOW_SHARED_GPU=1 (validation runs with gloo on a one-GPU box) skips it."""
    key = id(group)
    if key in _DEVICES_CHECKED or os.environ.get("NSR_ALLOW_SHARED_GPU", "0") == "1":
        return
    world, rank = world_info(group)
    if world > 1 and dist.get_backend(group) == "nccl":
        import socket
        mine = (socket.gethostname(), torch.cuda.current_device())
        seen = [None] * world
        dist.all_gather_object(seen, mine, group=group)
        if len(set(seen)) != world:
            raise RuntimeError("view sharding over RCCL needs one GPU per rank, but the ranks sit on %s: call "
                               "torch.cuda.set_device(int(os.environ['LOCAL_RANK'])) before the first render (INTEGRATION.md "
                               "5), or set NSR_AUTO_SHARD=0" % (sorted(seen),))
    _DEVICES_CHECKED.add(key)


def check_same_poses(poses, group=None, tol=1e-6):
    """Sharding by index only makes sense if every rank holds the SAME pose list.  The reference seeds its pose
    sampler from the wall clock (LL:273), so ranks of an unchanged script can disagree: fail loudly then."""
    world, rank = world_info(group)
    if world == 1 and not _FORCE:
        return
    dev = _comm_device(group)
    p0 = torch.as_tensor(poses, dtype=torch.float32).detach().to(dev).contiguous().clone()
    mine = p0.clone()
    dist.broadcast(p0, src=0, group=group)
    bad = ((mine - p0).abs().max() > tol).to(torch.int32).reshape(1) if mine.numel() else torch.zeros(1, dtype=torch.int32, device=dev)
    dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=group)
    if int(bad.item()):
        raise RuntimeError("view sharding: ranks hold different render_poses (the reference seeds sample_pose from "
                           "the wall clock, LL:273) -- seed the pose sampler identically on every rank, or set "
                           "NSR_AUTO_SHARD=0")


def gather_views(local, n_total, group=None):
    """local: [k_local, ...] tensor of this rank's views (in shard order).  Returns [n_total, ...] in global view
    order on every rank.  Ranks hold ceil or floor(n_total/world) views; shorter ranks pad to the maximum."""
    world, rank = world_info(group)
    if world == 1 and not _FORCE:
        return local
    k_max = (n_total + world - 1) // world
    if local.device != _comm_device(group):
        local = local.to(_comm_device(group))
    pad = torch.zeros((k_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    flat = torch.empty((world * k_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(flat, pad, group=group) if _has_into_tensor(group) else \
        dist.all_gather(list(flat.chunk(world, 0)), pad, group=group)
    # row r * k_max + j of `flat` is view r + j * world: ONE index_select puts the views in pose order
    v = torch.arange(n_total, device=local.device)
    return flat.index_select(0, (v % world) * k_max + v // world)


def _has_into_tensor(group=None):
    return hasattr(dist, "all_gather_into_tensor") and dist.get_backend(group) == "nccl"


def render_path_distributed(render_fn, render_poses, savedir=None, object_id=2, group=None, writer=None,
                            gather_disp=True):
    """render_path (RN:213-255) over all ranks.  `render_fn(poses [k,4,4]) -> (rgb [k,H,W,3], disp [k,H,W])`
    tensors on this rank's device (production: NsrModel.render_views; tests: any deterministic function).
    Returns (rgbs, disps) numpy arrays in pose order on every rank; savedir/<object_id>/%03d.png are written by the
    rank that rendered them (pose index in the name, as RN:248-249).  gather_disp=False: a caller that drops the
    disparities (NM:128 does) skips their all-gather and gets None.
    NSR_DIST_TIMING=1: the seconds spent in render / png / gather_rgb / gather_disp (max over ranks, device-synchronised
    per phase) are left in dist.LAST_TIMINGS and printed by rank 0 -- the numbers a first multi-GPU run is compared with
    (per rank ceil(K / N) views x 0.33 s of rendering; every other phase must stay in the milliseconds)."""
    from .run_nerf_helpers import to8b
    from . import png
    world, rank = world_info(group)
    timing = os.environ.get("NSR_DIST_TIMING", "0") == "1"
    marks = {}

    def mark(name, t0):
        if timing:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            marks[name] = marks.get(name, 0.0) + time.perf_counter() - t0
        return time.perf_counter()
    poses = torch.as_tensor(render_poses, dtype=torch.float32).detach()
    n = poses.shape[0]
    check_distinct_devices(group)
    check_same_poses(poses, group)
    mine = shard_indices(n, world, rank)
    t0 = time.perf_counter()
    rgb, disp = render_fn(poses[mine])
    t0 = mark("render", t0)
    if savedir is not None:                      # every rank writes its own views (one node, one file system):
        d = os.path.join(savedir, str(object_id))         # 6.5 ms of zlib per 400x400 PNG would otherwise serialise
        os.makedirs(d, exist_ok=True)                     # on rank 0 (100 views: 0.65 s against 4.2 s of rendering)
        local = rgb.cpu().numpy()
        names = [os.path.join(d, "{:03d}.png".format(i)) for i in mine]
        if writer is None:
            png.imwrite_many(names, [to8b(local[k]) for k in range(len(mine))])
        else:
            for k, name in enumerate(names):
                writer(name, to8b(local[k]))
    t0 = mark("png", t0)
    rgbs = gather_views(rgb, n, group).cpu().numpy()
    t0 = mark("gather_rgb", t0)
    disps = gather_views(disp, n, group).cpu().numpy() if gather_disp else None
    t0 = mark("gather_disp", t0)
    if savedir is not None and (world > 1 or _FORCE):
        dist.barrier(group=group)                # the files of all ranks exist when any rank returns
    if timing:
        names = ("render", "png", "gather_rgb", "gather_disp")
        buf = torch.tensor([marks.get(k, 0.0) for k in names], dtype=torch.float64, device=_comm_device(group))
        if world > 1 or _FORCE:
            dist.all_reduce(buf, op=dist.ReduceOp.MAX, group=group)
        LAST_TIMINGS.clear()
        LAST_TIMINGS.update({k: float(v) for k, v in zip(names, buf.cpu().tolist())}, views=n, ranks=world)
        if rank == 0:
            print("render_path over %d rank(s), %d views: %s" % (world, n, ", ".join("%s %.4f s" % (k, LAST_TIMINGS[k]) for k in names)))
    return rgbs, disps


def gather_handoff(ann, n_total, group=None):
    """The in-memory hand-off (handoff.annotate / render_path_inmemory) of this rank's views -> the same dict for all
    n_total views in pose order on every rank: uint8 images (0.48 MB per 400x400 view instead of 1.92 MB fp32),
    boxes, row counts and, if present, masks."""
    return {k: gather_views(v, n_total, group) for k, v in ann.items()}


def gather_patch_grads(local, n_poses, group=None):
    """render_path_grad's list of per-patch [n_cat] gradients (RN:190) when poses are sharded: `local`
    [k_local, n_patches, n_cat] for this rank's poses (shard order) -> [n_poses, n_patches, n_cat] in pose order on
    every rank (the reference appends pose-major, patch-minor, which is this tensor flattened)."""
    return gather_views(local, n_poses, group)


def mean_psi_grad(local_dLdpsis, group=None, n_cat=None):
    """torch.mean(torch.stack(dLdpsis), 0) (NM:191) when the per-patch gradients are spread over ranks:
    all-reduce(sum) of [sum of local [n_cat] vectors | local count | error flag].  n_cat (the length of psi; 8 in the
    reference, NM:1164) is taken from the local gradients; a rank that holds none (fewer poses than ranks) learns it from
    the others through one all-reduce(max) of (length, -length) -- skipped when the caller supplies n_cat.  A length
    disagreement is an error on EVERY rank (the collectives are finished first: a rank that raised alone would leave the
    others hanging in the next one)."""
    world, _ = world_info(group)
    comm = world > 1 or _FORCE
    if len(local_dLdpsis):
        s = torch.stack([torch.as_tensor(g, dtype=torch.float64) for g in local_dLdpsis]).sum(0)
    else:
        s = None
    mine = s.numel() if s is not None else 0
    length = int(n_cat) if n_cat else mine
    if comm and not n_cat:
        none = 1 << 40
        t = torch.tensor([mine, -(mine if mine else none)], dtype=torch.int64, device=_comm_device(group))
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        longest, shortest = int(t[0].item()), -int(t[1].item())
        if longest and shortest != longest:
            raise ValueError("mean_psi_grad: the ranks' gradients have between %d and %d entries" % (shortest, longest))
        length = longest
    if length == 0:
        raise ValueError("mean_psi_grad: no rank holds a gradient")
    bad = bool(mine) and mine != length                 # (only possible against a caller-supplied n_cat)
    if bad and not comm:
        raise ValueError("mean_psi_grad: local gradients have %d entries, n_cat=%d" % (mine, length))
    buf = torch.zeros(length + 2, dtype=torch.float64)
    if s is not None and not bad:
        buf[:length] = s
        buf[length] = len(local_dLdpsis)
    buf[length + 1] = 1.0 if bad else 0.0
    if comm:
        buf = buf.to(_comm_device(group))
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        buf = buf.cpu()
    if buf[length + 1] > 0:
        raise ValueError("mean_psi_grad: %d rank(s) hold gradients whose length is not n_cat=%d (this rank: %d entries)"
                         % (int(buf[length + 1]), length, mine))
    return (buf[:length] / buf[length]).to(torch.float32)
