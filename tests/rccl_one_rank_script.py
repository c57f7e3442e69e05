"""Run under `python -m torch.distributed.run --nproc-per-node 1` with NSR_DIST_FORCE_COLLECTIVES=1 on a one-GPU box: every
collective wrapper of neural_sim_nerf_amd/dist.py goes through RCCL (backend "nccl") for real -- device buffers, the dtypes
the render path gathers (uint8 images, fp32 images / disparities / gradients, int32 boxes), all_gather_into_tensor,
broadcast, all_reduce -- with ONE rank, where each must be the identity.  Prints RCCL-ONE-RANK-OK on success.
(tests/test_gpu_parity.py::test_collective_wrappers_over_rccl_with_one_rank)"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    assert os.environ.get("NSR_DIST_FORCE_COLLECTIVES") == "1"
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from neural_sim_nerf_amd import dist as D
    from neural_sim_nerf_amd import synthetic as S
    from neural_sim_nerf_amd.engine import NsrModel
    assert D._FORCE and dist.get_backend() == "nccl" and D.world_info() == (1, 0)
    dev = torch.device("cuda", local)
    g = torch.Generator(device=dev).manual_seed(0)
    for t in (torch.rand(5, 8, 8, 3, device=dev, generator=g), (torch.rand(5, 8, 8, 3, device=dev, generator=g) * 255).to(torch.uint8),
              torch.randint(0, 400, (5, 4), device=dev, generator=g, dtype=torch.int32), torch.rand(5, 4, 8, device=dev, generator=g)):
        out = D.gather_views(t, 5)
        assert out.is_cuda and out.dtype == t.dtype and torch.equal(out, t), t.dtype
    assert torch.equal(D.gather_patch_grads(torch.arange(24.0, device=dev).reshape(3, 2, 4), 3), torch.arange(24.0, device=dev).reshape(3, 2, 4))
    ann = dict(rgb8=torch.zeros(2, 8, 8, 3, dtype=torch.uint8, device=dev), boxes=torch.ones(2, 4, dtype=torch.int32, device=dev))
    got = D.gather_handoff(ann, 2)
    assert all(torch.equal(got[k], ann[k]) for k in ann)
    m = D.mean_psi_grad([torch.arange(8.0), torch.ones(8)])
    assert torch.allclose(m, (torch.arange(8.0) + 1) / 2)
    D.check_same_poses(torch.eye(4)[None].repeat(3, 1, 1))
    D.check_distinct_devices()
    # render_path over the group with the real renderer: 3 small views, PNGs, timing all-reduce
    os.environ["NSR_DIST_TIMING"] = "1"
    sd_c = S.synth_weights(0)
    sd_f = S.synth_weights(1000, fine_of=sd_c)
    model = NsrModel(sd_c, sd_f, device=local)
    K = S.scaled_K(25.0)
    poses = torch.as_tensor(np.stack(S.sweep_poses(3, 1)))

    def render_fn(p):
        o = model.render_views(p[:, :3, :4].to(dev), 16, 16, K, S.YCBV_NEAR, S.YCBV_FAR)
        return o["rgb_map"].reshape(-1, 16, 16, 3), o["disp_map"].reshape(-1, 16, 16)
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        rgbs, disps = D.render_path_distributed(render_fn, poses, savedir=tmp, object_id=7)
        assert rgbs.shape == (3, 16, 16, 3) and disps.shape == (3, 16, 16) and os.path.exists(os.path.join(tmp, "7", "002.png"))
    want, _ = render_fn(poses)
    assert np.array_equal(rgbs, want.cpu().numpy())
    assert set(D.LAST_TIMINGS) >= {"render", "png", "gather_rgb", "gather_disp"} and D.LAST_TIMINGS["ranks"] == 1
    model.close()
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL-ONE-RANK-OK")


if __name__ == "__main__":
    main()
