"""BASELINE config 5: all 21 YCB-V NeRFs (synthetic weights, seeds 0..20) rendered concurrently, one handle + one
HIP stream per model; under torch.distributed.run model m goes to rank m mod world.  One 400x400 view per model.
Prints one JSON object: aggregate Mray-samples/s back-to-back on one stream vs on 21 streams."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sim_nerf_amd import synthetic as S
from neural_sim_nerf_amd.engine import NsrModel

rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
torch.cuda.set_device(local)
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
H = W = int(os.environ.get("SIDE", "400"))
mine = list(range(21))[rank::world]
models, streams = [], []
for m in mine:
    sd_c = S.synth_weights(m)
    models.append(NsrModel(sd_c, S.synth_weights(1000 + m, fine_of=sd_c), device=local))
    streams.append(torch.cuda.Stream())
poses = torch.as_tensor(S.sweep_poses(21, seed=3)[:, :3, :4], device=models[0].device)
K = S.scaled_K(400.0 / H)

def run(concurrent):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = []
    for i, (m, st) in enumerate(zip(models, streams)):
        if concurrent:
            with torch.cuda.stream(st):
                outs.append(m.render_views(poses[mine[i]], H, W, K, S.YCBV_NEAR, S.YCBV_FAR)["rgb_map"])
        else:
            outs.append(m.render_views(poses[mine[i]], H, W, K, S.YCBV_NEAR, S.YCBV_FAR)["rgb_map"])
    torch.cuda.synchronize()
    return time.perf_counter() - t0, outs

run(False)
t_seq, o_seq = run(False)
t_con, o_con = run(True)
same = all(torch.equal(a, b) for a, b in zip(o_seq, o_con))
if world > 1:
    t = torch.tensor([t_seq, t_con], device=models[0].device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    t_seq, t_con = (float(v) for v in t)
if rank == 0:
    n = 21 * H * W * 192 / 1e6
    print(json.dumps({"models": 21, "n_gpus": world, "view": "%dx%d, 64+128" % (H, W), "one_stream_s": round(t_seq, 4),
                      "streams_s": round(t_con, 4), "Mray_samples_per_s_one_stream": round(n / t_seq, 2),
                      "Mray_samples_per_s_streams": round(n / t_con, 2), "identical_images": bool(same)}))
if world > 1:
    dist.destroy_process_group()
