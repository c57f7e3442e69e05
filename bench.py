#!/usr/bin/env python
"""bench.py -- Mray-samples/s of the NeRF volumetric render path on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of synthetic input = ONE 400x400 view rendered with 64 coarse
+ 128 importance samples per ray through the 8x256 NeRF MLP pair (BASELINE.json configs[1]), fp32 end to end,
rays generated in-kernel from the camera, outputs (rgb/disp/acc/rgb0/disp0/acc0/z_std) written to HBM.
With N GPUs every rank renders its own views (view sharding, no data-path collective); the rendered images
are all-gathered over RCCL once, at the outer-loop boundary, inside the timed region (BASELINE configs[2]).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  Extra objects:
  roofline     -- dominant kernel nsr::k_render16: algorithmic FLOP per launch / HIP-event kernel time, against
                  the fp32-input MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md) -- the datatype actually issued;
  cpu_baseline -- the oracle (CPU restatement of the reference path, "port") timed on this host's cores on a
                  bounded sample (a smaller view of the same scene, ~15 s of CPU work), rank 0, N=1 only;
  parity       -- PSNR / max-abs of the GPU render vs the oracle on that sample, and the exact-match rate of
                  the resampling indices.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from neural_sim_nerf_amd import synthetic as S          # noqa: E402
from neural_sim_nerf_amd.engine import NsrModel         # noqa: E402

H = W = 400
SAMPLES_PER_RAY = 64 + 128                 # the metric's unit
EVALS_PER_RAY = 64 + 192                   # network evaluations per ray (RN:477-483)
FLOP_PER_RAY = EVALS_PER_RAY * S.FLOP_PER_POINT          # 303 824 896
PEAK_F32_MFMA_TFLOPS = 157.3               # v_mfma_f32_32x32x2_f32, dense (MI355X_MICROARCH.md)


def cpu_baseline_and_parity(model, sd_c, sd_f, c2w):
    """Oracle on a bounded sample (64x64 view, 64+128) -- the checker, timed; never the thing shipped."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nerf_oracle as O
    run = lambda side: O.render(sd_c, sd_f, side, side, S.scaled_K(400.0 / side), c2w=c2w[:3, :4], near=S.YCBV_NEAR,
                                far=S.YCBV_FAR, chunk=4096)
    # Be fair to the CPU: calibrate the backend (MLP/encoding on torch-CPU ops like the reference, or numpy+OpenBLAS)
    # and the thread count on a 32x32 view, then time the sample with the fastest setting.
    ncpu = os.cpu_count() or 1
    best = None
    for backend, threads in [("torch", t) for t in sorted({min(ncpu, t) for t in (8, 16, 32, 64, 128, ncpu)})] + [("numpy", 0)]:
        O.set_backend(backend)
        if threads:
            torch.set_num_threads(threads)
        run(16)                                  # warm-up (thread pools, page faults)
        t0 = time.perf_counter()
        run(32)
        t32 = time.perf_counter() - t0
        if best is None or t32 < best[0]:
            best = (t32, backend, threads)
    t32, backend, threads = best
    O.set_backend(backend)
    if threads:
        torch.set_num_threads(threads)
    n_threads = threads if threads else min(ncpu, 64)        # scipy-openblas is built with MAX_THREADS=64
    side = int(min(160, max(48, 16 * round(32 * (15.0 / t32) ** 0.5 / 16))))      # aim at ~15 s of CPU work
    t0 = time.perf_counter()
    ref = run(side)
    dt = time.perf_counter() - t0
    O.set_backend("numpy")
    got = model.render_views(c2w, side, side, S.scaled_K(400.0 / side), S.YCBV_NEAR, S.YCBV_FAR, debug=True)
    rgb = got["rgb_map"].cpu().numpy().reshape(side, side, 3)
    rgb0 = got["rgb0"].cpu().numpy().reshape(side, side, 3)
    # resampling indices (the bit-exact quantity): the oracle's sample_pdf on the kernel's own coarse weights
    n = side * side
    z = O.coarse_z(np.full(n, S.YCBV_NEAR, np.float32), np.full(n, S.YCBV_FAR, np.float32))
    zs, inds, _ = O.sample_pdf((np.float32(0.5) * (z[:, 1:] + z[:, :-1])).astype(np.float32),
                               got["weights0"].cpu().numpy()[:, 1:-1])
    inds_match = float((got["inds"].cpu().numpy() == inds).mean())
    zs_match = float((got["z_samples"].cpu().numpy() == zs).mean())

    def maxabs(key, shape):
        a, b = got[key].cpu().numpy().reshape(shape), ref[key].reshape(shape)
        with np.errstate(invalid="ignore"):
            return float(np.nanmax(np.abs(a - b)))
    # PSNR delta against a pseudo ground truth T = oracle + N(0, 0.01^2)
    T = ref["rgb_map"] + np.random.RandomState(0).normal(0, 0.01, ref["rgb_map"].shape).astype(np.float32)
    cpu = {"value": round(side * side * SAMPLES_PER_RAY / dt / 1e6, 5), "unit": "Mray-samples/s", "cores": n_threads,
           "kind": "port", "sample": "one %dx%d view (%d rays x (64+128) samples, same scene, camera and networks), "
           "oracle/nerf_oracle.py, fastest of {torch-CPU ops x thread counts, numpy+OpenBLAS} on this host: %s backend, "
           "%d threads; %.1f s" % (side, side, side * side, backend, n_threads, dt)}
    par = {"psnr_vs_oracle_db": round(O.psnr(rgb, ref["rgb_map"]), 2),
           "psnr_delta_db": round(abs(O.psnr(rgb, T) - O.psnr(ref["rgb_map"], T)), 4),
           "max_abs_rgb_coarse": float(np.abs(rgb0 - ref["rgb0"]).max()),
           "mean_abs_rgb": float(np.abs(rgb - ref["rgb_map"]).mean()),
           "max_abs_rgb": float(np.abs(rgb - ref["rgb_map"]).max()),
           "max_abs_acc": maxabs("acc_map", (side, side)), "max_abs_disp": maxabs("disp_map", (side, side)),
           "max_abs_z_std": maxabs("z_std", (side, side)),
           "inds_exact_match_rate": inds_match, "z_samples_exact_match_rate": zs_match,
           "sample": "%dx%d view; indices/samples: oracle sample_pdf on the kernel's own coarse weights" % (side, side)}
    return cpu, par


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or "RANK" in os.environ:         # under torch.distributed.run, also for a single rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    sd_c = S.synth_weights(0)
    sd_f = S.synth_weights(1000, fine_of=sd_c)
    model = NsrModel(sd_c, sd_f, device=local)
    n_total = args.warmup + args.steps
    poses = S.sweep_poses(n_total * world, seed=0)[rank::world]      # view i -> rank i mod world
    poses_d = torch.as_tensor(poses[:, :3, :4], device=model.device)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        model.render_views(poses_d[i], H, W, S.YCBV_K, S.YCBV_NEAR, S.YCBV_FAR)
    if dist is not None:          # untimed: bring up the RCCL channels the timed all_gather will use (same shape)
        dummy = torch.zeros((args.steps, H * W, 3), device=model.device)
        dist.all_gather([torch.empty_like(dummy) for _ in range(world)], dummy)
        del dummy
    barrier()
    kernel_ms = []
    images = []
    t0 = time.perf_counter()
    for i in range(args.warmup, n_total):
        out = model.render_views(poses_d[i], H, W, S.YCBV_K, S.YCBV_NEAR, S.YCBV_FAR)
        kernel_ms.append(model.last_kernel_ms())      # HIP events on the launch stream (syncs on the stop event)
        images.append(out["rgb_map"])
    if dist is not None:                                # outer-loop boundary: gather the rendered images
        mine = torch.stack(images, 0)
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=model.device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        rays = args.steps * H * W * world
        value = rays * SAMPLES_PER_RAY / dt / 1e6
        k_ms = float(np.mean(kernel_ms))
        achieved = H * W * FLOP_PER_RAY / (k_ms * 1e-3) / 1e12
        traffic = None      # HBM bytes per launch from the committed PMC passes of this same command (profiles/)
        prof = os.path.join(ROOT, "profiles", "r01", "pmc_k_render.json")
        if os.path.exists(prof):
            traffic = json.load(open(prof))["x16_default"]["derived"]["hbm_traffic_bytes_per_launch"]
        line = {
            "metric": "Mray-samples/sec at 400x400, 64+128 samples, 8x256 MLP",
            "value": round(value, 3), "unit": "Mray-samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "YCB-V object-2 camera, 400x400 view per step per GPU, 64 coarse + 128 fine "
                                   "samples/ray, 8x256 NeRF MLP pair (seeded synthetic weights), fp32, fused "
                                   "persistent kernel (x16: 2 workgroups per CU), rays generated in-kernel",
                       "rays_per_step_per_gpu": H * W, "mlp_evals_per_ray": EVALS_PER_RAY,
                       "parallelism": "views sharded over %d GPU(s), image all-gather at the end" % world},
            "rays_per_s": round(rays / dt, 1), "mlp_evals_per_s": round(rays * EVALS_PER_RAY / dt, 1),
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                         "traffic_note": "bytes/launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from rocprofv3 --pmc passes "
                                         "(profiles/r01/pmc_k_render.json); L2 misses of the weight streams (4.6 MiB of "
                                         "networks vs 4 MiB L2 per XCD), served by Infinity Cache at 0.28 TB/s, not by "
                                         "HBM; a schedule with 2.9e9 exists and is 2.4 % slower (DESIGN.md, chunk queue); "
                                         "algorithmic HBM bytes are 7.0e6 per launch",
                         "kernel": "nsr::k_render16", "kernel_ms": round(k_ms, 3),
                         "flop_per_launch": H * W * FLOP_PER_RAY},
        }
        if world == 1 and not args.no_cpu_baseline:
            cpu, par = cpu_baseline_and_parity(model, sd_c, sd_f, poses[args.warmup])
            line["cpu_baseline"] = cpu
            line["parity"] = par
        print(json.dumps(line))
    model.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
