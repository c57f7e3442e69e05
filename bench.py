#!/usr/bin/env python
"""bench.py -- Mray-samples/s of the NeRF volumetric render path on MI355X (BASELINE.json metric).

Default workload (`--workload view400`, the one the driver records): a "step" = one pass of the hot path over one
batch of synthetic input = ONE 400x400 view rendered with 64 coarse + 128 importance samples per ray through the
8x256 NeRF MLP pair (BASELINE.json configs[1]), fp32 end to end, rays generated in-kernel from the camera, outputs
(rgb/disp/acc/rgb0/disp0/acc0/z_std) written to HBM.  With N GPUs every rank renders its own views (view sharding,
no data-path collective); the rendered images are all-gathered over RCCL once, at the outer-loop boundary, inside
the timed region (BASELINE configs[2]).

    python bench.py                                   # N=1, 3 steps, 1 warm-up
    python bench.py --gpus N --steps K --warmup W     # self-launches N ranks (torch.distributed.run) when not already
                                                      # under a launcher; under the driver's launcher it just runs
    python bench.py --gpus 8 --workload sweep100 --steps 1      # BASELINE configs[2]: 100-view pose sweep
    python bench.py --gpus 8 --workload models21 --steps 1      # BASELINE configs[4]: 21 models, one stream each

Prints ONE JSON line on rank 0.  Objects beside the contract fields:
  roofline       -- dominant kernel nsr::k_render16p (k_render16 with the global-phases schedule): algorithmic FLOP per launch / HIP-event kernel time, against the
                    fp32-input MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md) -- the datatype actually issued;
                    `traffic` comes from a rocprofv3 --pmc profile ONLY if that profile was collected from exactly
                    the kernel sources in this tree (hash check), else null;
  roofline_vjp   -- the same for nsr::k_render_vjp16p (render_path_grad's forward+input-gradient launch, config 4's render leg);
  cpu_baseline   -- the oracle (CPU restatement of the reference path, "port") timed on this host's cores on a bounded
                    sample (a smaller view of the same scene, ~15 s of CPU work), rank 0, N=1 only;
  parity         -- PSNR / max-abs of the GPU render vs the oracle on that sample, exact-match rate of the indices;
  extra_workloads.handoff -- f-3: device to8b + bbox extraction of 100 images next to the host route (oracle) on 5;
  extra_workloads.layered -- the layered renderer (include/nsr_wide.h) on the same view, and on an 8 x 512 network;
  extra_workloads.config1 -- BASELINE configs[0] (64x64, 64 coarse samples only): GPU throughput next to the oracle at
                    chunk 512 (the reference's config) and 4096;
  ranks_seen, kernel_ms_per_rank -- what RCCL actually saw (sum of ones over ranks; per-rank kernel time spread).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL otherwise fails in hipIpcGetMemHandle); the launcher's environment
# normally carries it -- set here too, before the HIP runtime initialises, so that a rank started without it still comes up
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np          # noqa: E402
import torch                # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from neural_sim_nerf_amd import synthetic as S          # noqa: E402
from neural_sim_nerf_amd import dist as D               # noqa: E402
from neural_sim_nerf_amd import _lib                    # noqa: E402
from neural_sim_nerf_amd.engine import NsrModel, DEFAULT_MLP         # noqa: E402

H = W = 400
SAMPLES_PER_RAY = 64 + 128                 # the metric's unit
EVALS_PER_RAY = 64 + 192                   # network evaluations per ray (RN:477-483)
FLOP_PER_RAY = EVALS_PER_RAY * S.FLOP_PER_POINT          # 303 824 896
FLOP_PER_RAY_VJP = (EVALS_PER_RAY + 192) * S.FLOP_PER_POINT   # + 192 transposed evaluations of the fine net: 531.7 M
PEAK_F32_MFMA_TFLOPS = 157.3               # v_mfma_f32_32x32x2_f32 / 16x16x4, dense (MI355X_MICROARCH.md)
PEAK_BF16_MFMA_TFLOPS = 2500.0             # v_mfma_f32_32x32x16_bf16, dense (MI355X_MICROARCH.md: ~2.5 PF)
# bf16x3 (NSR_FLAG_MLP_BF16X3): MFMA FLOP issued per point = 6 piece products x the k16-padded layer shapes
B3_ISSUED_FLOP_PER_POINT = 2 * 6 * (256 * 64 + 4 * 256 * 256 + 256 * (64 + 256) + 2 * 256 * 256 + 256 * 256 + 128 * 320)
# ... and per transposed evaluation (4-block zero-padded encoding GEMMs + 8-block GEMMs, csrc/nsr_b3.inc)
B3_ISSUED_FLOP_PER_POINT_BWD = 2 * 6 * ((128 + 256) * 128 + 256 * 256 + 2 * 256 * 256 + (128 + 256) * 256 + 4 * 256 * 256 + 128 * 256)
METRIC = "Mray-samples/sec at 400x400, 64+128 samples, 8x256 MLP"
MLP_MODES = ("fp32", "bf16x3", "f16x2")    # layer-GEMM arithmetics of the forward kernels (engine.NsrModel(mlp=...))
# f16x2 (NSR_FLAG_MLP_F16X2): 3 piece products x the k16-padded layer shapes (csrc/nsr_h2.inc); fp16 MFMA peak = bf16's
H2_ISSUED_FLOP_PER_POINT = 2 * 3 * (256 * 64 + 4 * 256 * 256 + 256 * (64 + 256) + 2 * 256 * 256 + 256 * 256 + 128 * 288)
# ... and per transposed evaluation (2-block encoding GEMMs + 8-block GEMMs, csrc/nsr_h2_bwd.inc)
H2_ISSUED_FLOP_PER_POINT_BWD = 2 * 3 * ((64 + 256) * 128 + 256 * 256 + 2 * 256 * 256 + (64 + 256) * 256 + 4 * 256 * 256 + 64 * 256)
MLP_DTYPE = {"bf16x3": "bf16x3: fp32 operands as three bf16 pieces each, six piece products per fp32 product, fp32 accumulate",
             "f16x2": "f16x2: fp32 operands as two fp16 pieces each (power-of-two range management), three piece products per "
                      "fp32 product, fp32 accumulate"}


def forward_kernel_name(model):
    if model.mlp == "bf16x3":
        return "nsr::k_render_b3"
    if model.mlp == "f16x2":
        return "nsr::k_render_h2"
    if model.variant == 32:
        return "nsr::k_render"
    return "nsr::k_render16p" if model.schedule == "phases" else "nsr::k_render16"


def forward_roofline(model, k_ms, n_rays=H * W):
    """roofline object of the forward render kernel for one launch of n_rays rays taking k_ms (HIP events)."""
    ach = n_rays * FLOP_PER_RAY / (k_ms * 1e-3) / 1e12
    r = {"bound": "mfma", "achieved": round(ach, 2), "unit": "TFLOP/s", "kernel": forward_kernel_name(model),
         "kernel_ms": round(k_ms, 3), "flop_per_launch": n_rays * FLOP_PER_RAY}
    if model.mlp in ("bf16x3", "f16x2"):
        per_point, n_prod, dt = ((B3_ISSUED_FLOP_PER_POINT, "six", "bf16") if model.mlp == "bf16x3" else
                                 (H2_ISSUED_FLOP_PER_POINT, "three", "fp16"))
        issued = n_rays * EVALS_PER_RAY * per_point / (k_ms * 1e-3) / 1e12
        r.update({"peak": PEAK_BF16_MFMA_TFLOPS, "frac": round(ach / PEAK_BF16_MFMA_TFLOPS, 4),
                  "frac_note": "frac = achieved (ALGORITHMIC fp32 FLOP/s) / peak of the datatype issued; issued_frac = the MFMA "
                               "work actually issued (piece products) / the same peak",
                  "issued": round(issued, 1), "issued_frac": round(issued / PEAK_BF16_MFMA_TFLOPS, 4),
                  # a scheme of n piece products per fp32 product cannot deliver more than 1/n of the piece datatype's peak
                  "attainable_frac": round(1.0 / (6 if model.mlp == "bf16x3" else 3), 4),
                  "frac_of_attainable": round(ach / PEAK_BF16_MFMA_TFLOPS * (6 if model.mlp == "bf16x3" else 3), 4),
                  "achieved_over_fp32_mfma_peak": round(ach / PEAK_F32_MFMA_TFLOPS, 3),
                  "note": "`achieved` counts ALGORITHMIC fp32 FLOP (as for the fp32 kernels); the datatype issued is %s "
                          "(peak 2.5 PFLOP/s dense): every fp32 product costs %s %s piece products, so the kernel "
                          "issues `issued` TFLOP/s of %s MFMA work = `issued_frac` of that peak, and delivers "
                          "`achieved_over_fp32_mfma_peak` x what the fp32 MFMA pipe could at 100 %%" % (dt, n_prod, dt, dt)})
    else:
        r.update({"peak": PEAK_F32_MFMA_TFLOPS, "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4)})
    return r


def alt_mlp_workload(mlp, sd_c, sd_f, device, c2w, ref, launches=3, sample=None, pmc_file=None):
    """extra_workloads.<mlp>: the SAME 400x400 view through a forward kernel with another layer-GEMM arithmetic --
    "bf16x3" (k_render_b3: bf16 MFMAs, fp32 operands split exactly into three bf16 pieces), "f16x2" (k_render_h2: fp16
    MFMAs, two fp16 pieces, power-of-two range management) or "fp32" (k_render16p, when the main line is one of the
    others) -- timed like the main line (HIP events per launch + wall clock), compared with the main line's image `ref`
    of that view, and with the same `parity` object against the same oracle render."""
    m = NsrModel(sd_c, sd_f, device=device, mlp=mlp)
    pose = torch.as_tensor(c2w[:3, :4], device=m.device)
    out = m.render_views(pose, H, W, S.YCBV_K, S.YCBV_NEAR, S.YCBV_FAR)            # warm-up
    torch.cuda.synchronize()
    ms = []
    t0 = time.perf_counter()
    for _ in range(launches):
        out = m.render_views(pose, H, W, S.YCBV_K, S.YCBV_NEAR, S.YCBV_FAR)
        ms.append(m.last_kernel_ms())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k_ms = float(np.mean(ms))
    d = {k: (out[k] - ref[k]).abs() for k in ("rgb_map", "acc_map", "rgb0")}
    mse = float(((out["rgb_map"] - ref["rgb_map"]).double() ** 2).mean())
    res = {"value": round(launches * H * W * SAMPLES_PER_RAY / dt / 1e6, 3), "unit": "Mray-samples/s",
           "ms_per_view": round(dt / launches * 1e3, 3), "launches": launches,
           "dtype": MLP_DTYPE.get(mlp, "f32"),
           "roofline": forward_roofline(m, k_ms),
           "vs_main_line_kernel_same_view": {
               "psnr_db": round(-10.0 * np.log10(mse), 2) if mse > 0 else None,
               "rgb_max_abs": float(d["rgb_map"].max()), "acc_max_abs": float(d["acc_map"].max()),
               "rgb0_max_abs": float(d["rgb0"].max()),
               "rays_with_rgb_diff_above_1e-4": int((d["rgb_map"].max(-1).values > 1e-4).sum())},
           "how_to_enable": "NsrModel(..., mlp='%s') / NSR_MLP=%s / bench.py --mlp %s" % (mlp, mlp, mlp)}
    res["roofline_vjp"] = vjp_roofline(m, c2w, pmc_file)
    if sample is not None:                   # the same `parity` object as the main line's, against the same oracle output
        res["parity"] = parity_vs_oracle(m, sample)
    m.close()
    return res


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nerf_oracle as O
    return O


def _fastest_cpu_setting(O, run):
    """Be fair to the CPU: calibrate the backend (MLP/encoding on torch-CPU ops like the reference, or
    numpy+OpenBLAS) and the thread count on a small case; returns (seconds, backend, threads)."""
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
    return best


def cpu_baseline_and_parity(model, sd_c, sd_f, c2w, side=400):
    """The oracle -- the checker, timed; never the thing shipped -- on ONE view of the metric's own configuration
    (400x400, 64+128: SURVEY.md 8d; ~2-3 min of CPU time on the GPU boxes' hosts; `--cpu-sample-side` shrinks the view).
    Its render, kept with the intermediates the census reads, is what every `parity` object of the line is computed
    against."""
    O = _oracle()
    run = lambda sd, extras=False: O.render(sd_c, sd_f, sd, sd, S.scaled_K(400.0 / sd), c2w=c2w[:3, :4], near=S.YCBV_NEAR,
                                            far=S.YCBV_FAR, chunk=4096, extras=extras)
    ncpu = os.cpu_count() or 1
    t32, backend, threads = _fastest_cpu_setting(O, run)
    O.set_backend(backend)
    if threads:
        torch.set_num_threads(threads)
    n_threads = threads if threads else min(ncpu, 64)        # scipy-openblas is built with MAX_THREADS=64
    t0 = time.perf_counter()
    ref = run(side, True)
    dt = time.perf_counter() - t0
    O.set_backend("numpy")
    ref = {k: v.reshape((side * side,) + v.shape[2:]) for k, v in ref.items() if k not in ("raw", "weights", "cdf")}
    ref["sigma0_last"] = ref.pop("raw0")[:, -1, 3].copy()
    cpu = {"value": round(side * side * SAMPLES_PER_RAY / dt / 1e6, 5), "unit": "Mray-samples/s", "cores": n_threads,
           "kind": "port", "cpu_model": cpu_model(), "host_logical_cpus": ncpu,
           "sample": "one %s%dx%d view (%d rays x (64+128) samples, same scene, camera and networks), "
           "oracle/nerf_oracle.py, fastest of {torch-CPU ops x thread counts, numpy+OpenBLAS} on this host: %s backend, "
           "%d threads; %.1f s" % ("full " if side == H else "", side, side, side * side, backend, n_threads, dt)}
    sample = {"ref": ref, "side": side, "c2w": c2w, "nets": (sd_c, sd_f)}
    return cpu, parity_vs_oracle(model, sample), (backend, threads), sample


def parity_vs_oracle(model, sample):
    """`parity` object: `model` on the view of cpu_baseline_and_parity against the oracle's render of it, end to end.
    `census` (oracle/census.py) counts the rays beyond 1e-4 on rgb / acc and proves for each that it sits on one of the
    reference's own discontinuities (sigma_last cliff, resampling index, denominator switch) or in its 1/denom
    conditioning; `unattributed` must be 0.  PSNR-delta is SURVEY.md 8d's, on the whole view, cliffs included."""
    O = _oracle()
    import census as C
    ref, side, c2w = sample["ref"], sample["side"], sample["c2w"]
    K = S.scaled_K(400.0 / side)
    got = model.render_views(c2w, side, side, K, S.YCBV_NEAR, S.YCBV_FAR, debug=True)
    taps = {k: got[k].cpu().numpy() for k in ("rgb_map", "acc_map", "disp_map", "rgb0", "acc0", "raw0", "weights0", "inds",
                                              "z_samples", "z_fine", "raw", "z_std")}
    del got
    n = side * side
    # resampling indices (the bit-exact quantity): the oracle's sample_pdf on the kernel's own coarse weights
    z = O.coarse_z(np.full(n, S.YCBV_NEAR, np.float32), np.full(n, S.YCBV_FAR, np.float32))
    zs, inds, _ = O.sample_pdf((np.float32(0.5) * (z[:, 1:] + z[:, :-1])).astype(np.float32), taps["weights0"][:, 1:-1])
    inds_match = float((taps["inds"] == inds).mean())
    zs_match = float((taps["z_samples"] == zs).mean())
    ro, rd = O.get_rays(side, side, K, c2w[:3, :4])
    c = C.census(sample["nets"], ro.reshape(-1, 3), rd.reshape(-1, 3), S.YCBV_NEAR, S.YCBV_FAR, taps, ref)

    def maxabs(key):
        with np.errstate(invalid="ignore"):
            return float(np.nanmax(np.abs(taps[key] - ref[key])))
    rgb = taps["rgb_map"]
    return {"psnr_vs_oracle_db": c["psnr_vs_oracle_db"], "psnr_delta_db": c["psnr_delta_db"],
            "psnr_delta_db_excluding_attributed_rays": c["psnr_delta_db_excluding_attributed"],
            "max_abs_rgb_coarse": maxabs("rgb0"), "mean_abs_rgb": float(np.abs(rgb - ref["rgb_map"]).mean()),
            "max_abs_rgb": maxabs("rgb_map"), "max_abs_acc": maxabs("acc_map"), "max_abs_disp": maxabs("disp_map"),
            "max_abs_z_std": maxabs("z_std"),
            "inds_exact_match_rate": inds_match, "z_samples_exact_match_rate": zs_match,
            "census": {k: c[k] for k in ("rays", "tol", "rays_above_tol", "cliff_rays", "fine_cliff_rays", "coarse_cliff_rays",
                                         "index_flip_rays", "denom_switch_rays", "illconditioned_shift_rays", "unattributed",
                                         "coarse_rays_above_tol", "coarse_unattributed", "rays_disp_above_tol",
                                         "inds_equal_rate_end_to_end")},
            "passes": bool(C.passes(c)),
            "sample": "%dx%d view, end to end against the oracle's render of the same rays; inds/z_samples_exact_match_rate: "
                      "the oracle's sample_pdf on the kernel's own coarse weights (the bit-exact stage)" % (side, side)}


def trained_workload(device, cpu_setting, side=160):
    """extra_workloads.trained: the end-to-end census on a TRAINED pair of networks -- trained by the
    reference's own code against an analytic scene (oracle/train_g26.py; stored in tests/golden/g26_trained.npz with the
    reference's render of them): sparse, saturated densities (sigma of several hundred, three quarters of the samples in empty
    space), opaque rays whose resampling denominators sit at the 1e-5 switch -- what the synthetic weight family of the main
    line does not have.  A side x side view of the fixture's camera, every arithmetic of the fused kernels and of the layered
    renderer against the oracle's render of the same rays (the oracle leg: checker, never the thing shipped)."""
    path = os.path.join(ROOT, "tests", "golden", "g26_trained.npz")
    if not os.path.exists(path):
        return {"error": "tests/golden/g26_trained.npz is missing"}
    from neural_sim_nerf_amd.wide import WideModel
    g = np.load(path)
    sd_c = {k[2:]: np.asarray(g[k], np.float32) for k in g.files if k.startswith("c.")}
    sd_f = {k[2:]: np.asarray(g[k], np.float32) for k in g.files if k.startswith("f.")}
    c2w = np.asarray(g["c2w"], np.float32)
    O = _oracle()
    if cpu_setting:
        O.set_backend(cpu_setting[0])
        if cpu_setting[1]:
            torch.set_num_threads(cpu_setting[1])
    t0 = time.perf_counter()
    ref = O.render(sd_c, sd_f, side, side, S.scaled_K(400.0 / side), c2w=c2w[:3, :4], near=S.YCBV_NEAR, far=S.YCBV_FAR, chunk=4096,
                   extras=True)
    dt = time.perf_counter() - t0
    O.set_backend("numpy")
    sig = ref["raw"][..., 3]
    stats = {"sigma_max": round(float(sig.max()), 1), "sigma_share_above_100": round(float((sig > 100).mean()), 4),
             "sigma_share_not_positive": round(float((sig <= 0).mean()), 4),
             "rays_acc_between_0.01_and_0.99": round(float(((ref["acc_map"] > 0.01) & (ref["acc_map"] < 0.99)).mean()), 4)}
    ref = {k: v.reshape((side * side,) + v.shape[2:]) for k, v in ref.items() if k not in ("raw", "weights", "cdf")}
    ref["sigma0_last"] = ref.pop("raw0")[:, -1, 3].copy()
    sample = {"ref": ref, "side": side, "c2w": c2w, "nets": (sd_c, sd_f)}
    out = {"workload": "%dx%d view of the trained pair (g26: %d Adam steps of the reference's render + img2mse on an analytic textured box), "
                       "64+128 samples, census against the oracle's render (%.1f s on the host)" % (side, side, int(g["train_steps"]), dt),
           "network": stats}

    def brief(par, model):
        c = par["census"]
        return {"rays": c["rays"], "rays_above_tol": c["rays_above_tol"], "unattributed": c["unattributed"],
                "cliff_rays": c["cliff_rays"], "index_flip_rays": c["index_flip_rays"], "denom_switch_rays": c["denom_switch_rays"],
                "illconditioned_shift_rays": c["illconditioned_shift_rays"], "psnr_delta_db": par["psnr_delta_db"],
                "max_abs_rgb": par["max_abs_rgb"], "max_abs_acc": par["max_abs_acc"], "inds_exact": par["inds_exact_match_rate"],
                "passes": par["passes"], "range_status": model.range_status()}
    for mlp in MLP_MODES:
        m = NsrModel(sd_c, sd_f, device=device, mlp=mlp)
        out[mlp] = brief(parity_vs_oracle(m, sample), m)
        m.close()
    for mlp in ("bf16x3", "fp32", "f16x2"):
        m = WideModel(sd_c, sd_f, device=device, mlp=mlp)
        out["layered-" + mlp] = brief(parity_vs_oracle(m, sample), m)
        m.close()
    out["passes"] = all(v["passes"] and v["unattributed"] == 0 for k, v in out.items() if isinstance(v, dict) and "passes" in v)
    return out


def api_overhead_workload(sd_c, sd_f, device, reps=60):
    """What the drop-in API costs on top of the engine: run_nerf_noscale.render(c2w=...) of a 64x64 view and render(rays=...)
    of a 512-ray patch (the bilevel loop's call pattern, RN:156-168) against the same launches through NsrModel.  Per call
    the API checks its kwargs, looks its native handle up -- which includes the content fingerprint of both networks'
    parameters (NeRF.weights_version: one native kernel + an 8-byte read-back per network) -- and reshapes the outputs."""
    import neural_sim_nerf_amd.run_nerf_noscale as R
    dev = torch.device("cuda", device)
    nets = []
    for sd in (sd_c, sd_f):
        n = R.NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
        n.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        nets.append(n.to(dev))
    kw = dict(network_query_fn=None, perturb=False, N_importance=128, network_fine=nets[1], N_samples=64, network_fn=nets[0],
              use_viewdirs=True, white_bkgd=False, raw_noise_std=0., ndc=False, lindisp=False, near=S.YCBV_NEAR, far=S.YCBV_FAR)
    side = 64
    K = S.scaled_K(400.0 / side)
    pose = torch.as_tensor(S.sweep_poses(1, seed=5)[0][:3, :4], device=dev)
    eng = NsrModel(sd_c, sd_f, device=device)
    ro, rd = eng.get_rays(side, side, K, pose)
    patch = torch.stack([ro.reshape(-1, 3)[:512], rd.reshape(-1, 3)[:512]], 0).contiguous()

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    with torch.no_grad():
        api_view = timed(lambda: R.render(side, side, K, chunk=32768, c2w=pose, **kw))
        eng_view = timed(lambda: eng.render_views(pose, side, side, K, S.YCBV_NEAR, S.YCBV_FAR))
        api_patch = timed(lambda: R.render(side, side, K, chunk=32768, rays=patch, **kw))
        eng_patch = timed(lambda: eng.render_rays(patch[0], patch[1], S.YCBV_NEAR, S.YCBV_FAR))
        fp = timed(lambda: nets[0].weights_version())
        # ... and the same patch through the LAYERED renderer (sample counts no fused kernel is built for: 40 + 72), drop-in
        # render(rays=...) against WideModel.render_rays = nsrw_render_rays
        from neural_sim_nerf_amd.wide import WideModel
        kwl = dict(kw, N_samples=40, N_importance=72)
        own = lambda net: {k: v.detach() for k, v in net.state_dict().items()}
        weng = WideModel(own(nets[0]), own(nets[1]), device=device, n_samples=40, n_importance=72)
        api_lay = timed(lambda: R.render(side, side, K, chunk=32768, rays=patch, **kwl))
        eng_lay = timed(lambda: weng.render_rays(patch[0], patch[1], S.YCBV_NEAR, S.YCBV_FAR))
        weng.close()
    eng.close()
    for n in nets:
        n.invalidate()
    return {"workload": "drop-in API vs engine, 64+128 samples, %d calls each, wall clock per call" % reps,
            "layered_patch_512": {"api": round(api_lay, 4), "engine": round(eng_lay, 4), "overhead_frac": round(api_lay / eng_lay - 1.0, 4),
                                  "note": "512 rays at 40 + 72 samples through the layered renderer: render(rays=...) vs WideModel.render_rays"},
            "weights_check": "every call checked; the patch form reads the fingerprint while its render runs (NSR_TRUST_VERSIONS unset)",
            "view_64x64_ms": {"api": round(api_view, 4), "engine": round(eng_view, 4),
                              "overhead_frac": round(api_view / eng_view - 1.0, 4)},
            "patch_512_rays_ms": {"api": round(api_patch, 4), "engine": round(eng_patch, 4),
                                  "overhead_frac": round(api_patch / eng_patch - 1.0, 4)},
            "weights_version_ms_per_network": round(fp, 4)}


def render_options_workload(sd_c, sd_f, c2w, device):
    """One 400x400 view through render(rays=...) with ALL the stochastic options on (perturb > 0: stratified depths and random
    resampling uniforms, raw_noise_std > 0: density noise in both passes -- RN:447-459, RH:211, RN:365-374) next to the same
    rays on the deterministic path: the per-ray extras cost 448 more floats of HBM reads per ray and a full sort of the
    unsorted importance samples, nothing inside the network passes."""
    eng = NsrModel(sd_c, sd_f, device=device)
    ro, rd = eng.get_rays(400, 400, S.YCBV_K, torch.as_tensor(c2w[:3, :4]))
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    n = ro.shape[0]
    dev = torch.device("cuda", device)
    gen = torch.Generator(device=dev).manual_seed(0)
    ex = dict(t_rand=torch.rand(n, 64, device=dev, generator=gen), noise0=torch.randn(n, 64, device=dev, generator=gen),
              u=torch.rand(n, 128, device=dev, generator=gen), noise1=torch.randn(n, 192, device=dev, generator=gen))

    def ms(extras):
        eng.render_rays(ro, rd, S.YCBV_NEAR, S.YCBV_FAR, extras=extras)
        t = []
        for _ in range(3):
            eng.render_rays(ro, rd, S.YCBV_NEAR, S.YCBV_FAR, extras=extras)
            t.append(eng.last_kernel_ms())
        return float(np.mean(t))
    plain, stoch = ms(None), ms(ex)
    eng.close()
    return {"workload": "400x400 rays, 64+128 samples, perturb + raw_noise_std via NsrRayExtras (kernel ms, HIP events)",
            "deterministic_ms": round(plain, 3), "stochastic_ms": round(stoch, 3), "ratio": round(stoch / plain, 4),
            "extra_hbm_bytes_per_ray": 448 * 4}


def layered_workload(sd_c, sd_f, c2w, device, ref):
    """extra_workloads.layered: the layered renderer (include/nsr_wide.h: one MFMA GEMM per network layer, activations in HBM)
    -- what serves the networks and sample counts the fused kernels are not built for.  The SAME view and networks as the main
    line through it (the price of leaving the fused kernels, and a cross-check of two independent implementations of the
    path), and a network the fused kernels cannot hold (8 x 512), on each of its three arithmetics.  Roofline of its GEMM
    kernels: algorithmic fp32 FLOP of the network evaluations / HIP-event time of the whole launch call (per-ray stages
    included) against the dense MFMA peak of the datatype issued."""
    from neural_sim_nerf_amd.wide import WideModel
    pose = torch.as_tensor(c2w[:3, :4])
    out = {"workload": "one 400x400 view, 64+128 samples, through nsrw_render_rays (HIP-event ms of the launch call, all chunks)",
           "kernel": "nsrw::kw_gemm_h2<4, *, 4> (r06, the default: fp32 in HBM, every operand as two fp16 pieces, three piece products on "
                     "v_mfma_f32_32x32x16_f16, fp32 accumulate, fp32 out; a pass that leaves fp16's range is re-run on bf16x3 inside the "
                     "call); bf16x3: nsrw::kw_gemm_b3<4, *, 4> (three bf16 pieces, six products, no range); fp32: nsrw::kw_gemm_f32<4, *, 4> "
                     "(the same kernel body on v_mfma_f32_32x32x2_f32: the strict mode)",
           "peak": PEAK_BF16_MFMA_TFLOPS, "peak_note": "dense fp16 = bf16 MFMA; `frac` = algorithmic FLOP / time / peak, `issued_frac` = 3 x "
           "(f16x2) or 6 x (bf16x3) that -- the piece products actually issued: the ceiling of `frac` is 1/3 resp. 1/6; "
           "`frac_of_fp32_mfma_peak` puts the same algorithmic rate over the 157.3 TFLOP/s the fp32 MFMAs could reach at most"}

    def run(model, flop_per_point, launches=2):
        ro, rd = model.get_rays(H, W, S.YCBV_K, pose)
        ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
        o = model.render_rays(ro, rd, S.YCBV_NEAR, S.YCBV_FAR)
        ms = []
        for _ in range(launches):
            o = model.render_rays(ro, rd, S.YCBV_NEAR, S.YCBV_FAR)
            ms.append(model.last_kernel_ms())
        t = float(np.mean([m[0] for m in ms]))
        flop = H * W * 256 * flop_per_point
        tf = flop / t / 1e9
        r = {"ms_per_view": round(t, 2), "chunks_of_rays": ms[-1][1], "Mray_samples_per_s": round(H * W * 192 / t / 1e3, 2),
             "achieved": round(tf, 1), "unit": "TFLOP/s", "mlp": model.mlp, "frac_of_fp32_mfma_peak": round(tf / PEAK_F32_MFMA_TFLOPS, 4)}
        if model.mlp.endswith("bf16x3"):
            r.update(frac=round(tf / PEAK_BF16_MFMA_TFLOPS, 4), issued_frac=round(6 * tf / PEAK_BF16_MFMA_TFLOPS, 4))
        elif model.mlp.endswith("f16x2"):
            r.update(frac=round(tf / PEAK_BF16_MFMA_TFLOPS, 4), issued_frac=round(3 * tf / PEAK_BF16_MFMA_TFLOPS, 4),
                     range_status=model.range_status())
        else:
            r.update(frac=round(tf / PEAK_F32_MFMA_TFLOPS, 4), peak=PEAK_F32_MFMA_TFLOPS)
        return o, r
    m = WideModel(sd_c, sd_f, device=device, mlp="f16x2")
    o, r = run(m, S.FLOP_PER_POINT)
    a, b = o["rgb_map"], ref["rgb_map"]
    d = (a - b).abs().max(-1)[0]
    mse = float(((a - b) ** 2).mean())
    r["vs_main_line_kernel_same_view"] = {"psnr_db": round(-10.0 * np.log10(max(mse, 1e-30)), 2), "rays": int(d.numel()),
                                          "rays_rgb_above_1e-4": int((d > 1e-4).sum())}
    out["ycbv_8x256"] = r
    m.close()
    for mlp, key in (("bf16x3", "ycbv_8x256_bf16x3"), ("fp32", "ycbv_8x256_strict_fp32")):
        m = WideModel(sd_c, sd_f, device=device, mlp=mlp)
        _, out[key] = run(m, S.FLOP_PER_POINT, launches=1)
        m.close()
    rng = np.random.RandomState(3)
    wide = {}
    Wd = 512
    for name, o_, i_ in [("pts_linears.0", Wd, 63)] + [("pts_linears.%d" % k, Wd, Wd + (63 if k == 5 else 0)) for k in range(1, 8)] + \
            [("feature_linear", Wd, Wd), ("alpha_linear", 1, Wd), ("views_linears.0", Wd // 2, Wd + 27), ("rgb_linear", 3, Wd // 2)]:
        bnd = 1.0 / np.sqrt(i_)
        wide[name + ".weight"] = (rng.uniform(-bnd, bnd, (o_, i_)) * (1.13 if name.startswith("pts") else 25.0 if name == "alpha_linear" else 1.0)).astype(np.float32)
        wide[name + ".bias"] = rng.uniform(-bnd, bnd, (o_,)).astype(np.float32)
    fpp = 2 * sum(v.size for k, v in wide.items() if k.endswith(".weight"))
    m = WideModel(wide, wide, device=device, mlp="f16x2")
    ps = PowerSampler(device)
    ps.start()
    _, r = run(m, fpp)
    r["power_and_clock"] = ps.stop()
    out["wide_8x512"] = r
    # forward + input gradient (nsrw_render_rays_vjp): + the fine pass's transposed GEMMs
    ro, rd = m.get_rays(H, W, S.YCBV_K, pose)
    cot = torch.ones(H * W, 3, device=ro.device)
    m.render_rays_vjp(ro.reshape(-1, 3), rd.reshape(-1, 3), S.YCBV_NEAR, S.YCBV_FAR, cot)
    m.render_rays_vjp(ro.reshape(-1, 3), rd.reshape(-1, 3), S.YCBV_NEAR, S.YCBV_FAR, cot)
    t, ch = m.last_kernel_ms()
    tf = (H * W * (256 + 192) * fpp) / t / 1e9
    out["wide_8x512_with_input_gradient"] = {"ms_per_view": round(t, 2), "chunks_of_rays": ch, "achieved": round(tf, 1), "unit": "TFLOP/s",
                                             "frac": round(tf / PEAK_BF16_MFMA_TFLOPS, 4), "issued_frac": round(3 * tf / PEAK_BF16_MFMA_TFLOPS, 4),
                                             "range_status": m.range_status()}
    m.close()
    for mlp, key in (("bf16x3", "wide_8x512_bf16x3"), ("fp32", "wide_8x512_strict_fp32")):
        m = WideModel(wide, wide, device=device, mlp=mlp)
        _, out[key] = run(m, fpp, launches=1)
        m.close()
    return out


def range_stress_workload(sd_c, sd_f, c2w, device, target=0.10):
    """f16x2 range safety net under load: a network ON the edge of the fp16 range -- a hidden bias of the
    coarse network's layer 0 just below the kernels' ceiling, calibrated (bisection on a 100x100 view) so that about `target`
    of the rays of this 400x400 view overflow -- rendered by the default handle (reported items re-rendered by the bf16x3
    kernel within the same launch call, r05) and by a handle whose fallback is the fp32-MFMA kernel (r04's route), next to
    the same view with the plain weights.  last_kernel_ms covers both launches of a call."""
    edge_c = {k: np.array(v, copy=True) for k, v in sd_c.items()}
    pose = torch.as_tensor(c2w[:3, :4])
    K100 = S.scaled_K(4.0)

    def frac(off, fallback, side, K):
        edge_c["pts_linears.0.bias"][7] = np.float32(65488.0 - off)
        m = NsrModel(edge_c, sd_f, device=device, range_fallback=fallback)
        m.render_views(pose, side, side, K, S.YCBV_NEAR, S.YCBV_FAR)
        t = []
        for _ in range(3):
            m.render_views(pose, side, side, K, S.YCBV_NEAR, S.YCBV_FAR)
            t.append(m.last_kernel_ms())
        st = m.range_status()
        m.close()
        return st["rays"] / (4.0 * side * side), float(np.mean(t)), st
    lo, hi = 0.0, 4.0                      # offset below the ceiling: larger = fewer rays overflow
    for _ in range(9):
        mid = 0.5 * (lo + hi)
        f, _, _ = frac(mid, "bf16x3", 100, K100)
        lo, hi = (mid, hi) if f > target else (lo, mid)
    off = 0.5 * (lo + hi)
    fb3, ms_b3, st_b3 = frac(off, "bf16x3", H, S.YCBV_K)
    f32, ms_32, st_32 = frac(off, "fp32", H, S.YCBV_K)
    plain = NsrModel(sd_c, sd_f, device=device)
    plain.render_views(pose, H, W, S.YCBV_K, S.YCBV_NEAR, S.YCBV_FAR)
    t = []
    for _ in range(3):
        plain.render_views(pose, H, W, S.YCBV_K, S.YCBV_NEAR, S.YCBV_FAR)
        t.append(plain.last_kernel_ms())
    plain.close()
    return {"workload": "400x400 view of a network whose coarse layer-0 bias[7] = 65488 - %.3f: %.1f %% of the rays leave the fp16 "
                        "range and are re-rendered inside the same launch call (kernel ms, HIP events over both launches)" % (off, 100 * fb3),
            "rays_rerendered_frac": round(fb3, 4), "ms_per_view_bf16x3_fallback": round(ms_b3, 3),
            "ms_per_view_fp32_fallback_r04": round(ms_32, 3), "ms_per_view_in_range_network": round(float(np.mean(t)), 3),
            "dropped_items": st_b3["dropped_items"] + st_32["dropped_items"]}


def config1_workload(sd_c, c2w, device, cpu_setting):
    """BASELINE configs[0]: 64x64 view, 64 coarse samples only.  GPU: 50 launches, HIP-event mean.
    CPU: the oracle on the same view at the reference's chunk (512, CF:25) and at 4096."""
    side, n_s = 64, 64
    K = S.scaled_K(400.0 / side)
    m1 = NsrModel(sd_c, None, device=device, n_importance=0)
    pose = torch.as_tensor(c2w[:3, :4], device=m1.device)
    for _ in range(3):
        got = m1.render_views(pose, side, side, K, S.YCBV_NEAR, S.YCBV_FAR)
    ms = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        got = m1.render_views(pose, side, side, K, S.YCBV_NEAR, S.YCBV_FAR)
        ms.append(m1.last_kernel_ms())
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 50
    k_ms = float(np.mean(ms))
    rays = side * side
    flop = rays * n_s * S.FLOP_PER_POINT
    out = {"workload": "YCB-V object-2 camera, 64x64 view, N_samples=64 coarse only (BASELINE configs[0])",
           "gpu": {"value": round(rays * n_s / wall / 1e6, 3), "unit": "Mray-samples/s", "ms_per_view_wall": round(wall * 1e3, 4),
                   "kernel_ms": round(k_ms, 4), "kernel_TFLOPs": round(flop / (k_ms * 1e-3) / 1e12, 2),
                   "note": "4096 rays = 8 rays per workgroup of the 512-workgroup grid: launch- and tail-bound, not "
                           "MFMA-bound; the per-view wall time includes the host's call overhead"}}
    if cpu_setting is not None:
        O = _oracle()
        backend, threads = cpu_setting
        O.set_backend(backend)
        if threads:
            torch.set_num_threads(threads)
        cpu = {}
        ref = None
        for chunk in (512, 4096):
            O.render(sd_c, None, side, side, K, c2w=c2w[:3, :4], near=S.YCBV_NEAR, far=S.YCBV_FAR, chunk=chunk, n_importance=0)
            t0 = time.perf_counter()
            ref = O.render(sd_c, None, side, side, K, c2w=c2w[:3, :4], near=S.YCBV_NEAR, far=S.YCBV_FAR, chunk=chunk,
                           n_importance=0)
            dt = time.perf_counter() - t0
            cpu["chunk_%d" % chunk] = {"value": round(rays * n_s / dt / 1e6, 4), "unit": "Mray-samples/s", "seconds": round(dt, 3)}
        O.set_backend("numpy")
        cpu.update(kind="port", cores=threads if threads else min(os.cpu_count() or 1, 64), backend=backend,
                   cpu_model=cpu_model())
        out["cpu_baseline"] = cpu
        # end to end against the oracle, per forward kernel: a coarse-only render has one discontinuity, the sigma_last
        # cliff (RN:358-359); every ray beyond 1e-4 must be one (oracle/census.py)
        import census as C
        ro, rd = O.get_rays(side, side, K, c2w[:3, :4])
        ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
        ref1 = O.render_rays(sd_c, None, ro, rd, O.normalize_dirs(rd), S.YCBV_NEAR, S.YCBV_FAR, n_importance=0, extras=True)
        out["parity"] = {}
        for mlp in MLP_MODES:
            mk = NsrModel(sd_c, None, device=device, n_importance=0, mlp=mlp)
            g1 = mk.render_views(pose, side, side, K, S.YCBV_NEAR, S.YCBV_FAR, debug=True)
            c = C.census((sd_c, None), ro, rd, S.YCBV_NEAR, S.YCBV_FAR,
                         {k: g1[k].cpu().numpy() for k in ("rgb_map", "acc_map", "disp_map", "raw0")}, ref1, coarse_only=True)
            out["parity"][mlp] = {k: c[k] for k in ("rays", "tol", "rays_above_tol", "cliff_rays", "unattributed", "max_abs_rgb",
                                                    "max_abs_acc", "psnr_vs_oracle_db", "psnr_delta_db",
                                                    "psnr_delta_db_excluding_attributed")}
            out["parity"][mlp]["passes"] = bool(C.passes(c))
            mk.close()
    m1.close()
    return out


def handoff_workload(model, with_cpu):
    """f-3: to8b (RH:14) + the detector's bbox extraction (NM:786-797) for 100 rendered-size images on the device
    (HBM-bound byte kernels), next to the reference's route on the host -- to8b + PNG encode + PNG decode + the oracle's
    restatement of the cv2 calls -- timed on 5 images (the oracle is the checker: boxes / masks must agree)."""
    from neural_sim_nerf_amd import png
    K_, H_, W_ = 100, 400, 400
    rng = np.random.RandomState(0)
    yy, xx = np.mgrid[:H_, :W_]
    rgb = np.zeros((K_, H_, W_, 3), np.float32)
    for i in range(K_):                                  # an object-like blob on black, plus a few specks
        cy, cx, r = rng.randint(120, 280), rng.randint(120, 280), rng.randint(40, 110)
        sel = (yy - cy) ** 2 + ((xx - cx) * rng.uniform(0.6, 1.4)) ** 2 <= r * r
        rgb[i][sel] = rng.uniform(0.05, 1.0, (sel.sum(), 3))
        for _ in range(5):
            rgb[i][rng.randint(0, H_), rng.randint(0, W_)] = 0.5
    x = torch.as_tensor(rgb, device=model.device)

    def timed(fn, reps=20):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    img8 = model.to8b(x)
    ms_to8b = timed(lambda: model.to8b(x))
    ms_bbox = timed(lambda: model.find_bbox(img8, with_mask=True))
    bbox, count, mask = model.find_bbox(img8, with_mask=True)
    n = K_ * H_ * W_
    out = {"workload": "to8b + find_bbox of 100 images 400x400x3 (f-3 hand-off)",
           "to8b": {"ms": round(ms_to8b, 4), "algorithmic_bytes": n * 3 * 5, "GBps": round(n * 3 * 5 / ms_to8b / 1e6, 1),
                    "frac_of_8TBps": round(n * 3 * 5 / ms_to8b / 1e6 / 8000, 4)},
           "find_bbox": {"ms": round(ms_bbox, 4), "algorithmic_bytes": n * 4},
           "views_per_s_gpu": round(K_ / ((ms_to8b + ms_bbox) * 1e-3), 1)}
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import handoff_oracle as HO
        sample = 5
        d = tempfile.mkdtemp(prefix="nsr_handoff_")
        t0 = time.perf_counter()
        for i in range(sample):
            png.imwrite(os.path.join(d, "%03d.png" % i), HO.to8b(rgb[i]))
        ok = True
        for i in range(sample):
            want, rows, mk = HO.get_annotation(png.imread(os.path.join(d, "%03d.png" % i)))
            ok &= (list(bbox[i].cpu().numpy()) == [int(v) for v in want] and int(count[i]) == rows
                   and np.array_equal(mask[i].cpu().numpy(), mk))
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(sample / dt, 2), "unit": "views/s", "cores": 1, "kind": "port",
                               "cpu_model": cpu_model(), "sample": "%d images: to8b + PNG encode + PNG decode + "
                               "oracle/handoff_oracle.get_annotation, %.2f s" % (sample, dt)}
        out["parity_on_sample"] = bool(ok)
    return out


def vjp_roofline(model, c2w, pmc_file=None):
    """The input-gradient kernel (nsr::k_render_vjp16 by default) on one 400x400 image = render_path_grad's per-pose
    launch: mean HIP-event time of 2 launches after one warm-up."""
    o, d = model.get_rays(H, W, S.YCBV_K, c2w)
    cot = torch.randn(H * W, 3, device=model.device, generator=torch.Generator(device=model.device).manual_seed(0))
    ms = []
    for _ in range(3):
        model.render_rays_vjp(o.reshape(-1, 3), d.reshape(-1, 3), S.YCBV_NEAR, S.YCBV_FAR, cot)
        ms.append(model.last_kernel_ms())
    k_ms = float(np.mean(ms[1:]))
    ach = H * W * FLOP_PER_RAY_VJP / (k_ms * 1e-3) / 1e12
    flop_note = ("per ray: 256 forward evaluations + 192 evaluations of the transposed fine network (input-side "
                 "VJP only: weights are constants) x 1 186 816 FLOP = 531.7 MFLOP")
    if model.mlp in ("bf16x3", "f16x2"):
        fwd, bwd = ((B3_ISSUED_FLOP_PER_POINT, B3_ISSUED_FLOP_PER_POINT_BWD) if model.mlp == "bf16x3" else
                    (H2_ISSUED_FLOP_PER_POINT, H2_ISSUED_FLOP_PER_POINT_BWD))
        issued = H * W * (EVALS_PER_RAY * fwd + 192 * bwd) / (k_ms * 1e-3) / 1e12
        traffic, key = None, "vjp_" + model.mlp
        if pmc_file and os.path.exists(pmc_file):
            prof = json.load(open(pmc_file))
            if prof.get("kernel_source_sha256") == _lib.kernel_source_hash() and key in prof:
                traffic = prof[key]["derived"].get("hbm_traffic_bytes_per_launch")   # same hash rule as roofline.traffic
        return {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_BF16_MFMA_TFLOPS, 4), "issued": round(issued, 1),
                "issued_frac": round(issued / PEAK_BF16_MFMA_TFLOPS, 4),
                "attainable_frac": round(1.0 / (6 if model.mlp == "bf16x3" else 3), 4),
                "frac_of_attainable": round(ach / PEAK_BF16_MFMA_TFLOPS * (6 if model.mlp == "bf16x3" else 3), 4),
                "achieved_over_fp32_mfma_peak": round(ach / PEAK_F32_MFMA_TFLOPS, 3), "traffic": traffic,
                "kernel": "nsr::k_render_vjp_b3" if model.mlp == "bf16x3" else "nsr::k_render_vjp_h2",
                "kernel_ms": round(k_ms, 3), "flop_per_launch": H * W * FLOP_PER_RAY_VJP, "flop_note": flop_note}
    traffic = None
    if pmc_file and os.path.exists(pmc_file) and model.variant != 32:
        prof = json.load(open(pmc_file))
        if prof.get("kernel_source_sha256") == _lib.kernel_source_hash() and "vjp" in prof:
            traffic = prof["vjp"]["derived"].get("hbm_traffic_bytes_per_launch")     # same hash rule as roofline.traffic
    return {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
            "kernel": "nsr::k_render_vjp" if model.variant == 32 else
                      ("nsr::k_render_vjp16p" if model.schedule == "phases" else "nsr::k_render_vjp16"),
            "kernel_ms": round(k_ms, 3), "flop_per_launch": H * W * FLOP_PER_RAY_VJP, "flop_note": flop_note}


def pmc_traffic(pmc_file, schedule="phases"):
    """HBM-side bytes per launch from a rocprofv3 --pmc profile (tools/collect_profiles.sh + summarize_pmc.py), used
    ONLY when the profile was collected from exactly the kernel sources in this tree; otherwise null."""
    if not os.path.exists(pmc_file):
        return None, "no PMC profile at %s" % os.path.relpath(pmc_file, ROOT)
    prof = json.load(open(pmc_file))
    here = _lib.kernel_source_hash()
    if prof.get("kernel_source_sha256") != here:
        return None, ("PMC profile %s was collected from other kernel sources (%s..., this tree %s...): not reported"
                      % (os.path.relpath(pmc_file, ROOT), str(prof.get("kernel_source_sha256"))[:12], here[:12]))
    blob = hashlib.sha1(b"blob %d\0" % os.path.getsize(pmc_file) + open(pmc_file, "rb").read()).hexdigest()
    key = {"phases": "x16_phases_schedule", "bf16x3": "bf16x3", "f16x2": "f16x2"}.get(schedule, "x16_queue_schedule")
    if key not in prof:
        return None, "PMC profile %s has no passes for the %s schedule" % (os.path.relpath(pmc_file, ROOT), schedule)
    t = prof[key]["derived"]["hbm_traffic_bytes_per_launch"]
    return t, ("offline measurement: bytes/launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes of "
               "this kernel and schedule (%s [%s], git blob %s, kernel sources sha256 %s... = this tree); FETCH_SIZE counts "
               "L2 misses that Infinity Cache serves: re-streaming of the weight images (4.6 MiB per network pair in the fp32 "
               "and f16x2 layouts, 6.9 MiB in the bf16x3 layout, vs 4 MiB L2 per XCD), not HBM reads; algorithmic HBM bytes "
               "are 7.0e6 per launch (DESIGN.md 4)"
               % (os.path.relpath(pmc_file, ROOT), key, blob[:12], here[:12]))


_KNAME = {"f16x2": "k_render_h2", "bf16x3": "k_render_b3", "fp32": "k_render16p"}


def live_traffic(mlp):
    """HBM-side bytes of ONE launch of the forward kernel, measured NOW: two extra launches of the same view in child
    processes under `rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE` and `--pmc WRITE_SIZE` (separate passes, counters + kernel
    trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes; FETCH_SIZE doubled per that guide's gfx950 correction;
    GRBM has its own counter slots), when rocprofv3 is on the box.  The first pass also gives the shader clock the kernel
    sustained: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the kernel's duration in that pass.
    Returns (bytes or None, note, dict(clock_GHz, kernel_ms_under_pmc) or None)."""
    import csv
    import glob
    import shutil
    import subprocess
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found on this box", None
    if any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_")) or k == "HSA_TOOLS_LIB" for k in os.environ):
        return None, "bench.py is itself running under a profiler: the live PMC passes are skipped (no nested rocprofv3)", None
    tot, clock = {}, None
    tmp = tempfile.mkdtemp(prefix="nsr_pmc_")
    try:
        for ctrs in (("FETCH_SIZE", "GRBM_GUI_ACTIVE"), ("WRITE_SIZE",)):
            d = os.path.join(tmp, ctrs[0])
            r = subprocess.run([exe, "--pmc"] + list(ctrs) + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                                os.path.join(ROOT, "tools", "one_view.py"), "0"], cwd=tmp, capture_output=True, text=True,
                               timeout=120, env=dict(os.environ, NSR_MLP=mlp, TMPDIR=tmp))
            files = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s failed (rc %d): %s" % (ctrs[0], r.returncode, (r.stderr or r.stdout)[-200:]), None
            first = None
            for row in csv.DictReader(open(files[0])):
                if _KNAME[mlp] in row["Kernel_Name"] and row["Counter_Name"] in ctrs:
                    first = first or row["Dispatch_Id"]
                    if row["Dispatch_Id"] == first:
                        tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            if ctrs[0] not in tot:
                return None, "rocprofv3 --pmc %s: no %s dispatch in the counter file" % (ctrs[0], _KNAME[mlp]), None
            if "GRBM_GUI_ACTIVE" in ctrs and "GRBM_GUI_ACTIVE" in tot:
                for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
                    for row in csv.DictReader(open(f)):
                        if _KNAME[mlp] in row["Kernel_Name"] and clock is None:
                            ns = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                            clock = {"clock_GHz": round(tot["GRBM_GUI_ACTIVE"] / 8.0 / ns, 4), "kernel_ms_under_pmc": round(ns / 1e6, 3)}
    except (OSError, subprocess.SubprocessError, KeyError, ValueError) as e:
        return None, "live PMC pass failed: %r" % (e,), None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0, (
        "measured in THIS run: one extra launch of the same view per counter under rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE in "
        "separate passes, summed over the XCDs); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; L2 misses served by Infinity Cache "
        "(weight re-streaming), not HBM reads -- algorithmic HBM bytes are 7.0e6 per launch (DESIGN.md 4)"), clock


class PowerSampler:
    """Socket power and shader clock of one GPU while the timed steps run, from the amdgpu hwmon files of its PCI device
    (power1_average / power1_input in microwatt, freq1_input in Hz), sampled by a thread every 20 ms; falls back to ONE
    `amd-smi metric --json` / `rocm-smi --json` reading taken by a child process during the steps.  Reported as evidence
    for "these kernels run at the power limit" (DESIGN.md 4): a number the driver's own bench line carries."""

    def __init__(self, device_index):
        import glob
        import threading
        self.samples, self.note, self.stop_flag, self.thread, self.proc = [], None, False, None, None
        self.files = None
        try:
            bus = torch.cuda.get_device_properties(device_index)
            want = "%04x:%02x:%02x" % (getattr(bus, "pci_domain_id", 0), getattr(bus, "pci_bus_id", -1), getattr(bus, "pci_device_id", 0))
        except Exception:                              # noqa: BLE001 -- evidence only, never fatal
            want = None
        cands = []
        for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            pw = next((os.path.join(hw, f) for f in ("power1_average", "power1_input") if os.path.exists(os.path.join(hw, f))), None)
            fq = os.path.join(hw, "freq1_input") if os.path.exists(os.path.join(hw, "freq1_input")) else None
            if pw or fq:
                real = os.path.realpath(os.path.join(hw, "..", ".."))
                cands.append((want is not None and want in real, pw, fq, real))
        cands.sort(key=lambda c: not c[0])
        if cands and (cands[0][0] or len(cands) == 1 or want is None):
            self.files = cands[0][1:3]
            self.note = "amdgpu hwmon of %s (%s, %s), 20 ms samples over the timed steps" % (
                os.path.basename(cands[0][3]), os.path.basename(self.files[0] or "-"), os.path.basename(self.files[1] or "-"))
            self.thread = threading.Thread(target=self._run, daemon=True)
        self.device_index = device_index

    def _read(self, f):
        try:
            return float(open(f).read().strip())
        except (OSError, ValueError):
            return None

    def _run(self):
        while not self.stop_flag:
            self.samples.append((self._read(self.files[0]) if self.files[0] else None, self._read(self.files[1]) if self.files[1] else None))
            time.sleep(0.02)

    def start(self):
        import shutil
        if self.thread is not None:
            self.thread.start()
            return
        exe = shutil.which("amd-smi")
        if exe:          # one reading, taken while the steps run (the child starts now and samples ~0.5-1 s later)
            self.proc = subprocess.Popen([exe, "metric", "-g", str(self.device_index), "-p", "-c", "--json"], stdout=subprocess.PIPE,
                                         stderr=subprocess.DEVNULL, text=True)
            self.note = "one `amd-smi metric -p -c --json` reading taken while the timed steps ran"

    def stop(self):
        self.stop_flag = True
        if self.thread is not None:
            self.thread.join(timeout=1.0)
            pw = [a for a, _ in self.samples if a]
            fq = [b for _, b in self.samples if b]
            if not pw and not fq:
                return {"note": "hwmon files unreadable"}
            out = {"samples": len(self.samples), "source": self.note}
            if pw:
                out.update(socket_power_W_mean=round(float(np.mean(pw)) / 1e6, 1), socket_power_W_max=round(max(pw) / 1e6, 1))
            if fq:
                out.update(sclk_MHz_mean=round(float(np.mean(fq)) / 1e6, 1), sclk_MHz_min=round(min(fq) / 1e6, 1))
            return out
        if self.proc is not None:
            try:
                txt = self.proc.communicate(timeout=20)[0]
                data = json.loads(txt)
                return {"source": self.note, "amd_smi": data[0] if isinstance(data, list) and data else data}
            except Exception as e:                     # noqa: BLE001
                return {"note": "amd-smi reading failed: %r" % (e,)}
        return {"note": "no hwmon power / clock files and no amd-smi on this box"}


def strict_fp32_line(sd_c, sd_f, device, c2w, launches=3):
    """The same view on the strict-fp32 kernel (k_render16p: fp32-input MFMAs, exact fp32 products), in the same run: what a
    reader who wants BASELINE configs[1]'s "fp32" taken literally should look at next to a split-arithmetic headline."""
    m = NsrModel(sd_c, sd_f, device=device, mlp="fp32")
    pose = torch.as_tensor(c2w[:3, :4], device=m.device)
    m.render_views(pose, H, W, S.YCBV_K, S.YCBV_NEAR, S.YCBV_FAR)
    torch.cuda.synchronize()
    ms = []
    t0 = time.perf_counter()
    for _ in range(launches):
        m.render_views(pose, H, W, S.YCBV_K, S.YCBV_NEAR, S.YCBV_FAR)
        ms.append(m.last_kernel_ms())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k_ms = float(np.mean(ms))
    r = {"value": round(launches * H * W * SAMPLES_PER_RAY / dt / 1e6, 3), "unit": "Mray-samples/s", "kernel": forward_kernel_name(m),
         "kernel_ms": round(k_ms, 3), "frac": round(H * W * FLOP_PER_RAY / (k_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
         "peak": PEAK_F32_MFMA_TFLOPS}
    m.close()
    return r


def self_launch(args):
    """`python bench.py --gpus N` outside a launcher: re-exec under torch.distributed.run, one rank per GPU."""
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and not (args.share_gpu and n_dev > 0):
        raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible on this node" % (args.gpus, n_dev))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=("view400", "sweep100", "models21"), default="view400")
    ap.add_argument("--views", type=int, default=100, help="sweep100: number of views in the sweep (config 3 uses 100)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--trained-side", type=int, default=160,
                    help="side of the view of extra_workloads.trained (the census on the TRAINED pair g26 against the oracle's render: 19 s of "
                         "host time at 160, 2.5 min at the full 400)")
    ap.add_argument("--cpu-sample-side", type=int, default=400,
                    help="side of the view the oracle renders for cpu_baseline and parity (default: the metric's own "
                         "400x400 view, ~2-3 min of CPU time; e.g. 128 for a quick run)")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline_vjp and the config-1 workload")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="nccl = RCCL over xGMI (the measured configuration); gloo stages collectives through the host")
    ap.add_argument("--share-gpu", action="store_true",
                    help="validation only: ranks share the visible GPUs round-robin (e.g. --gpus 2 --backend gloo on a "
                         "1-GPU box exercises the N>1 code path end to end; the number is not a scaling result)")
    ap.add_argument("--pmc-file", default=next((f for f in (os.path.join(ROOT, "profiles", r, "pmc_k_render.json") for r in ("r05", "r04", "r03"))
                                                 if os.path.exists(f)), os.path.join(ROOT, "profiles", "r05", "pmc_k_render.json")))
    ap.add_argument("--mlp", choices=MLP_MODES, default=None,
                    help="layer-GEMM arithmetic of the forward kernel (default: the engine's, engine.DEFAULT_MLP / $NSR_MLP)")
    args = ap.parse_args()

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if launched and args.gpus != world:
        raise SystemExit("bench.py --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if args.share_gpu and torch.cuda.device_count() > 0:
        local = local % torch.cuda.device_count()
    if local >= torch.cuda.device_count():
        raise SystemExit("bench.py: local rank %d has no HIP device (%d visible)" % (local, torch.cuda.device_count()))
    if args.share_gpu and args.backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit("bench.py --share-gpu: RCCL cannot put two ranks on one device; use --backend gloo")
    torch.cuda.set_device(local)
    dist = None
    if launched:                                  # under torch.distributed.run, also for a single rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("NCCL_DEBUG", "VERSION").upper() == "VERSION":
            os.environ["NCCL_DEBUG"] = "WARN"     # RCCL's version banner goes to stdout: keep stdout to the ONE JSON line
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    dev = torch.device("cuda", local)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")       # where collective buffers live

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def allmax(x):
        if dist is None:
            return float(x)
        t = torch.tensor([x], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def rank_stats(k_ms):
        """ranks_seen = all-reduce(sum) of ones; per-rank mean kernel time gathered to every rank."""
        if dist is None:
            return 1, [float(k_ms)]
        ones = torch.ones(1, device=cdev, dtype=torch.int32)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        mine = torch.tensor([k_ms], device=cdev, dtype=torch.float64)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        return int(ones.item()), [float(p.item()) for p in parts]

    sd_c = S.synth_weights(0)
    sd_f = S.synth_weights(1000, fine_of=sd_c)
    line = {"metric": METRIC, "unit": "Mray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic"}

    # ------------------------------------------------------------------------------------------------------------
    if args.workload == "view400":
        model = NsrModel(sd_c, sd_f, device=local, mlp=args.mlp)
        n_total = args.warmup + args.steps
        poses = S.sweep_poses(n_total * world, seed=0)[rank::world]      # view i -> rank i mod world
        poses_d = torch.as_tensor(poses[:, :3, :4], device=model.device)
        for i in range(args.warmup):
            model.render_views(poses_d[i], H, W, S.YCBV_K, S.YCBV_NEAR, S.YCBV_FAR)
        if dist is not None:          # untimed: bring up the RCCL channels the timed all_gather will use (same shape)
            dummy = torch.zeros((args.steps, H * W, 3), device=cdev)
            dist.all_gather([torch.empty_like(dummy) for _ in range(world)], dummy)
            del dummy
        sampler = PowerSampler(local) if rank == 0 else None      # (a reader thread / child process: nothing on the GPU)
        barrier()
        if sampler is not None:
            sampler.start()
        kernel_ms, images = [], []
        t0 = time.perf_counter()
        for i in range(args.warmup, n_total):
            out = model.render_views(poses_d[i], H, W, S.YCBV_K, S.YCBV_NEAR, S.YCBV_FAR)
            kernel_ms.append(model.last_kernel_ms())      # HIP events on the launch stream (syncs on the stop event)
            images.append(out["rgb_map"])
        if dist is not None:                                # outer-loop boundary: gather the rendered images
            mine = torch.stack(images, 0).to(cdev)
            gathered = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(gathered, mine)
        barrier()
        dt = allmax(time.perf_counter() - t0)
        power = sampler.stop() if sampler is not None else None
        k_ms = float(np.mean(kernel_ms))
        ranks_seen, per_rank = rank_stats(k_ms)
        if rank == 0:
            rays = args.steps * H * W * world
            roof = forward_roofline(model, k_ms)
            roof["power_and_clock_over_the_timed_steps"] = power
            if model.mlp == "bf16x3":
                traffic, traffic_note = pmc_traffic(args.pmc_file, "bf16x3")
                kernel_desc = "fused persistent kernel k_render_b3 (one workgroup per CU, 32 points per wave, layer GEMMs on bf16 MFMAs with three-way split fp32 operands)"
                line["dtype"] = "bf16x3"
            elif model.mlp == "f16x2":
                traffic, traffic_note = pmc_traffic(args.pmc_file, "f16x2")
                kernel_desc = "fused persistent kernel k_render_h2 (one workgroup per CU, 32 points per wave, layer GEMMs on fp16 MFMAs with two-way split fp32 operands and power-of-two range management)"
                line["dtype"] = "f16x2"
            else:
                traffic, traffic_note = pmc_traffic(args.pmc_file, model.schedule)
                kernel_desc = "fp32, fused persistent kernel (x16: 2 workgroups per CU, %s schedule)" % model.schedule
            traffic_source = ("offline rocprofv3 --pmc passes of this kernel, used only if their recorded kernel-source hash "
                              "equals this tree's (else null)")
            if world == 1 and not args.no_extras:            # measured in this run when rocprofv3 is on the box
                live, live_note, clock = live_traffic(model.mlp)
                if live is not None:
                    traffic, traffic_note, traffic_source = live, live_note, "live"
                else:
                    traffic_note = "%s | live pass: %s" % (traffic_note, live_note)
                if clock is not None:          # GRBM_GUI_ACTIVE / 8 XCDs / kernel duration, both from the live FETCH_SIZE pass
                    roof["clock_GHz"] = clock["clock_GHz"]
                    roof["clock_note"] = ("shader clock the kernel sustained in the live PMC pass of this run (kernel %.3f ms under "
                                          "the profiler; profiled passes clock ~3 %% below un-profiled ones, MI355X_MICROARCH.md "
                                          "DVFS note); peak 2.4 GHz" % clock["kernel_ms_under_pmc"])
            roof.update({"traffic": traffic, "traffic_source": traffic_source, "traffic_note": traffic_note})
            if world == 1 and model.mlp != "fp32":
                roof["strict_fp32"] = strict_fp32_line(sd_c, sd_f, local, poses[args.warmup])
                # (r06: also at the top level of the line -- the figure a reader who takes BASELINE configs[1]'s "fp32" literally wants)
                line["strict_fp32"] = roof["strict_fp32"]
            line["power_and_clock_over_the_timed_steps"] = power
            line["range_status"] = model.range_status()      # f16x2 range safety net: all zero = nothing left the fp16 range
            line.update({
                "value": round(rays * SAMPLES_PER_RAY / dt / 1e6, 3), "ms_per_step": round(dt / args.steps * 1e3, 3),
                "config": {"workload": "YCB-V object-2 camera, 400x400 view per step per GPU, 64 coarse + 128 fine "
                                       "samples/ray, 8x256 NeRF MLP pair (seeded synthetic weights), %s, rays generated "
                                       "in-kernel" % kernel_desc,
                           "rays_per_step_per_gpu": H * W, "mlp_evals_per_ray": EVALS_PER_RAY,
                           "parallelism": "views sharded over %d rank(s), image all-gather (%s) at the end%s"
                                          % (world, "RCCL" if args.backend == "nccl" else "gloo via the host",
                                             "; ranks SHARE a GPU (validation run)" if args.share_gpu else "")},
                "rays_per_s": round(rays / dt, 1), "mlp_evals_per_s": round(rays * EVALS_PER_RAY / dt, 1),
                "ranks_seen": ranks_seen,
                "kernel_ms_per_rank": {"min": round(min(per_rank), 3), "max": round(max(per_rank), 3),
                                       "mean": round(float(np.mean(per_rank)), 3)},
                "roofline": roof,
            })
            cpu_setting, cpu_sample = None, None
            if world == 1 and not args.no_cpu_baseline:
                cpu, par, cpu_setting, cpu_sample = cpu_baseline_and_parity(model, sd_c, sd_f, poses[args.warmup],
                                                                            side=args.cpu_sample_side)
                line["cpu_baseline"] = cpu
                line["parity"] = par
                line["parity_summary"] = {"rays": par["census"]["rays"], "rays_above_tol": par["census"]["rays_above_tol"],
                                          "unattributed": par["census"]["unattributed"], "psnr_delta_db": par["psnr_delta_db"],
                                          "inds_exact": par["inds_exact_match_rate"], "passes": par["passes"]}
            if world == 1 and not args.no_extras:
                line["roofline_vjp"] = vjp_roofline(model, poses[args.warmup], args.pmc_file)
                line["extra_workloads"] = {"config1": config1_workload(sd_c, poses[args.warmup], local, cpu_setting),
                                           "handoff": handoff_workload(model, not args.no_cpu_baseline),
                                           "api_overhead": api_overhead_workload(sd_c, sd_f, local),
                                           "render_options": render_options_workload(sd_c, sd_f, poses[0], local),
                                           "range_stress": range_stress_workload(sd_c, sd_f, poses[args.warmup], local)}
                ref = model.render_views(poses_d[args.warmup], H, W, S.YCBV_K, S.YCBV_NEAR, S.YCBV_FAR)
                try:                 # (a side workload must never cost the main line)
                    line["extra_workloads"]["layered"] = layered_workload(sd_c, sd_f, poses[args.warmup], local, ref)
                except Exception as e:                     # noqa: BLE001
                    line["extra_workloads"]["layered"] = {"error": repr(e)}
                try:
                    line["extra_workloads"]["trained"] = trained_workload(local, cpu_setting, side=args.trained_side)
                except Exception as e:                     # noqa: BLE001
                    line["extra_workloads"]["trained"] = {"error": repr(e)}
                for mlp in MLP_MODES:
                    if mlp != model.mlp:
                        line["extra_workloads"][mlp] = alt_mlp_workload(mlp, sd_c, sd_f, local, poses[args.warmup], ref,
                                                                        sample=cpu_sample, pmc_file=args.pmc_file)
        model.close()

    # ------------------------------------------------------------------------------------------------------------
    elif args.workload == "sweep100":
        # BASELINE configs[2]: 100-view pose sweep, view i -> rank i mod N (13/12 views per rank at N=8), each rank
        # renders its views in ONE launch, converts to uint8 on the device (to8b, RH:14), writes its own PNGs; the
        # images are all-gathered both as uint8 (what the detector consumes, 0.48 MB/view) and as fp32 (what
        # render_path returns, 1.92 MB/view).  A step = one whole sweep.
        from neural_sim_nerf_amd import png
        n_views = args.views
        model = NsrModel(sd_c, sd_f, device=local, mlp=args.mlp)
        poses = S.sweep_poses(n_views, seed=0)
        mine = D.shard_indices(n_views, world, rank)
        poses_d = torch.as_tensor(poses[mine][:, :3, :4], device=model.device)
        tmp = tempfile.mkdtemp(prefix="nsr_sweep_")
        for _ in range(max(1, args.warmup)):
            out = model.render_views(poses_d[:1], H, W, S.YCBV_K, S.YCBV_NEAR, S.YCBV_FAR)
            D.gather_views(model.to8b(out["rgb_map"].reshape(1, H, W, 3)), world)     # RCCL channel bring-up
        phases = {"render": 0.0, "gather_u8": 0.0, "gather_f32": 0.0, "png": 0.0}
        kernel_ms = []
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            t = time.perf_counter()
            out = model.render_views(poses_d, H, W, S.YCBV_K, S.YCBV_NEAR, S.YCBV_FAR)
            rgb = out["rgb_map"].reshape(-1, H, W, 3)
            u8 = model.to8b(rgb)
            kernel_ms.append(model.last_kernel_ms())
            torch.cuda.synchronize()
            phases["render"] += time.perf_counter() - t
            t = time.perf_counter()
            all_u8 = D.gather_views(u8, n_views)
            torch.cuda.synchronize()
            phases["gather_u8"] += time.perf_counter() - t
            t = time.perf_counter()
            all_f32 = D.gather_views(rgb, n_views)
            torch.cuda.synchronize()
            phases["gather_f32"] += time.perf_counter() - t
            t = time.perf_counter()
            host = u8.cpu().numpy()
            png.imwrite_many([os.path.join(tmp, "%03d.png" % i) for i in mine],     # every rank writes its own views,
                             [host[k] for k in range(len(mine))])                   # named by pose index
            phases["png"] += time.perf_counter() - t
        barrier()
        dt = allmax(time.perf_counter() - t0)
        phases = {k: round(allmax(v) / args.steps, 4) for k, v in phases.items()}
        ranks_seen, per_rank = rank_stats(float(np.mean(kernel_ms)))
        assert tuple(all_u8.shape) == (n_views, H, W, 3) and tuple(all_f32.shape) == (n_views, H, W, 3)
        if rank == 0:
            rays = args.steps * n_views * H * W
            k_max = (n_views + world - 1) // world
            line.update({
                "value": round(rays * SAMPLES_PER_RAY / dt / 1e6, 3), "ms_per_step": round(dt / args.steps * 1e3, 3),
                "scaling": "strong",
                "config": {"workload": "%d-view pose sweep (BASELINE configs[2]), 400x400, 64+128 samples/ray, view i -> "
                                       "rank i mod N, one launch per rank, device to8b, uint8 + fp32 image all-gather, "
                                       "PNG write per rank" % n_views, "views": n_views, "views_on_busiest_rank": k_max,
                           "parallelism": "views sharded over %d GPU(s)" % world},
                "views_per_s": round(args.steps * n_views / dt, 3),
                "ideal_speedup_over_1_gpu": round(n_views / k_max, 3),
                "seconds_per_sweep_by_phase_max_over_ranks": phases,
                "ranks_seen": ranks_seen,
                "kernel_ms_per_rank": {"min": round(min(per_rank), 3), "max": round(max(per_rank), 3),
                                       "mean": round(float(np.mean(per_rank)), 3)},
            })
        model.close()

    # ------------------------------------------------------------------------------------------------------------
    else:
        # BASELINE configs[4]: all 21 YCB-V NeRFs (synthetic weights, seeds 0..20), model m -> rank m mod N, one native
        # handle + one HIP stream per model, one 400x400 view per model per step; no collective on the data path.
        n_models = 21
        mine = D.shard_models(n_models, world, rank)
        models, streams = [], []
        for m in mine:
            c = S.synth_weights(m)
            models.append(NsrModel(c, S.synth_weights(1000 + m, fine_of=c), device=local, mlp=args.mlp))
            streams.append(torch.cuda.Stream(device=dev))
        poses = torch.as_tensor(S.sweep_poses(n_models, seed=3)[:, :3, :4], device=dev)

        def sweep():
            outs = []
            for i, (m, st) in enumerate(zip(models, streams)):
                with torch.cuda.stream(st):
                    outs.append(m.render_views(poses[mine[i]], H, W, S.YCBV_K, S.YCBV_NEAR, S.YCBV_FAR)["rgb_map"])
            return outs
        for _ in range(max(1, args.warmup)):
            sweep()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            outs = sweep()
        barrier()
        dt = allmax(time.perf_counter() - t0)
        k_ms = float(np.mean([m.last_kernel_ms() for m in models])) if models else 0.0
        ranks_seen, per_rank = rank_stats(k_ms)
        if rank == 0:
            rays = args.steps * n_models * H * W
            k_max = (n_models + world - 1) // world
            line.update({
                "value": round(rays * SAMPLES_PER_RAY / dt / 1e6, 3), "ms_per_step": round(dt / args.steps * 1e3, 3),
                "scaling": "strong",
                "config": {"workload": "21 NeRF models (BASELINE configs[4]), one 400x400 64+128 view each per step, model m "
                                       "-> rank m mod N, one handle + HIP stream per model", "models": n_models,
                           "models_on_busiest_rank": k_max, "parallelism": "models sharded over %d GPU(s)" % world},
                "views_per_s": round(args.steps * n_models / dt, 3),
                "ideal_speedup_over_1_gpu": round(n_models / k_max, 3),
                "ranks_seen": ranks_seen,
                "kernel_ms_per_rank": {"min": round(min(per_rank), 3), "max": round(max(per_rank), 3),
                                       "mean": round(float(np.mean(per_rank)), 3)},
            })
        for m in models:
            m.close()

    if rank == 0:
        eff_mlp = args.mlp or os.environ.get("NSR_MLP") or DEFAULT_MLP          # what NsrModel(mlp=args.mlp) resolved to
        if args.workload != "view400" and eff_mlp in MLP_DTYPE:
            line["dtype"] = eff_mlp
            line["config"]["mlp"] = MLP_DTYPE[eff_mlp]
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
