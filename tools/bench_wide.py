"""Timing of the layered renderer (include/nsr_wide.h) on one 400x400 view: ms per view forward and forward + input gradient,
algorithmic fp32 TFLOP/s of the network evaluations against the fp32-MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md) -- the
roofline of its GEMM kernel (kw_gemm: fp32 in, v_mfma_f32_32x32x2_f32, fp32 out).

    python tools/bench_wide.py [--hw 400] [--cases ycbv,w512,d10w384,small] [--steps 3]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_sim_nerf_amd import synthetic as S  # noqa: E402
from neural_sim_nerf_amd.wide import WideModel  # noqa: E402
from bench import PowerSampler  # noqa: E402  (hwmon power / clock of the GPU while the launches run)

PEAK_FP32_MFMA = 157.3

# name -> (D, W, multires, multires_views, skips, N_samples, N_importance)
CASES = {
    "ycbv": (8, 256, 10, 4, [4], 64, 128),         # the fused kernels' own network, for the price of leaving them
    "w512": (8, 512, 10, 4, [4], 64, 128),
    "d10w384": (10, 384, 10, 4, [4], 64, 128),
    "w1024": (8, 1024, 10, 4, [4], 64, 128),
    "small": (4, 128, 6, 2, [1], 48, 100),
    # networks the FUSED kernels serve by re-expression at the 8 x 256 price (108 ms per view): is the layered renderer cheaper?
    "d8w64": (8, 64, 10, 4, [4], 64, 128),
    "d8w128": (8, 128, 10, 4, [4], 64, 128),
    "d8w192": (8, 192, 10, 4, [4], 64, 128),
    "d4w256": (4, 256, 10, 4, [], 64, 128),
}


def net_sd(D, W, L, Lv, skips, seed):
    rng = np.random.RandomState(seed)
    in_ch, in_v = 3 + 6 * L, 3 + 6 * Lv
    sd = {}

    def lin(name, o, i, scale=1.0):
        b = 1.0 / np.sqrt(i)
        sd[name + ".weight"] = (rng.uniform(-b, b, (o, i)) * scale).astype(np.float32)
        sd[name + ".bias"] = rng.uniform(-b, b, (o,)).astype(np.float32)
    g = 1.6 * np.sqrt(256.0 / W)
    lin("pts_linears.0", W, in_ch, g)
    for i in range(D - 1):
        lin("pts_linears.%d" % (i + 1), W, W + in_ch if i in skips else W, g)
    lin("feature_linear", W, W)
    lin("alpha_linear", 1, W, 50.0 * 256.0 / W)
    sd["alpha_linear.bias"][:] = -0.5
    lin("views_linears.0", W // 2, W + in_v)
    lin("rgb_linear", 3, W // 2)
    return sd


def flop_per_point(sd):
    return 2 * sum(v.size for k, v in sd.items() if k.endswith(".weight"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hw", type=int, default=400)
    ap.add_argument("--cases", default="ycbv,w512,d10w384,small")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--no-grad", action="store_true")
    ap.add_argument("--mlp", default=None, help="bf16x3 | fp32 | f16x2 (default: the library's)")
    a = ap.parse_args()
    K = S.scaled_K(400.0 / a.hw)
    pose = S.sweep_poses(1, seed=0)[0]
    out = {}
    for name in a.cases.split(","):
        D, W, L, Lv, skips, ns, ni = CASES[name]
        sd = net_sd(D, W, L, Lv, skips, 1)
        m = WideModel(sd, sd, n_samples=ns, n_importance=ni, mlp=a.mlp)
        n = a.hw * a.hw
        evals = n * (ns + ns + ni)
        flop = evals * flop_per_point(sd)
        ro, rd = m.get_rays(a.hw, a.hw, K, torch.tensor(pose[:3, :4]))
        ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
        cot = torch.randn(n, 3, device=m.device)
        res = {"network": "%d x %d, skips %s, %d + %d samples" % (D, W, skips, ns, ni), "rays": n,
               "flop_per_point": flop_per_point(sd), "mlp": m.mlp}
        for what in ("forward",) + (() if a.no_grad else ("forward+input-gradient",)):
            ms = []
            ps = None
            for it in range(a.steps + 1):
                if it == 1:
                    ps = PowerSampler(m.device.index)
                    ps.start()
                if what == "forward":
                    m.render_rays(ro, rd, S.YCBV_NEAR, S.YCBV_FAR)
                else:
                    m.render_rays_vjp(ro, rd, S.YCBV_NEAR, S.YCBV_FAR, cot)
                t, chunks = m.last_kernel_ms()
                ms.append(t)
            power = ps.stop() if ps else None
            t = float(np.median(ms[1:]))
            f = flop if what == "forward" else flop + n * (ns + ni) * flop_per_point(sd)     # + the fine pass's transposed GEMMs
            res[what] = {"ms_per_view": round(t, 2), "chunks": chunks, "workspace_GB": round(m.workspace_bytes / 2 ** 30, 2),
                         "algorithmic_TFLOPs": round(f / t / 1e9, 1), "frac_of_fp32_mfma_peak": round(f / t / 1e9 / PEAK_FP32_MFMA, 3)}
            if m.mlp.endswith("bf16x3"):       # six bf16 piece products per product: ceiling 2500 / 6 TFLOP/s of algorithmic work
                res[what]["issued_frac_of_bf16_peak"] = round(6 * f / t / 1e9 / 2500.0, 3)
            if m.mlp.endswith("f16x2") and what == "forward":       # three fp16 piece products per product (the gradient GEMMs are bf16x3)
                res[what]["issued_frac_of_fp16_peak"] = round(3 * f / t / 1e9 / 2500.0, 3)
            if power:
                res[what]["power_and_clock"] = {k: v for k, v in power.items() if k != "source"}
            if what == "forward":
                res[what]["Mray_samples_per_s"] = round(n * (ns + ni) / t / 1e3, 2)
        if m.mlp.endswith("f16x2"):
            res["range_status"] = m.range_status()
        out[name] = res
        print(name, json.dumps(res), flush=True)
        m.close()
        del m
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
