"""The 256x256 layer GEMM on fp16 MFMAs with two-way split fp32 operands (three piece products per fp32 product) in isolation:
mode 28 = the production gemm_h2 with the layer epilogue, 30 = without the epilogue, 29 = its MFMAs alone (operands in
registers).  Prints fp32-EQUIVALENT TFLOP/s (algorithmic FLOPs of the layer / time, whole chip) and, with NSR_PROBE_VERBOSE=1
(set here), cycles per MFMA and the clock workgroup 0 ran at."""
import os, sys
os.environ.setdefault("NSR_PROBE_VERBOSE", "1")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _probe import probe
iters = int(os.environ.get("ITERS", "4000"))
for mode in [int(m) for m in os.environ.get("MODES", "29,30,28,29,30,28").split(",")]:
    ms = probe(mode, iters)
    flop = 256 * 4 * iters * 2.0 * 256 * 256 * 32          # 256 CUs x 4 waves x iters layers x 2 K N M(points)
    print("mode %2d  ms %8.2f  fp32-equivalent TFLOP/s %7.1f   (x3 = fp16 MFMA work issued: %7.1f)" % (mode, ms, flop / ms / 1e9, 3 * flop / ms / 1e9))
