"""The 256x256 layer GEMM on bf16 MFMAs with three-way split fp32 operands (6 products), against the fp32-MFMA
production segment: fp32-EQUIVALENT TFLOP/s (algorithmic FLOPs of the layer / time), whole chip.
mode 2 = production x32 fp32 segment, 12 = bf16x3 MFMAs only (issue ceiling), 11 = bf16x3 with ring, split and epilogue."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _probe import probe
iters = int(os.environ.get("ITERS", "2000"))
for mode in [int(m) for m in os.environ.get("MODES", "2,12,11,2,12,11").split(",")]:
    ms = probe(mode, iters)
    flop = 256 * 4 * iters * 1024 * 4096.0
    print("mode %2d  ms %8.2f  fp32-equivalent TFLOP/s %7.1f" % (mode, ms, flop / ms / 1e9))
