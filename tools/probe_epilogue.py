"""Layer GEMM + epilogue (relu, re-bias) per wave: x32 as the kernels run it (nsr_probe mode 10) vs the two-tile
16x16x4 scheme that interleaves one tile's epilogue with the other tile's MFMAs (mode 9); mode 2 = x32 GEMM only."""
import sys, ctypes as C
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from _probe import probe
iters = 2000
for mode in (2, 10, 9, 2, 10, 9):
    ms = C.c_float(probe(mode, iters, partner_prio=int(__import__('os').environ.get('NSR_PROBE_PARTNER_PRIO', '0'))))
    flop = 256 * 4 * iters * 1024 * 4096.0
    print("mode %2d  ms %.2f  TFLOP/s %.1f" % (mode, ms.value, flop / ms.value / 1e9))
