"""Layer GEMM + epilogue (relu, re-bias) per wave: x32 as the kernels run it (nsr_probe mode 10) vs the two-tile
16x16x4 scheme that interleaves one tile's epilogue with the other tile's MFMAs (mode 9); mode 2 = x32 GEMM only."""
import sys, ctypes as C
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from neural_sim_nerf_amd import synthetic as S, _lib
from neural_sim_nerf_amd.engine import NsrModel
m = NsrModel(S.synth_weights(0), None, n_importance=0)
iters = 2000
for mode in (2, 10, 9, 2, 10, 9):
    ms = C.c_float()
    _lib.check(m.lib.nsr_probe(m.h, mode, iters, C.byref(ms), None))
    flop = 256 * 4 * iters * 1024 * 4096.0
    print("mode %2d  ms %.2f  TFLOP/s %.1f" % (mode, ms.value, flop / ms.value / 1e9))
