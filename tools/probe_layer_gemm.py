import sys, ctypes as C
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from neural_sim_nerf_amd import synthetic as S, _lib
from neural_sim_nerf_amd.engine import NsrModel
m = NsrModel(S.synth_weights(0), None, n_importance=0)
iters = 2000
for mode in (0, 1, 2, 3, 0, 1, 2, 3):
    ms = C.c_float()
    _lib.check(m.lib.nsr_probe(m.h, mode, iters, C.byref(ms), None))
    flop = 256 * 4 * iters * 1024 * 4096.0      # mode 3: twice the waves, half the FLOP per MFMA -> same total
    print("mode", mode, "ms %.2f" % ms.value, "TFLOP/s %.1f" % (flop / ms.value / 1e9))
