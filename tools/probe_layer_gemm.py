import sys, ctypes as C
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from _probe import probe
iters = 2000
for mode in (0, 1, 2, 3, 0, 1, 2, 3):
    ms = C.c_float(probe(mode, iters, partner_prio=int(__import__('os').environ.get('NSR_PROBE_PARTNER_PRIO', '0'))))
    flop = 256 * 4 * iters * 1024 * 4096.0      # mode 3: twice the waves, half the FLOP per MFMA -> same total
    print("mode", mode, "ms %.2f" % ms.value, "TFLOP/s %.1f" % (flop / ms.value / 1e9))
