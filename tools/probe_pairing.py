"""What the x16 layer GEMM of one workgroup loses when the CU's second workgroup runs something else
(nsr_probe modes 4..8).  Prints the GEMM workgroup's rate relative to the fp32 MFMA peak of its CU."""
import sys, ctypes as C
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from _probe import probe
iters = 1000
names = {3: "x16 GEMM x2 (both workgroups)", 4: "partner: none", 5: "partner: dense fp32 VALU", 6: "partner: sin/cos",
         7: "partner: LDS latency chain", 8: "partner: fp64 chain"}
import os
os.environ['NSR_PROBE_VERBOSE'] = '1'
for mode, iters in [(3, 1000), (4, 1000)] + [(k, n) for k in (5, 6, 7, 8) for n in (1, 1000)]:
    ms = C.c_float(probe(mode, iters, partner_prio=int(__import__('os').environ.get('NSR_PROBE_PARTNER_PRIO', '0'))))
    wgs = 2 if mode == 3 else 1
    flop = 256 * wgs * 4 * iters * 1024 * 2048.0
    print("mode %d iters %d %-32s ms %.2f  GEMM TFLOP/s %.1f" % (mode, iters, names[mode], ms.value, flop / ms.value / 1e9))
