"""ctypes loader of csrc/libnsr_probe.so (include/nsr_probe.h; `make -C neural_sim_nerf_amd/csrc probe`)."""
import ctypes as C, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None
def probe(mode, iters, device=0, partner_prio=0):
    """-> ms of nsr_probe(mode, iters)."""
    global _lib
    if _lib is None:
        import torch                                              # one HIP runtime per process (see _lib.py)
        hip = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(hip):
            C.CDLL(hip, mode=C.RTLD_GLOBAL)
        _lib = C.CDLL(os.path.join(ROOT, "neural_sim_nerf_amd", "csrc", "libnsr_probe.so"))
        _lib.nsr_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
        _lib.nsr_probe_last_error.restype = C.c_char_p
    ms = C.c_float()
    if _lib.nsr_probe(device, mode, iters, partner_prio, C.byref(ms)) != 0:
        raise RuntimeError(_lib.nsr_probe_last_error().decode())
    return ms.value
