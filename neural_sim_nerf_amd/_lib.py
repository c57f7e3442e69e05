"""ctypes binding of csrc/libnsr.so (C ABI: include/nsr.h).  This is the binding a maintainer of the reference
would add next to utils/run_nerf_noscale.py.  No fallback: if the library is missing or
was not built for the device at hand, every entry point raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NSR_LIB_PATH", os.path.join(_HERE, "csrc", "libnsr.so"))   # override: A/B builds

ABI_VERSION = 5
PACKED_FLOATS = 145 * 4096 + 3328


class NsrConfig(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("n_samples", C.c_int32),
                ("n_importance", C.c_int32), ("max_workgroups", C.c_int32), ("variant", C.c_int32), ("flags", C.c_int32), ("chunk", C.c_int32)]


class NsrDebugOut(C.Structure):
    _fields_ = [("d_weights0", C.c_void_p), ("d_z_samples", C.c_void_p), ("d_inds", C.c_void_p),
                ("d_z_fine", C.c_void_p), ("d_raw0", C.c_void_p), ("d_raw", C.c_void_p)]


class NsrRenderOut(C.Structure):
    _fields_ = [("d_rgb", C.c_void_p), ("d_disp", C.c_void_p), ("d_acc", C.c_void_p), ("d_rgb0", C.c_void_p),
                ("d_disp0", C.c_void_p), ("d_acc0", C.c_void_p), ("d_z_std", C.c_void_p)]


class NsrRayExtras(C.Structure):
    _fields_ = [("d_viewdirs", C.c_void_p), ("d_t_rand", C.c_void_p), ("d_u", C.c_void_p), ("d_noise0", C.c_void_p),
                ("d_noise1", C.c_void_p), ("d_near", C.c_void_p), ("d_far", C.c_void_p)]


class NsrVjpDebugOut(C.Structure):
    _fields_ = [("d_relu_masks", C.c_void_p), ("d_grad_raw", C.c_void_p), ("d_grad_pts", C.c_void_p)]


# name -> (restype, argtypes); every symbol include/nsr.h declares
SIGNATURES = {
    "nsr_last_error": (C.c_char_p, []),
    "nsr_abi_version": (C.c_int, []),
    "nsr_create": (C.c_int, [C.POINTER(NsrConfig), C.POINTER(C.c_void_p)]),
    "nsr_destroy": (C.c_int, [C.c_void_p]),
    "nsr_upload_weights": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_size_t]),
    "nsr_upload_weights16": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_size_t]),
    "nsr_upload_weights_b3": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_size_t]),
    "nsr_upload_weights_bwd_b3": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_size_t]),
    "nsr_upload_weights_h2": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_size_t]),
    "nsr_upload_weights_bwd_h2": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_size_t]),
    "nsr_upload_weights_bwd": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_size_t]),
    "nsr_upload_weights_bwd16": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_size_t]),
    "nsr_upload_tables": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.c_int]),
    "nsr_render_rays": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                  C.POINTER(NsrRenderOut), C.POINTER(NsrDebugOut), C.c_void_p]),
    "nsr_render_rays_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                     C.POINTER(NsrRayExtras), C.POINTER(NsrRenderOut), C.POINTER(NsrDebugOut), C.c_void_p]),
    "nsr_render_rays_vjp_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                         C.POINTER(NsrRayExtras), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.POINTER(NsrRenderOut), C.c_void_p]),
    "nsr_render_rays_vjp_dbg": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                          C.POINTER(NsrRayExtras), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.POINTER(NsrRenderOut), C.POINTER(NsrVjpDebugOut), C.c_void_p]),
    "nsr_range_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint),
                                   C.POINTER(C.c_uint)]),
    "nsr_reserve_range": (C.c_int, [C.c_void_p, C.c_int64]),
    "nsr_ndc_rays": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_double,
                               C.c_void_p, C.c_void_p, C.c_void_p]),
    "nsr_ndc_rays_vjp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_double,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nsr_render_views": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double),
                                   C.c_float, C.c_float, C.POINTER(NsrRenderOut), C.POINTER(NsrDebugOut),
                                   C.c_void_p]),
    "nsr_render_rays_vjp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(NsrRenderOut),
                                      C.c_void_p]),
    "nsr_pose_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double),
                                C.c_int, C.c_void_p, C.c_void_p]),
    "nsr_get_rays": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p,
                               C.c_void_p, C.c_void_p]),
    "nsr_get_rays_views": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "nsr_to8b": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "nsr_find_bbox": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p]),
    "nsr_sample_pose": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double,
                                  C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nsr_sample_pose_nograd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                         C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nsr_embed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "nsr_run_network": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "nsr_raw2outputs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nsr_sample_pdf": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nsr_sort_merge": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "nsr_selftest": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nsr_reserve_bbox": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "nsr_fingerprint": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "nsr_schedule_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint)]),
    "nsr_debug_bounds_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_uint)]),
    "nsr_last_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
}

_lib = None

KERNEL_SOURCES = ("nsr_device.h", "nsr_kernels.hip", "nsr_b3.inc", "nsr_h2.inc", "nsr_h2_bwd.inc", "nsr_handoff.hip", "nsr_pose.hip", "nsr_api.hip")


def kernel_source_hash():
    """sha256 over the sources libnsr.so is built from: ties a committed rocprofv3 profile to the kernels it measured
    (bench.py reports a PMC-derived figure only when this matches the hash recorded with the profile)."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(_HERE, "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()


class NsrError(RuntimeError):
    pass


def load():
    """dlopen the library and bind every symbol.  Raises NsrError if it is missing (no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NsrError("libnsr.so not found at %s -- build it first (python -c 'import __graft_entry__ as g; "
                       "g.build()' or make -C neural_sim_nerf_amd/csrc); there is no CPU fallback" % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so (SONAME libamdhip64.so.7, the
    # same SONAME libnsr.so was linked against).  Loading it FIRST makes the dynamic loader bind libnsr.so to
    # that very instance, so device pointers and hipStream_t handles mean the same thing on both sides.  The
    # other order would put two runtimes in the process and the second one finds no device.
    import torch
    torch_hip = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(torch_hip):
        C.CDLL(torch_hip, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so is stale
        fn.restype, fn.argtypes = res, args
    if lib.nsr_abi_version() != ABI_VERSION:
        raise NsrError("libnsr.so ABI version %d, binding expects %d" % (lib.nsr_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise NsrError(load().nsr_last_error().decode("utf-8", "replace"))
