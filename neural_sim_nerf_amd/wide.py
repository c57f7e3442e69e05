"""WideModel: a handle of the LAYERED renderer in libnsr.so (C ABI: include/nsr_wide.h) -- render_rays (RN:390-501) and its
input-side VJP (RN:168-178) for what the fused kernels are not built for: a NeRF (RH:70-122) of any depth / width / skip
list, any N_samples / N_importance (RN:439, RN:474).  Same interface as engine.NsrModel where the drop-in API
(run_nerf_noscale.py) touches it, so `_model_for` can hand out either.

One fp32-MFMA GEMM kernel per layer over a chunk of rays, activations resident in HBM between the layers; torch owns the
buffers (including the workspace) and the stream, nothing else.  No fallback: without the library or a GPU every call raises."""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib

MAX_SKIPS = 16
MAX_SAMPLES = 512


class NsrwConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("n_samples", C.c_int32), ("n_importance", C.c_int32), ("flags", C.c_int32),
                ("reserved", C.c_int32 * 4)]


class NsrwNet(C.Structure):
    _fields_ = [("D", C.c_int32), ("W", C.c_int32), ("multires", C.c_int32), ("multires_views", C.c_int32),
                ("use_viewdirs", C.c_int32), ("output_ch", C.c_int32), ("n_skips", C.c_int32), ("skips", C.c_int32 * MAX_SKIPS)]


class NsrwExtras(C.Structure):
    _fields_ = [("d_viewdirs", C.c_void_p), ("d_near", C.c_void_p), ("d_far", C.c_void_p), ("d_t_rand", C.c_void_p),
                ("d_u", C.c_void_p), ("d_noise0", C.c_void_p), ("d_noise1", C.c_void_p), ("d_z_fine", C.c_void_p)]


class NsrwOut(C.Structure):
    _fields_ = [("d_rgb", C.c_void_p), ("d_disp", C.c_void_p), ("d_acc", C.c_void_p), ("d_rgb0", C.c_void_p),
                ("d_disp0", C.c_void_p), ("d_acc0", C.c_void_p), ("d_z_std", C.c_void_p), ("d_raw", C.c_void_p),
                ("d_z_vals", C.c_void_p), ("d_weights0", C.c_void_p), ("d_z_samples", C.c_void_p), ("d_inds", C.c_void_p),
                ("d_raw0", C.c_void_p)]


# name -> (restype, argtypes); every symbol include/nsr_wide.h declares
SIGNATURES = {
    "nsrw_last_error": (C.c_char_p, []),
    "nsrw_create": (C.c_int, [C.POINTER(NsrwConfig), C.POINTER(C.c_void_p)]),
    "nsrw_destroy": (C.c_int, [C.c_void_p]),
    "nsrw_upload_network": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(NsrwNet), C.POINTER(C.c_float), C.c_size_t]),
    "nsrw_network_floats": (C.c_size_t, [C.POINTER(NsrwNet)]),
    "nsrw_upload_tables": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.c_int]),
    "nsrw_workspace_bytes": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_size_t)]),
    "nsrw_render_rays": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.POINTER(NsrwExtras),
                                   C.POINTER(NsrwOut), C.c_void_p, C.c_size_t, C.c_void_p]),
    "nsrw_render_rays_vjp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                       C.POINTER(NsrwExtras), C.c_void_p, C.POINTER(NsrwOut), C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_size_t, C.c_void_p]),
    "nsrw_run_network": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t,
                                   C.c_void_p]),
    "nsrw_last_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "nsrw_debug_bounds_status": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_uint)]),
    "nsrw_range_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "nsrw_sample_pdf": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "nsrw_embed_vjp": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
}

FLAG_WHITE_BKGD, FLAG_LINDISP, FLAG_MLP_BF16X3, FLAG_MLP_F16X2 = 1, 2, 4, 8
MLPS = ("bf16x3", "fp32", "f16x2")

DEFAULT_MLP = "f16x2"           # r06: the fused default kernel's arithmetic; same parity bounds as the fp32 MFMAs and bf16x3 (tests/test_gpu_wide.py
                                # runs every case on all three), 2.6x / 1.5x their speed; never range-dependent (re-run on bf16x3)

_bound = None


def load():
    """The nsrw_* entry points of libnsr.so, bound; raises NsrError when the library is missing or stale."""
    global _bound
    if _bound is None:
        lib = _lib.load()
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                raise _lib.NsrError("libnsr.so does not export %s -- rebuild it (python -c 'import __graft_entry__ as g; g.build()')"
                                    % name)
            fn.restype, fn.argtypes = res, args
        _bound = lib
    return _bound


def check(rc):
    if rc != 0:
        raise _lib.NsrError(load().nsrw_last_error().decode("utf-8", "replace"))


def _np(v):
    return (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)).astype(np.float32)


def describe(sd):
    """State dict with the reference's parameter names (RH:82-97) -> (NsrwNet, flat fp32 parameter block in the order
    include/nsr_wide.h: nsrw_upload_network names).  The shape is read off the weights (layer i + 1 takes W + input_ch inputs
    exactly when i is a skip, RH:82-83; `output_linear` present = use_viewdirs=False, RH:95-96)."""
    a = {k: _np(v) for k, v in sd.items()}
    D = 1 + max(int(k.split(".")[1]) for k in a if k.startswith("pts_linears."))
    W, input_ch = a["pts_linears.0.weight"].shape
    skips = [i for i in range(D - 1) if a["pts_linears.%d.weight" % (i + 1)].shape[1] == W + input_ch]
    use_viewdirs = "output_linear.weight" not in a
    if (input_ch - 3) % 6:
        raise NotImplementedError("input_ch %r is not 3 + 6 L (get_embedder RH:51-66)" % (input_ch,))
    in_v = a["views_linears.0.weight"].shape[1] - W if "views_linears.0.weight" in a else 3
    if use_viewdirs and (in_v < 3 or (in_v - 3) % 6):
        raise NotImplementedError("input_ch_views %r is not 3 + 6 L" % (in_v,))
    if len(skips) > MAX_SKIPS:
        raise NotImplementedError("%d skips (at most %d)" % (len(skips), MAX_SKIPS))
    net = NsrwNet(D, W, (input_ch - 3) // 6, (in_v - 3) // 6 if use_viewdirs else 0, 1 if use_viewdirs else 0,
                  4 if use_viewdirs else int(a["output_linear.weight"].shape[0]), len(skips),
                  (C.c_int32 * MAX_SKIPS)(*(skips + [0] * (MAX_SKIPS - len(skips)))))
    names = ["pts_linears.%d" % i for i in range(D)]
    names += ["feature_linear", "alpha_linear", "views_linears.0", "rgb_linear"] if use_viewdirs else ["output_linear"]
    flat = np.concatenate([np.concatenate([a[n + ".weight"].ravel(), a[n + ".bias"].ravel()]) for n in names]).astype(np.float32)
    return net, np.ascontiguousarray(flat)


def _dev(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def workspace_cap_env():
    """Upper bound of the workspace a launch call may use: $NSR_WIDE_WORKSPACE_GB, else 64 GiB (r06: of 288 GB; 16 GiB cost a 400x400 forward + gradient call
    6-8 % in chunk overheads, profiles/r06/extra/ab_wide_h2_w512.txt) -- and never more than half of the free memory
    (workspace_cap_bytes).  A smaller workspace means more,
    smaller chunks of rays -- never another result."""
    return int(float(os.environ.get("NSR_WIDE_WORKSPACE_GB", "64")) * (1 << 30))


def workspace_cap_bytes(device, reusable=0):
    """... and never more than half of what the device has free (`reusable`: bytes of a buffer that is about to be released).
    Only consulted when the shared workspace has to GROW (WideModel._workspace): one driver query per growth, not per launch."""
    free, _ = torch.cuda.mem_get_info(device)
    return min(workspace_cap_env(), (free + reusable) // 2)


def _cuda_f32(x, device=None):
    t = torch.as_tensor(x, dtype=torch.float32)
    if not t.is_cuda:
        if not torch.cuda.is_available():
            raise _lib.NsrError("no HIP device visible: the render path has no CPU fallback")
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    return t.contiguous()


def sample_pdf(bins, weights, u):
    """nsrw_sample_pdf: sample_pdf (RH:199-243) for any bin / sample count.  bins [..., B], weights [..., B - 1]; u: [S] (shared by
    all rows: det=True) or [..., S] (det=False).  -> (samples [..., S] float32, inds [..., S] int64) on the device."""
    lib = load()
    bins = _cuda_f32(bins)
    dev = bins.device
    weights, u = _cuda_f32(weights, dev), _cuda_f32(u, dev)
    lead, B = tuple(bins.shape[:-1]), int(bins.shape[-1])
    if tuple(weights.shape) != lead + (B - 1,):
        raise ValueError("sample_pdf: weights must be bins' shape with one entry less (got %r, %r)" % (tuple(bins.shape), tuple(weights.shape)))
    S = int(u.shape[-1])
    per_row = u.dim() > 1
    if per_row and tuple(u.shape[:-1]) != lead:
        raise ValueError("sample_pdf: per-row uniforms must have bins' leading shape")
    if not 2 <= B <= MAX_SAMPLES or not 1 <= S <= MAX_SAMPLES:
        raise NotImplementedError("sample_pdf: 2..%d bins and 1..%d samples (got %d, %d)" % (MAX_SAMPLES, MAX_SAMPLES, B, S))
    n = int(np.prod(lead)) if lead else 1
    samples = torch.empty(lead + (S,), dtype=torch.float32, device=dev)
    inds = torch.empty(lead + (S,), dtype=torch.int64, device=dev)
    scratch = torch.empty(2 * n * (B + 1), dtype=torch.float32, device=dev)
    check(lib.nsrw_sample_pdf(dev.index, _dev(bins), _dev(weights), n, B, _dev(u), 1 if per_row else 0, S, _dev(samples), _dev(inds),
                              _dev(scratch), _stream_ptr(dev)))
    return samples, inds


def embed_vjp(x, grad_out, multires):
    """nsrw_embed_vjp: d(sum(embed(x) * grad_out)) / dx for x [..., 3], grad_out [..., 3 + 6 multires]."""
    lib = load()
    x = _cuda_f32(x)
    g = _cuda_f32(grad_out, x.device)
    if x.shape[-1:] != (3,) or tuple(g.shape) != tuple(x.shape[:-1]) + (3 + 6 * int(multires),):
        raise ValueError("embed_vjp: x [..., 3] and grad_out [..., 3 + 6 multires] (got %r, %r)" % (tuple(x.shape), tuple(g.shape)))
    gx = torch.empty_like(x)
    n = x.numel() // 3
    check(lib.nsrw_embed_vjp(x.device.index, _dev(x), _dev(g), n, int(multires), _dev(gx), _stream_ptr(x.device)))
    return gx


# ONE workspace per (device, stream), shared by every handle that launches there: a handle's launch calls are stream-ordered, so
# two handles on one stream can never use it at the same time, and the drop-in API's cache of up to eight handles per module
# (render options x streams) does not multiply a buffer of tens of GiB.  release_workspaces() hands the memory back to torch.
_WORKSPACES = {}


def release_workspaces():
    _WORKSPACES.clear()


class WideModel:
    variant = 0
    schedule = "layers"

    def __init__(self, sd_coarse, sd_fine=None, device=None, n_importance=128, white_bkgd=False, lindisp=False, n_samples=64,
                 mlp=None):
        """mlp: the arithmetic of the layer GEMMs -- "bf16x3" (bf16 MFMAs on three-way split fp32 operands, fp32-grade results at
        2.67x the matrix-pipe rate: csrc/nsr_wide_b3.inc), "fp32" (fp32 MFMAs, the strict mode) or "f16x2" (fp16 MFMAs with
        two-piece operands -- the fused default kernel's arithmetic; a network pass that leaves fp16's range is re-run on bf16x3
        inside the call); default $NSR_WIDE_MLP, else DEFAULT_MLP."""
        mlp = mlp or os.environ.get("NSR_WIDE_MLP") or DEFAULT_MLP
        if mlp not in MLPS:
            raise ValueError("mlp must be one of %s (got %r)" % (MLPS, mlp))
        self.mlp = "layered-" + mlp
        if not torch.cuda.is_available():
            raise _lib.NsrError("no HIP device visible: the render path has no CPU fallback")
        self.lib = load()
        if isinstance(device, torch.device):
            device = device.index
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        n_samples, n_importance = int(n_samples), int(n_importance)
        if not 3 <= n_samples <= MAX_SAMPLES or not 0 <= n_importance <= MAX_SAMPLES:
            raise NotImplementedError("N_samples must be 3..%d and N_importance 0..%d (got %r, %r)"
                                      % (MAX_SAMPLES, MAX_SAMPLES, n_samples, n_importance))
        self.n_samples, self.n_importance = n_samples, n_importance
        self.ni_kernel = n_importance
        self.u_width = n_importance                       # rows of the u draws as nsrw_render_rays reads them
        self.nf_kernel = n_samples + n_importance
        self.white_bkgd, self.lindisp = bool(white_bkgd), bool(lindisp)
        self.rays_launched = 0
        self.workspace_bytes = 0
        self._util = None
        self._ws_need = {}                                # (rays, grad) -> bytes one chunk of that many rays needs
        cfg = NsrwConfig(self.device.index, n_samples, n_importance,
                         (FLAG_WHITE_BKGD if white_bkgd else 0) | (FLAG_LINDISP if lindisp else 0) |
                         (FLAG_MLP_BF16X3 if mlp == "bf16x3" else 0) | (FLAG_MLP_F16X2 if mlp == "f16x2" else 0))
        h = C.c_void_p()
        check(self.lib.nsrw_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self.upload(sd_coarse, sd_fine)
        t = torch.linspace(0., 1., steps=n_samples).numpy().astype(np.float32)          # RN:439: on the host, then moved
        u = torch.linspace(0., 1., steps=max(n_importance, 1)).numpy().astype(np.float32)   # RH:208
        check(self.lib.nsrw_upload_tables(self.h, t.ctypes.data_as(C.POINTER(C.c_float)), n_samples,
                                          u.ctypes.data_as(C.POINTER(C.c_float)), n_importance))

    def upload(self, sd_coarse, sd_fine=None):
        self.nets = []
        for net_id, sd in enumerate((sd_coarse, sd_fine)):
            if sd is None:
                continue
            net, flat = describe(sd)
            check(self.lib.nsrw_upload_network(self.h, net_id, C.byref(net), flat.ctypes.data_as(C.POINTER(C.c_float)), flat.size))
            self.nets.append(net)
        last = self.nets[-1] if self.n_importance > 0 else self.nets[0]
        self.raw_ch = 4 if last.use_viewdirs else int(last.output_ch)
        self.raw_ch0 = 4 if self.nets[0].use_viewdirs else int(self.nets[0].output_ch)

    def close(self):
        if getattr(self, "h", None):
            self.lib.nsrw_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers ------------------------------------------------------------------------------------
    def _f32(self, x, shape=None):
        t = torch.as_tensor(x, dtype=torch.float32, device=self.device).contiguous()
        return t.reshape(shape) if shape is not None else t

    def _new(self, *shape, dtype=torch.float32):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def _need(self, n_rays, grad):
        """bytes one chunk of n_rays rays needs (cached per handle: the bilevel loop asks for the same 512 rays 313 times a pose)"""
        key = (int(n_rays), bool(grad))
        v = self._ws_need.get(key)
        if v is None:
            need = C.c_size_t()
            check(self.lib.nsrw_workspace_bytes(self.h, key[0], 1 if grad else 0, C.byref(need)))
            v = self._ws_need[key] = int(need.value)
        return v

    def _workspace(self, n_rays, grad):
        """(shared workspace of this stream, bytes of it the launch call may use).  A buffer that already holds the call's one
        chunk -- or is as large as the cap allowed when it was made -- is reused without a driver query (the cap used
        to be recomputed from mem_get_info on every launch, AFTER the workspace itself had shrunk the free memory)."""
        floor = self._need(64, grad)
        need = max(min(self._need(n_rays, grad), workspace_cap_env()), floor)
        key = (self.device.index, torch.cuda.current_stream(self.device).cuda_stream)
        ws, capped = _WORKSPACES.get(key, (None, False))
        have = ws.numel() if ws is not None else 0
        if have < need and not (capped and have >= floor):
            want = max(min(need, workspace_cap_bytes(self.device, have)), floor)
            if want > have:
                _WORKSPACES.pop(key, None)                     # release before asking for the larger one
                ws = None
                ws = torch.empty(want, dtype=torch.uint8, device=self.device)
                have = want
            _WORKSPACES[key] = (ws, have < need)               # capped: as large as the device allows -- do not ask again
        use = min(need, have)
        self.workspace_bytes = use                             # what the launch call is told it may use (<= the shared buffer)
        return ws, use

    def _extras(self, extras, n):
        if not extras or all(v is None for v in extras.values()):
            return None, None
        widths = dict(viewdirs=3, near=1, far=1, t_rand=self.n_samples, u=self.n_importance, noise0=self.n_samples,
                      noise1=self.nf_kernel, z_fine=self.nf_kernel)
        unknown = set(extras) - set(widths)
        if unknown:
            raise ValueError("unknown ray extras: %s" % sorted(unknown))
        keep = {k: self._f32(v, (n, widths[k])) for k, v in extras.items() if v is not None}
        if ("near" in keep) != ("far" in keep):
            raise ValueError("per-ray bounds: near and far come together")
        ex = NsrwExtras(*[_dev(keep.get(k)) for k in ("viewdirs", "near", "far", "t_rand", "u", "noise0", "noise1", "z_fine")])
        return ex, keep

    def _outs(self, n, debug):
        fine = self.n_importance > 0
        o = dict(rgb_map=self._new(n, 3), disp_map=self._new(n), acc_map=self._new(n))
        if fine:
            o.update(rgb0=self._new(n, 3), disp0=self._new(n), acc0=self._new(n), z_std=self._new(n))
        d = {}
        if debug:                                   # engine.NsrModel's tap names
            d = dict(weights0=self._new(n, self.n_samples), raw0=self._new(n, self.n_samples, self.raw_ch0))
            if fine:
                d.update(z_samples=self._new(n, self.n_importance), inds=self._new(n, self.n_importance, dtype=torch.int64),
                         z_fine=self._new(n, self.nf_kernel), raw=self._new(n, self.nf_kernel, self.raw_ch))
            else:
                d.update(z_coarse=self._new(n, self.n_samples))
            o.update(d)
        last_raw, last_z = (d.get("raw"), d.get("z_fine")) if fine else (d.get("raw0"), d.get("z_coarse"))
        out = NsrwOut(_dev(o["rgb_map"]), _dev(o["disp_map"]), _dev(o["acc_map"]), _dev(o.get("rgb0")), _dev(o.get("disp0")),
                      _dev(o.get("acc0")), _dev(o.get("z_std")), _dev(last_raw), _dev(last_z), _dev(d.get("weights0")),
                      _dev(d.get("z_samples")), _dev(d.get("inds")), _dev(d.get("raw0")) if fine else None)
        return o, out

    @property
    def util(self):
        """the fused library's handle for the network-free stage kernels (get_rays, pose_grad, ndc_rays)"""
        if self._util is None:
            from .run_nerf_noscale import _util_model
            self._util = _util_model(self.device)
        return self._util

    # ---- the path -----------------------------------------------------------------------------------
    def render_rays(self, rays_o, rays_d, near, far, debug=False, extras=None):
        """render(rays=...) (RN:58-123): rays_o, rays_d [N,3] -> dict of [N,...] tensors on the device (engine.NsrModel's keys;
        debug adds raw / z_fine / weights0 / z_samples / inds)."""
        rays_o, rays_d = self._f32(rays_o, (-1, 3)), self._f32(rays_d, (-1, 3))
        n = rays_o.shape[0]
        self.rays_launched += n
        o, out = self._outs(n, debug)
        ex, keep = self._extras(extras, n)
        ws, nbytes = self._workspace(n, False)
        check(self.lib.nsrw_render_rays(self.h, _dev(rays_o), _dev(rays_d), n, float(near), float(far), C.byref(ex) if ex else None,
                                        C.byref(out), _dev(ws), nbytes, _stream_ptr(self.device)))
        return o

    def render_views(self, c2w, H, W, K, near, far, debug=False):
        """render(c2w=...) for V views: get_rays (RH:156-165) for all of them in ONE launch into one buffer (nsr_get_rays_views;
        r05 ran a launch per view and concatenated), then ONE render_rays call."""
        ro, rd = self.util.get_rays_views(int(H), int(W), K, self._f32(c2w))
        return self.render_rays(ro, rd, near, far, debug=debug)

    def render_rays_vjp(self, rays_o, rays_d, near, far, grad_rgb, with_forward=False, z_fine=None, extras=None, debug=False):
        """Forward + input-side VJP (RN:168-178): grad_rgb [N,3] -> (grad_rays_o, grad_rays_d) [N,3] each (+ grad_viewdirs with
        extras["viewdirs"], + the forward's rgb / disp / acc with with_forward)."""
        if debug:
            raise NotImplementedError("the layered renderer has no relu taps")
        if z_fine is not None:                      # given depths for the fine pass (constants of the gradient, RN:475)
            extras = dict(extras or {}, z_fine=z_fine)
        rays_o, rays_d = self._f32(rays_o, (-1, 3)), self._f32(rays_d, (-1, 3))
        n = rays_o.shape[0]
        self.rays_launched += n
        g = self._f32(grad_rgb, (n, 3))
        go, gd = self._new(n, 3), self._new(n, 3)
        fwd, out = None, None
        if with_forward:
            fwd = dict(rgb_map=self._new(n, 3), disp_map=self._new(n), acc_map=self._new(n))
            out = NsrwOut(_dev(fwd["rgb_map"]), _dev(fwd["disp_map"]), _dev(fwd["acc_map"]))
        ex, keep = self._extras(extras, n)
        gv = self._new(n, 3) if (keep and "viewdirs" in keep) else None
        ws, nbytes = self._workspace(n, True)
        check(self.lib.nsrw_render_rays_vjp(self.h, _dev(rays_o), _dev(rays_d), n, float(near), float(far),
                                            C.byref(ex) if ex else None, _dev(g), C.byref(out) if out else None, _dev(go), _dev(gd),
                                            _dev(gv), _dev(ws), nbytes, _stream_ptr(self.device)))
        res = (go, gd) if gv is None else (go, gd, gv)
        return res + (fwd,) if with_forward else res

    def run_network(self, pts, viewdirs, net_id=0):
        """run_network (RN:26-40): pts [..,3], viewdirs [..,3] (unit length) -> raw [.., 4 or output_ch]."""
        shape = tuple(torch.as_tensor(pts).shape[:-1])
        p = self._f32(pts, (-1, 3))
        net = self.nets[net_id]
        v = self._f32(viewdirs, (-1, 3)) if net.use_viewdirs else None
        ch = 4 if net.use_viewdirs else int(net.output_ch)
        raw = self._new(p.shape[0], ch)
        ws, nbytes = self._workspace(max(1, p.shape[0] // max(self.nf_kernel, 1) + 1), False)
        check(self.lib.nsrw_run_network(self.h, int(net_id), _dev(p), _dev(v), p.shape[0], _dev(raw), _dev(ws), nbytes,
                                        _stream_ptr(self.device)))
        return raw.reshape(shape + (ch,))

    # network-free stage kernels: the fused library's
    def get_rays(self, H, W, K, c2w):
        return self.util.get_rays(H, W, K, c2w)

    def pose_grad(self, grad_o, grad_d, H, W, K, patch):
        return self.util.pose_grad(grad_o, grad_d, H, W, K, patch)

    def ndc_rays(self, *a, **k):
        return self.util.ndc_rays(*a, **k)

    def ndc_rays_vjp(self, *a, **k):
        return self.util.ndc_rays_vjp(*a, **k)

    def range_status(self):
        """fp32 / bf16x3: no range to leave.  f16x2: `passes` network passes launched so far, `passes_rerun` of them re-run on
        bf16x3 because an activation reached fp16's largest number; nothing is ever dropped.  (The keys of NsrModel.range_status
        are kept: `points` / `rays` stay 0 -- the unit of the layered renderer's safety net is a pass over a chunk of rays.)"""
        out = dict(last_items=0, points=0, rays=0, dropped_items=0, passes=0, passes_rerun=0)
        if self.mlp == "layered-f16x2":
            a, b = C.c_ulonglong(), C.c_ulonglong()
            check(self.lib.nsrw_range_status(self.h, C.byref(a), C.byref(b)))
            out["passes"], out["passes_rerun"] = int(a.value), int(b.value)
        return out

    def debug_bounds_status(self):
        """(built_with_checks, first_bad_line): see nsrw_debug_bounds_status / `make debug`."""
        built, line = C.c_int(), C.c_uint()
        check(self.lib.nsrw_debug_bounds_status(C.byref(built), C.byref(line)))
        return bool(built.value), int(line.value)

    def last_kernel_ms(self):
        """(device ms of the last launch call, chunks of rays it ran)"""
        ms, ch = C.c_float(), C.c_int()
        check(self.lib.nsrw_last_ms(self.h, C.byref(ms), C.byref(ch)))
        return float(ms.value), int(ch.value)
