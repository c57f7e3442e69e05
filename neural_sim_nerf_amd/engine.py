"""NsrModel: one native renderer handle (libnsr.so) = one (coarse, fine) NeRF pair on one GPU and stream.

PyTorch is plumbing here: it owns the device buffers (torch tensors -> raw device pointers) and the stream;
all arithmetic happens inside the hand-written gfx950 kernels.  Mirrors what the reference keeps in its
`render_kwargs` dict (network_fn, network_fine, N_samples, N_importance; RN:318-334)."""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .pack import (pack_network, pack_network16, pack_network_backward, pack_network_backward16, pack_network_b3,
                   pack_network_backward_b3, pack_network_h2, pack_network_backward_h2, h2_report,
                   PACKED_FLOATS, PACKED_B3_FLOATS)

N_SAMPLES = 64
N_IMPORTANCE = 128
# Work schedule of the x16 forward kernel (NSR_FLAG_SCHED_PHASES, include/nsr.h).  "phases" is the default: measured
# 328.1 vs 326.6 ms per 400x400 view (+0.5 %) for 4-5 GB instead of 20-90 GB of L2-miss (fabric) traffic per view
# (profiles/r02/pmc_k_render.json); results are bit-identical.
DEFAULT_SCHEDULE = "phases"
# Layer-GEMM arithmetic of the forward render kernel (NsrModel(mlp=...)).  r03: "f16x2" -- 112-115 ms per 400x400 view against
# 328 (fp32 MFMAs) and 192 (bf16x3), network outputs as close to an fp64 evaluation as the fp32-MFMA kernel's (4.9e-6 vs
# 4.6e-6 max error on the golden rays), every BASELINE config view inside the end-to-end acceptance rule with PSNR-delta
# 0.000 dB (tests/test_gpu_parity.py census tests, bench.py `parity`).  NSR_MLP=fp32 in the environment restores r02's default.
DEFAULT_MLP = "f16x2"
MLP_MODES = ("fp32", "bf16x3", "f16x2")


def _host_tables(n_importance=N_IMPORTANCE, native=False, n_samples=N_SAMPLES):
    """The reference builds both linspace tables on the HOST and moves them (RN:439, RH:208); torch's CPU
    linspace is not bit-equal to numpy's, so the same call is made here.
    n_importance < 128 (a divisor of 128): the kernels always draw 128 importance samples, from the 128 uniforms of this
    table -- filled with the reference's linspace(0, 1, n_importance), every value 128 / n_importance times.  The extra
    samples are exact duplicates: a duplicated depth is a zero-length interval (alpha = 1 - exp(0) = 0: no weight, and the
    transmittance moves by the 1e-10 of RN:376 only), and every value being repeated equally often leaves
    std(z_samples) (RN:495) what it was.  The render is the reference's N_importance = n render to 2.4e-7
    (tests/test_oracle_golden.py::test_fewer_importance_samples_by_duplicated_uniforms), at the price of all 192 fine
    evaluations.
    native: the handle's kernels are specialised to n_importance (f16x2 handles, 64 and 32: NATIVE_IMPORTANCE) -- 64 + n fine
    evaluations per ray, two network passes per item instead of three; the table is the reference's linspace itself."""
    n = n_importance if n_importance else N_IMPORTANCE
    u = torch.linspace(0., 1., steps=n)
    if native:                 # a kernel specialised to n importance samples reads the first n entries
        u = torch.cat([u, torch.zeros(N_IMPORTANCE - n)])
    else:
        u = u.repeat_interleave(N_IMPORTANCE // n)
    return torch.linspace(0., 1., steps=n_samples).numpy().astype(np.float32), u.numpy().astype(np.float32)


IMPORTANCE_COUNTS = (0, 1, 2, 4, 8, 16, 32, 64, 96, 128)   # 0 = coarse only; the divisors of the kernels' 128 (any handle) and 96 (f16x2)
NATIVE_IMPORTANCE = (96, 64, 32)   # f16x2 handles: kernels specialised to these counts (k_render_h2_n96 / _n64 / _n32 and their VJPs)
# (N_samples, N_importance) pairs beyond N_samples = 64 that f16x2 handles serve with kernels of their own (r05; RN:439, RN:474)
NATIVE_COUNTS = ((32, 64), (32, 0), (128, 128), (128, 0))


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dev(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class NsrModel:
    def __init__(self, sd_coarse, sd_fine=None, device=None, n_importance=N_IMPORTANCE, max_workgroups=0, variant=0,
                 white_bkgd=False, lindisp=False, chunk=None, schedule=None, mlp=None, range_fallback="bf16x3",
                 n_samples=N_SAMPLES):
        """sd_*: mappings with the reference's state_dict keys (RH:82-97) -> array-likes (numpy / torch cpu).
        white_bkgd / lindisp: the render options of RN:384-385 / RN:443 (both off in the YCB-V configuration).
        chunk: rays per work-queue chunk of the x16 kernel (None: $NSR_CHUNK, read HERE once, else the library default).
        schedule: "queue" (per-ray queue) or "phases" (global phases, NSR_FLAG_SCHED_PHASES); None: $NSR_SCHEDULE, else
        the library default.  Both give bit-identical results.
        mlp: arithmetic of the layer GEMMs of the FORWARD render kernel: "fp32" (fp32 MFMAs) or "bf16x3" (bf16 MFMAs on
        fp32 operands split exactly into three bf16 pieces, NSR_FLAG_MLP_BF16X3: fp32-grade results, ~1.9x the MFMA
        rate) or "f16x2" (fp16 MFMAs on fp32 operands split into two fp16 pieces with power-of-two range management,
        NSR_FLAG_MLP_F16X2: fp32-grade results at half of bf16x3's MFMA work; the input-gradient kernel of such a handle
        runs the same scheme with per-point normalised gradients); None: "fp32" when `variant` (16 / 32) or `schedule` is given -- they name
        fp32 forward kernels --, else $NSR_MLP, else DEFAULT_MLP.
        range_fallback (f16x2 handles): the arithmetic that re-renders the rays whose activations left the fp16 range --
        "bf16x3" (default: fp32's exponent range, fp32-grade error, 1.7x the fp32-MFMA kernels) or "fp32" (r04's route; kept
        for the A/B in bench.py's range_stress workload; N_importance 128 / 0 only)."""
        if range_fallback not in ("bf16x3", "fp32"):
            raise ValueError("range_fallback must be 'bf16x3' or 'fp32'")
        self.range_fallback = range_fallback
        if not torch.cuda.is_available():
            raise _lib.NsrError("no HIP device visible: the render path has no CPU fallback")
        self.lib = _lib.load()
        if isinstance(device, torch.device):
            device = device.index                       # torch.device("cuda") has no index: use the current device
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        if chunk is None:
            chunk = int(os.environ.get("NSR_CHUNK", "0") or 0)
        if not 0 <= int(chunk) <= 256:
            raise ValueError("chunk must be in 0..256")
        schedule_given = schedule is not None
        if schedule is None:
            schedule = os.environ.get("NSR_SCHEDULE", DEFAULT_SCHEDULE)
        if schedule not in ("queue", "phases"):
            raise ValueError("schedule must be 'queue' or 'phases'")
        if mlp is None:
            # `variant` 16 / 32 and `schedule` name fp32 forward kernels: a caller who asks for one of them gets it
            mlp = "fp32" if (variant in (16, 32) or schedule_given) else os.environ.get("NSR_MLP", DEFAULT_MLP)
        if mlp not in MLP_MODES:
            raise ValueError("mlp must be one of %s" % (MLP_MODES,))
        self.mlp = mlp
        # the x16 coarse+fine kernels only (the bf16x3 / f16x2 kernels take their items from the per-item queue)
        phases = schedule == "phases" and variant != 32 and n_importance > 0 and mlp == "fp32"
        self.schedule = "phases" if phases else "queue"
        if n_importance not in IMPORTANCE_COUNTS or (n_importance == 96 and mlp != "f16x2"):
            raise NotImplementedError("N_importance must be 128, 0 (coarse only), a divisor of 128 (rendered with duplicated "
                                      "importance samples, see _host_tables) or -- f16x2 handles -- 96; got %r" % (n_importance,))
        n_samples = int(n_samples)
        if n_samples != N_SAMPLES and not (mlp == "f16x2" and (n_samples, n_importance) in NATIVE_COUNTS):
            raise NotImplementedError("N_samples = %r with N_importance = %r: the kernels serve N_samples = 64 with every supported "
                                      "N_importance and, on f16x2 handles, (N_samples, N_importance) in %s"
                                      % (n_samples, n_importance, NATIVE_COUNTS))
        self.n_samples = n_samples
        if mlp == "f16x2" and range_fallback == "fp32" and (n_importance not in (0, N_IMPORTANCE) or n_samples != N_SAMPLES):
            # the fp32 kernels exist for (64, 128) and (64, 0) only; without the bf16x3 images such a handle would
            # have NO fallback and drop every out-of-range ray to NaN
            raise NotImplementedError("range_fallback='fp32' serves N_samples = 64 with N_importance = 128 or 0 only (got %r, %r): "
                                      "use the default 'bf16x3'" % (n_samples, n_importance))
        if n_importance > 0 and sd_fine is None:
            sd_fine = sd_coarse          # RN:482: run_fn = network_fn if network_fine is None
        self.n_importance = n_importance
        # importance samples per ray of the KERNELS this handle runs: n_importance itself where they are specialised to it
        # (f16x2, 64 / 32: 0.75x the time of 128), else 128 with duplicated uniforms (_host_tables); 0 = coarse only
        native = mlp == "f16x2" and (n_importance in NATIVE_IMPORTANCE or n_samples != N_SAMPLES)
        self.ni_kernel = n_importance if (native or n_importance == 0) else N_IMPORTANCE
        self.nf_kernel = n_samples + self.ni_kernel
        if variant not in (0, 16, 32):
            raise NotImplementedError("variant must be 0 (library default), 16 or 32")
        self.variant = variant
        self.white_bkgd, self.lindisp = bool(white_bkgd), bool(lindisp)
        cfg = _lib.NsrConfig(_lib.ABI_VERSION, self.device.index, n_samples, self.ni_kernel, max_workgroups, variant,
                             (1 if white_bkgd else 0) | (2 if lindisp else 0) | (4 if phases else 0)
                             | (8 if mlp == "bf16x3" else 0) | (16 if mlp == "f16x2" else 0), int(chunk))
        self._bbox_reserved = (0, 0)
        self._bwd_ready = self._bwd32_ready = False
        self.rays_launched = 0                       # rays handed to the render / input-gradient launches so far (host count)
        self.h2_range = None                         # f16x2: pack.h2_report of (coarse, fine) -- pack-time range report
        h = C.c_void_p()
        _lib.check(self.lib.nsr_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self.upload(sd_coarse, sd_fine)
        t, u = _host_tables(n_importance, native and n_importance not in (0, N_IMPORTANCE), n_samples)
        _lib.check(self.lib.nsr_upload_tables(self.h, _fptr(t), n_samples, _fptr(u), 128))

    def upload(self, sd_coarse, sd_fine=None):
        to_np = lambda sd: {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                            for k, v in sd.items()}
        sd_c = to_np(sd_coarse)
        p = pack_network(sd_c)
        _lib.check(self.lib.nsr_upload_weights(self.h, 0, _fptr(p), PACKED_FLOATS))
        # only the images this handle's kernels read: the x32 fp32 image always (stage kernels; an fp32 handle's per-ray
        # extras), the x16 image for fp32 handles of variant 0 / 16 only, the bf16x3 image for bf16x3 handles and -- as the
        # fallback of the range safety net -- for f16x2 handles
        x16 = self.variant != 32 and self.mlp == "fp32"
        if x16:
            p = pack_network16(sd_c)
            _lib.check(self.lib.nsr_upload_weights16(self.h, 0, _fptr(p), PACKED_FLOATS))
        fb3 = self.mlp == "bf16x3" or (self.mlp == "f16x2" and self.range_fallback == "bf16x3")
        if fb3:                                      # (f16x2: the images of the range safety net's bf16x3 fallback)
            p = pack_network_b3(sd_c)
            _lib.check(self.lib.nsr_upload_weights_b3(self.h, 0, _fptr(p), PACKED_B3_FLOATS))
        if self.mlp == "f16x2":
            p = pack_network_h2(sd_c)
            _lib.check(self.lib.nsr_upload_weights_h2(self.h, 0, _fptr(p), PACKED_FLOATS))
            self.h2_range = (h2_report(sd_c), None)
        self._sd_fine_np = None
        self._bwd_ready = self._bwd32_ready = False     # both transposed streams belong to the previous fine network
        if sd_fine is not None:
            self._sd_fine_np = to_np(sd_fine)
            p = pack_network(self._sd_fine_np)
            _lib.check(self.lib.nsr_upload_weights(self.h, 1, _fptr(p), PACKED_FLOATS))
            if x16:
                p = pack_network16(self._sd_fine_np)
                _lib.check(self.lib.nsr_upload_weights16(self.h, 1, _fptr(p), PACKED_FLOATS))
            if fb3:
                p = pack_network_b3(self._sd_fine_np)
                _lib.check(self.lib.nsr_upload_weights_b3(self.h, 1, _fptr(p), PACKED_B3_FLOATS))
            if self.mlp == "f16x2":
                p = pack_network_h2(self._sd_fine_np)
                _lib.check(self.lib.nsr_upload_weights_h2(self.h, 1, _fptr(p), PACKED_FLOATS))
                self.h2_range = (self.h2_range[0], h2_report(self._sd_fine_np))

    def close(self):
        if getattr(self, "h", None):
            self.lib.nsr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers ------------------------------------------------------------------------------------
    def _f32(self, x, shape=None):
        t = torch.as_tensor(x, dtype=torch.float32, device=self.device).contiguous()
        if shape is not None:
            t = t.reshape(shape)
        return t

    def _new(self, *shape, dtype=torch.float32):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def _outs(self, n, debug):
        fine = self.n_importance > 0
        o = dict(rgb_map=self._new(n, 3), disp_map=self._new(n), acc_map=self._new(n))
        if fine:
            o.update(rgb0=self._new(n, 3), disp0=self._new(n), acc0=self._new(n), z_std=self._new(n))
        ro = _lib.NsrRenderOut(_dev(o["rgb_map"]), _dev(o["disp_map"]), _dev(o["acc_map"]), _dev(o.get("rgb0")),
                               _dev(o.get("disp0")), _dev(o.get("acc0")), _dev(o.get("z_std")))
        dbg = None
        if debug:
            d = dict(weights0=self._new(n, self.n_samples), raw0=self._new(n, self.n_samples, 4))
            if fine:
                ni, nf = self.ni_kernel, self.nf_kernel
                d.update(z_samples=self._new(n, ni), inds=self._new(n, ni, dtype=torch.int64),
                         z_fine=self._new(n, nf), raw=self._new(n, nf, 4))
            o.update(d)
            dbg = _lib.NsrDebugOut(_dev(d["weights0"]), _dev(d.get("z_samples")), _dev(d.get("inds")),
                                   _dev(d.get("z_fine")), _dev(d["raw0"]), _dev(d.get("raw")))
        return o, ro, dbg

    # ---- the path -----------------------------------------------------------------------------------
    EXTRA_WIDTHS = dict(viewdirs=3, t_rand=64, u=128, noise0=64, noise1=192, near=1, far=1)

    def _extras(self, extras, n):
        """dict with any of viewdirs [N,3], t_rand [N,64], u [N,128], noise0 [N,64], noise1 [N,192], near + far [N]
        (include/nsr.h: NsrRayExtras) -> (struct, tensors kept alive) or (None, None)."""
        if not extras or all(v is None for v in extras.values()):
            return None, None
        unknown = set(extras) - set(self.EXTRA_WIDTHS)
        if unknown:
            raise ValueError("unknown ray extras: %s" % sorted(unknown))
        widths = dict(self.EXTRA_WIDTHS, noise1=self.nf_kernel, t_rand=self.n_samples, noise0=self.n_samples)
        # (u keeps its row stride of 128: the first ni_kernel are read)
        keep = {k: self._f32(v, (n, widths[k])) for k, v in extras.items() if v is not None}
        if ("near" in keep) != ("far" in keep):
            raise ValueError("per-ray bounds: near and far come together")
        ex = _lib.NsrRayExtras(*[_dev(keep.get(k)) for k in ("viewdirs", "t_rand", "u", "noise0", "noise1", "near", "far")])
        return ex, keep

    def _count(self, n):
        """host bookkeeping of a launch of n rays (the launch calls grow the f16x2 safety net's list themselves)"""
        self.rays_launched += n

    def reserve_range(self, n_rays):
        """nsr_reserve_range: room in the f16x2 safety net's list for launches of up to n_rays rays.  Eager launches make it
        themselves; needed only before CAPTURING (torch.cuda.graph) a launch larger than any this handle has made."""
        _lib.check(self.lib.nsr_reserve_range(self.h, int(n_rays)))

    def render_rays(self, rays_o, rays_d, near, far, debug=False, extras=None):
        """render(rays=...) (RN:58-123): rays_o, rays_d [N,3] -> dict of [N,...] tensors on the device.
        extras: the per-ray inputs of the stochastic options / given view directions (see _extras); the draws are the
        caller's."""
        rays_o = self._f32(rays_o, (-1, 3))
        rays_d = self._f32(rays_d, (-1, 3))
        n = rays_o.shape[0]
        self._count(n)
        o, ro, dbg = self._outs(n, debug)
        ex, keep = self._extras(extras, n)
        _lib.check(self.lib.nsr_render_rays_ex(self.h, _dev(rays_o), _dev(rays_d), n, float(near), float(far),
                                               C.byref(ex) if ex else None, C.byref(ro), C.byref(dbg) if dbg else None,
                                               _stream_ptr(self.device)))
        return o

    def ndc_rays(self, rays_o, rays_d, H, W, focal, near=1.0):
        """ndc_rays (RH:168-186) on the device: [..,3] x 2 -> [..,3] x 2."""
        shape = tuple(torch.as_tensor(rays_o).shape)
        ro, rd = self._f32(rays_o, (-1, 3)), self._f32(rays_d, (-1, 3))
        n = ro.shape[0]
        o, d = self._new(n, 3), self._new(n, 3)
        _lib.check(self.lib.nsr_ndc_rays(self.h, _dev(ro), _dev(rd), n, int(H), int(W), float(focal), float(near), _dev(o),
                                         _dev(d), _stream_ptr(self.device)))
        return o.reshape(shape), d.reshape(shape)

    def ndc_rays_vjp(self, rays_o, rays_d, H, W, focal, grad_o_ndc, grad_d_ndc, near=1.0):
        """(dL/d o', dL/d d') of ndc_rays -> (dL/d rays_o, dL/d rays_d), [N,3] each."""
        ro, rd = self._f32(rays_o, (-1, 3)), self._f32(rays_d, (-1, 3))
        n = ro.shape[0]
        g0, g1 = self._f32(grad_o_ndc, (n, 3)), self._f32(grad_d_ndc, (n, 3))
        go, gd = self._new(n, 3), self._new(n, 3)
        _lib.check(self.lib.nsr_ndc_rays_vjp(self.h, _dev(ro), _dev(rd), n, int(H), int(W), float(focal), float(near),
                                             _dev(g0), _dev(g1), _dev(go), _dev(gd), _stream_ptr(self.device)))
        return go, gd

    def render_views(self, c2w, H, W, K, near, far, debug=False):
        """render(c2w=...) for V views in one launch: c2w [V,3,4] (or [3,4]) -> dict of [V*H*W,...] tensors."""
        c2w = self._f32(c2w)
        if c2w.dim() == 2:
            c2w = c2w[None]
        c2w = c2w[:, :3, :4].contiguous()
        v = c2w.shape[0]
        n = v * int(H) * int(W)
        self._count(n)
        K9 = (C.c_double * 9)(*[float(K[i][j]) for i in range(3) for j in range(3)])
        o, ro, dbg = self._outs(n, debug)
        _lib.check(self.lib.nsr_render_views(self.h, _dev(c2w), v, int(H), int(W), K9, float(near), float(far),
                                             C.byref(ro), C.byref(dbg) if dbg else None, _stream_ptr(self.device)))
        return o

    def render_rays_vjp(self, rays_o, rays_d, near, far, grad_rgb, with_forward=False, z_fine=None, extras=None,
                        debug=False):
        """Forward + input-side VJP (RN:168-178): grad_rgb [N,3] -> (grad_rays_o, grad_rays_d) [N,3] each.
        z_fine (optional [N,192]): sorted fine sample depths to differentiate at, instead of the kernel's own
        resampling (they are constants of the backward, RN:475).
        extras: as for render_rays; with extras["viewdirs"] the view directions are an input of their own and the result
        gains dL/d viewdirs: (grad_o, grad_d, grad_viewdirs[, forward]).
        debug: append a dict with the taps of include/nsr.h: NsrVjpDebugOut (relu_masks uint32 [ceil(N/2),3,9,256,4],
        grad_raw [N,192,4], grad_pts [N,192,6]) -- served by the x32-structured kernels."""
        if self.n_importance == 0:
            raise NotImplementedError("the VJP kernel needs the coarse+fine configuration (N_importance > 0)")
        has_extras = bool(extras) and any(v is not None for v in extras.values())
        need32 = (has_extras or debug) and self.mlp == "fp32" and self.variant != 32
        if need32 and not self._bwd32_ready:
            # the fp32 x32 transposed stream: the extras and the debug taps are served by the x32-structured kernels (an fp32
            # handle of another variant runs k_render_vjp for them)
            b = pack_network_backward(self._sd_fine_np)
            _lib.check(self.lib.nsr_upload_weights_bwd(self.h, _fptr(b), b.size))
            self._bwd32_ready = True
        if not self._bwd_ready:                      # the transposed stream is packed on first use only
            if self.mlp == "bf16x3":
                b = pack_network_backward_b3(self._sd_fine_np)
                _lib.check(self.lib.nsr_upload_weights_bwd_b3(self.h, _fptr(b), b.size))
            elif self.mlp == "f16x2":
                b = pack_network_backward_h2(self._sd_fine_np)
                _lib.check(self.lib.nsr_upload_weights_bwd_h2(self.h, _fptr(b), b.size))
                if self.range_fallback == "bf16x3":                     # ... and the range safety net's fallback stream
                    b = pack_network_backward_b3(self._sd_fine_np)
                    _lib.check(self.lib.nsr_upload_weights_bwd_b3(self.h, _fptr(b), b.size))
                else:
                    b = pack_network_backward(self._sd_fine_np)
                    _lib.check(self.lib.nsr_upload_weights_bwd(self.h, _fptr(b), b.size))
            elif self.variant == 32:
                b = pack_network_backward(self._sd_fine_np)
                _lib.check(self.lib.nsr_upload_weights_bwd(self.h, _fptr(b), b.size))
            else:                                    # 0 = library default = x16
                b = pack_network_backward16(self._sd_fine_np)
                _lib.check(self.lib.nsr_upload_weights_bwd16(self.h, _fptr(b), b.size))
            self._bwd_ready = True
        rays_o = self._f32(rays_o, (-1, 3))
        rays_d = self._f32(rays_d, (-1, 3))
        n = rays_o.shape[0]
        self._count(n)
        g = self._f32(grad_rgb, (n, 3))
        zf = self._f32(z_fine, (n, self.nf_kernel)) if z_fine is not None else None
        go, gd = self._new(n, 3), self._new(n, 3)
        ro, fwd = None, None
        if with_forward:
            fwd = dict(rgb_map=self._new(n, 3), disp_map=self._new(n), acc_map=self._new(n))
            ro = _lib.NsrRenderOut(_dev(fwd["rgb_map"]), _dev(fwd["disp_map"]), _dev(fwd["acc_map"]), None, None,
                                   None, None)
        ex, keep = self._extras(extras, n)
        gv = self._new(n, 3) if (keep and "viewdirs" in keep) else None
        taps, dbg = None, None
        if debug:
            nf = self.nf_kernel
            taps = dict(relu_masks=torch.zeros(((n + 1) // 2, (2 * nf + 127) // 128, 9, 256, 4), dtype=torch.int32, device=self.device),
                        grad_raw=self._new(n, nf, 4), grad_pts=self._new(n, nf, 6))
            dbg = _lib.NsrVjpDebugOut(_dev(taps["relu_masks"]), _dev(taps["grad_raw"]), _dev(taps["grad_pts"]))
        _lib.check(self.lib.nsr_render_rays_vjp_dbg(self.h, _dev(rays_o), _dev(rays_d), n, float(near), float(far),
                                                    C.byref(ex) if ex else None, _dev(g), _dev(go), _dev(gd), _dev(gv),
                                                    _dev(zf), C.byref(ro) if ro else None, C.byref(dbg) if dbg else None,
                                                    _stream_ptr(self.device)))
        res = (go, gd) if gv is None else (go, gd, gv)
        if with_forward:
            res = res + (fwd,)
        return res + (taps,) if debug else res

    def pose_grad(self, grad_o, grad_d, H, W, K, patch):
        """dL/d c2w[3,4] per patch of `patch` consecutive pixels, given dL/d rays of a full H x W image."""
        grad_o = self._f32(grad_o, (int(H) * int(W), 3))
        grad_d = self._f32(grad_d, (int(H) * int(W), 3))
        n_patches = (int(H) * int(W) + int(patch) - 1) // int(patch)
        out = self._new(n_patches, 3, 4)
        K9 = (C.c_double * 9)(*[float(K[i][j]) for i in range(3) for j in range(3)])
        _lib.check(self.lib.nsr_pose_grad(self.h, _dev(grad_o), _dev(grad_d), int(H), int(W), K9, int(patch),
                                          _dev(out), _stream_ptr(self.device)))
        return out

    def get_rays(self, H, W, K, c2w):
        c2w = self._f32(c2w)[:3, :4].contiguous()
        K9 = (C.c_double * 9)(*[float(K[i][j]) for i in range(3) for j in range(3)])
        o, d = self._new(H, W, 3), self._new(H, W, 3)
        _lib.check(self.lib.nsr_get_rays(self.h, _dev(c2w), int(H), int(W), K9, _dev(o), _dev(d),
                                         _stream_ptr(self.device)))
        return o, d

    def get_rays_views(self, H, W, K, c2w):
        """get_rays for V cameras c2w [V, >=3, 4] in one launch: (rays_o, rays_d) [V*H*W, 3] each, view-major."""
        c2w = self._f32(c2w)
        if c2w.dim() == 2:
            c2w = c2w[None]
        c2w = c2w[:, :3, :4].contiguous()
        V = int(c2w.shape[0])
        K9 = (C.c_double * 9)(*[float(K[i][j]) for i in range(3) for j in range(3)])
        o, d = self._new(V * H * W, 3), self._new(V * H * W, 3)
        _lib.check(self.lib.nsr_get_rays_views(self.h, _dev(c2w), V, int(H), int(W), K9, _dev(o), _dev(d),
                                               _stream_ptr(self.device)))
        return o, d

    def sample_pose(self, prob, gumbel, uniform, theta, gumbel_T, radius=1.01, want_jac=True):
        """sample_pose (LL:202-247) on the device in torch's fp32 arithmetic: prob [n_cat] (fp32), recorded noise
        (sample_log lists / arrays, fp64) -> (poses [K,4,4], jac [K,12,n_cat] = d c2w[:3,:4] / d prob or None)."""
        prob = self._f32(prob).reshape(-1)
        f64 = lambda x, shape: torch.as_tensor(np.asarray(x, dtype=np.float64), device=self.device).reshape(shape).contiguous()
        n_cat = prob.numel()
        th = f64(theta, (-1,))
        K = th.numel()
        g, u = f64(gumbel, (K, n_cat)), f64(uniform, (K,))
        poses = self._new(K, 4, 4)
        jac = self._new(K, 12, n_cat) if want_jac else None
        _lib.check(self.lib.nsr_sample_pose(self.h, _dev(prob), _dev(g), _dev(u), _dev(th), K, n_cat, float(gumbel_T),
                                            float(radius), _dev(poses), None, _dev(jac), _stream_ptr(self.device)))
        return poses, jac

    def sample_pose_nograd(self, logits64, gumbel, uniform, theta, gumbel_T, radius=1.01):
        """sample_pose_nograd's deterministic part (LL:275-293) on the device in numpy's fp64 arithmetic."""
        f64 = lambda x, shape: torch.as_tensor(np.asarray(x, dtype=np.float64), device=self.device).reshape(shape).contiguous()
        lg = f64(logits64, (-1,))
        th = f64(theta, (-1,))
        K, n_cat = th.numel(), lg.numel()
        g, u = f64(gumbel, (K, n_cat)), f64(uniform, (K,))
        poses = self._new(K, 4, 4)
        _lib.check(self.lib.nsr_sample_pose_nograd(self.h, _dev(lg), _dev(g), _dev(u), _dev(th), K, n_cat,
                                                   float(gumbel_T), float(radius), _dev(poses), None,
                                                   _stream_ptr(self.device)))
        return poses

    def to8b(self, x):
        """to8b (RH:14) on device: float tensor -> uint8 tensor of the same shape."""
        x = self._f32(x)
        out = torch.empty(x.shape, dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.nsr_to8b(self.h, _dev(x), x.numel(), _dev(out), _stream_ptr(self.device)))
        return out

    def fingerprint(self, ptr_table, n_words, out):
        """64-bit content hash of the device buffers listed in ptr_table (int64 device tensor of addresses) with n_words
        (int64 device tensor of 32-bit word counts) into `out` (int64 device tensor, 1 element); enqueue only."""
        _lib.check(self.lib.nsr_fingerprint(self.h, _dev(ptr_table), _dev(n_words), int(ptr_table.numel()), _dev(out),
                                            _stream_ptr(self.device)))
        return out

    def find_bbox(self, rgb8, with_mask=False):
        """get_annotation / find_bbox (NM:786-797) on device: uint8 [K,H,W,3] RGB -> (bbox [K,4] int32 XYWH,
        count [K] int32[, mask [K,H,W] uint8])."""
        rgb8 = torch.as_tensor(rgb8, dtype=torch.uint8, device=self.device).contiguous()
        if rgb8.dim() == 3:
            rgb8 = rgb8[None]
        k, hh, ww, _ = rgb8.shape
        if hh * ww > self._bbox_reserved[0] * self._bbox_reserved[1]:       # setup, once per image size
            _lib.check(self.lib.nsr_reserve_bbox(self.h, hh, ww))
            self._bbox_reserved = (hh, ww)
        bbox = torch.empty((k, 4), dtype=torch.int32, device=self.device)
        count = torch.empty((k,), dtype=torch.int32, device=self.device)
        mask = torch.empty((k, hh, ww), dtype=torch.uint8, device=self.device) if with_mask else None
        _lib.check(self.lib.nsr_find_bbox(self.h, _dev(rgb8), k, hh, ww, _dev(bbox), _dev(count),
                                          _dev(mask) if with_mask else None, _stream_ptr(self.device)))
        return (bbox, count, mask) if with_mask else (bbox, count)

    def embed(self, x, multires):
        """Embedder.embed (RH:39-48): [..., 3] -> [..., 3 + 6*multires]."""
        x = self._f32(x)
        lead = tuple(x.shape[:-1])
        flat = x.reshape(-1, 3).contiguous()
        out = self._new(flat.shape[0], 3 + 6 * int(multires))
        _lib.check(self.lib.nsr_embed(self.h, _dev(flat), flat.shape[0], int(multires), _dev(out),
                                      _stream_ptr(self.device)))
        return out.reshape(*lead, 3 + 6 * int(multires))

    def run_network(self, pts, viewdirs, net_id=0):
        pts = self._f32(pts, (-1, 3))
        viewdirs = self._f32(viewdirs, (-1, 3))
        raw = self._new(pts.shape[0], 4)
        _lib.check(self.lib.nsr_run_network(self.h, int(net_id), _dev(pts), _dev(viewdirs), pts.shape[0],
                                            _dev(raw), _stream_ptr(self.device)))
        return raw

    def raw2outputs(self, raw, z_vals, rays_d):
        raw = self._f32(raw)
        n, s = raw.shape[0], raw.shape[1]
        z_vals = self._f32(z_vals, (n, s))
        rays_d = self._f32(rays_d, (n, 3))
        rgb, disp, acc, w, depth = self._new(n, 3), self._new(n), self._new(n), self._new(n, s), self._new(n)
        _lib.check(self.lib.nsr_raw2outputs(self.h, _dev(raw), _dev(z_vals), _dev(rays_d), n, s, _dev(rgb),
                                            _dev(disp), _dev(acc), _dev(w), _dev(depth), _stream_ptr(self.device)))
        return rgb, disp, acc, w, depth

    def _stage_tables_ok(self, what):
        if self.ni_kernel not in (0, N_IMPORTANCE) or self.n_samples != N_SAMPLES:      # (a native handle's u table is linspace(0,1,n) + padding)
            raise NotImplementedError("%s is specialised to 128 importance samples and reads the handle's uniforms table; this "
                                      "handle's kernels (and table) are specialised to N_importance = %d -- use a 128 handle"
                                      % (what, self.ni_kernel))

    def sample_pdf(self, bins, weights):
        self._stage_tables_ok("sample_pdf")
        bins = self._f32(bins)
        n = bins.shape[0]
        if bins.shape[1] != 63 or tuple(weights.shape) != (n, 62):
            raise NotImplementedError("sample_pdf is specialised to 63 bins / 62 weights / 128 samples")
        weights = self._f32(weights)
        samples, inds = self._new(n, 128), self._new(n, 128, dtype=torch.int64)
        _lib.check(self.lib.nsr_sample_pdf(self.h, _dev(bins), _dev(weights), n, _dev(samples), _dev(inds),
                                           _stream_ptr(self.device)))
        return samples, inds

    def sort_merge(self, z_coarse, z_samples):
        """torch.sort(torch.cat([z_vals, z_samples], -1), -1) values (RN:477): [N,64] + [N,128] -> [N,192]."""
        z_coarse = self._f32(z_coarse)
        n = z_coarse.shape[0]
        z_samples = self._f32(z_samples, (n, 128))
        if z_coarse.shape[1] != 64:
            raise NotImplementedError("sort_merge is specialised to 64 + 128 samples")
        out = self._new(n, 192)
        _lib.check(self.lib.nsr_sort_merge(self.h, _dev(z_coarse), _dev(z_samples), n, _dev(out),
                                           _stream_ptr(self.device)))
        return out

    def selftest(self):
        _lib.check(self.lib.nsr_selftest(self.h, _stream_ptr(self.device)))

    def schedule_stats(self):
        """Rays (cumulative) whose fine task recomputed its coarse pass instead of using the handed-over depths
        (global-phases schedule; 0 in normal operation and for the per-ray queue)."""
        n = C.c_uint()
        _lib.check(self.lib.nsr_schedule_stats(self.h, C.byref(n)))
        return int(n.value)

    def range_status(self):
        """f16x2 range safety net (include/nsr.h: nsr_range_status): dict(last_items, points, rays, dropped_items) --
        items (2 rays) the last launch handed to its fp32 fallback; cumulative network evaluations with NaN outputs /
        gradients, rays re-rendered by the fp32 kernel, items that could not be (their rays hold NaN; only a launch captured
        into a graph before reserve_range gets there).  Synchronises the device."""
        v = [C.c_uint() for _ in range(4)]
        _lib.check(self.lib.nsr_range_status(self.h, *[C.byref(x) for x in v]))
        return dict(zip(("last_items", "points", "rays", "dropped_items"), (int(x.value) for x in v)))

    def debug_bounds_status(self):
        """(built_with_checks, first_bad_source_line): see nsr_debug_bounds_status / `make debug`."""
        built, line = C.c_int(), C.c_uint()
        _lib.check(self.lib.nsr_debug_bounds_status(self.h, C.byref(built), C.byref(line)))
        return bool(built.value), int(line.value)

    def last_kernel_ms(self):
        ms = C.c_float()
        _lib.check(self.lib.nsr_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value
