"""Drop-in for the reference's utils/run_nerf_noscale.py (RN): `create_nerf`, `render`, `render_path`,
`render_path_grad`, `to8b`, `device` with the reference's signatures and return conventions, so that
optimization/neural_sim_main.py (NM:35 star-import, call sites NM:67, NM:128, NM:184) runs unchanged on top.

Everything below L1 of SURVEY.md section 1 -- batchify_rays, render_rays, run_network, raw2outputs, sample_pdf,
Embedder, the 8x256 MLP -- is ONE persistent gfx950 kernel (csrc/nsr_kernels.hip) reached through the C ABI
of include/nsr.h.  This file is host glue: argument checking, handle caching, reshapes, PNG side effects.

Which kernels run a call (`_model_for`): a NeRF that can be written as the kernels' 8x256 network (run_nerf_helpers.fits_kernel;
smaller networks and use_viewdirs=False networks, RH:95-96, through NeRF.native_state_dict) at sample counts a fused kernel is
built for (N_samples 64 with N_importance 0 / 128 / 96 / a divisor of 128; (32, 64), (128, 128)) -> the fused kernels
(engine.NsrModel); every other network or pair of counts -> the layered renderer (wide.WideModel, include/nsr_wide.h: one
fp32-MFMA GEMM per layer, activations in HBM).  What neither serves raises NotImplementedError (the reference has no error
convention; silently taking a different path is worse).  white_bkgd (RN:384-385) and lindisp (RN:443) are handle flags (one
native handle per option pair); ndc=True (RN:101-103), c2w_staticcam (RN:91-96), perturb>0 (RN:447-459, RH:211) and
raw_noise_std>0 (RN:365-374) reach the kernels as per-ray extras (include/nsr.h: NsrRayExtras, include/nsr_wide.h: NsrwExtras)
-- see _draws for where the random numbers come from.

`create_nerf` deliberately restates RN:258-340 statement by statement: the args it reads, the kwargs keys, the
checkpoint keys and the returned 5-tuple ARE the drop-in contract, so there is nothing to redesign there."""
import os
import time

import numpy as np
import torch

from .run_nerf_helpers import NeRF, get_embedder, to8b, img2mse, mse2psnr, get_rays, sample_pdf  # noqa: F401
from . import png

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")      # RN:9
DEBUG = False


# ------------------------------------------------------------------------------------------------------
# native handles
# ------------------------------------------------------------------------------------------------------
_UTIL = {}
_MAX_HANDLES_PER_MODULE = 8        # (render options, device, stream) handles kept per network pair, least recently used out


def _native_sd(net):
    """State dict in the architecture the kernels are built for (a use_viewdirs=False module maps itself onto it)."""
    return net.native_state_dict() if hasattr(net, "native_state_dict") else net.state_dict()


def _util_model(dev=None):
    """A weight-less handle for the stage kernels that need no network (get_rays, sample_pdf)."""
    from .engine import NsrModel
    from .synthetic import synth_weights
    idx = torch.cuda.current_device() if dev is None or dev.index is None else dev.index
    if idx not in _UTIL:
        _UTIL[idx] = NsrModel(synth_weights(0), None, device=idx, n_importance=0, mlp="fp32")   # stage kernels only
    return _UTIL[idx]


def _model_for(network_fn, network_fine, n_importance, kw=None, trust=False):
    """One native handle per (network_fn, network_fine, render options) tuple, repacked when the parameters change.
    `kw`: the render kwargs, read for white_bkgd / lindisp (RN:384-385, RN:443).
    The cache is keyed on the parameters' storage identity + autograd versions; their CONTENT fingerprint (run_nerf_helpers:
    NSR_TRUST_VERSIONS) is compared on top of that by every call that takes it -- trust=True skips it: render() passes that
    for its 512-ray patch form, whose fingerprint it reads AFTER the launch (deferred check), or not at all when
    NSR_TRUST_VERSIONS says so."""
    from .engine import NsrModel
    network_fn, network_fine = NeRF.adopt(network_fn), NeRF.adopt(network_fine)     # (a module with the reference's layout: shared weights)
    white, lindisp = bool((kw or {}).get("white_bkgd", False)), bool((kw or {}).get("lindisp", False))
    n_samples = int((kw or {}).get("N_samples", 64))
    ident, fp = NeRF.weights_version_of(network_fn, network_fine, trust=trust)
    forced = network_fn.__dict__.get("_nsr_force_mlp")     # set by _note_range: THESE weights keep leaving f16x2's range
    forced = forced[0] if forced and forced[1] == ident else None
    from .engine import NATIVE_COUNTS
    forced_any = forced
    if forced and ((n_samples, n_importance) in NATIVE_COUNTS or n_importance == 96):
        forced = None                 # sample counts only the f16x2 kernels are specialised to: stay there (per-item fallback)
    layered = _layered_why(network_fn, network_fine, n_samples, n_importance, forced or os.environ.get("NSR_MLP"),
                           retraw=bool((kw or {}).get("retraw", False)))
    if layered:
        forced = forced_any           # (the layered renderer has every arithmetic at every sample count)
    key = (n_samples, n_importance, ident, ("layered", forced) if layered else (forced or os.environ.get("NSR_MLP")))     # which kernels
    # a native handle owns ONE argument block / work queue / scratch set (include/nsr.h: one handle per (model,
    # stream)), so the cache is also keyed on the device and on the torch stream the launch will be issued on
    p0 = next(network_fn.parameters())
    dev = p0.device.index if p0.is_cuda else torch.cuda.current_device()
    stream_id = torch.cuda.current_stream(dev).cuda_stream
    pair = network_fn.__dict__.setdefault("_nsr_pair", {})
    slot = (white, lindisp, dev, stream_id)
    cache = pair.pop(slot, None) or {}
    pair[slot] = cache                                    # most recently used last
    while len(pair) > _MAX_HANDLES_PER_MODULE:            # each handle holds 3 x 2 packed networks + ~19 MB of scratch
        # dropped, not closed: an autograd graph of an earlier render() may still hold the handle (ctx.cfg of _RenderRays /
        # _RenderRaysEx) for its backward; NsrModel.__del__ frees the native side with the last reference
        del pair[next(iter(pair))]
    stale = fp != () and cache.get("fp", fp) != fp        # same storage, same versions, other bits: a write through .data
    if cache.get("key") != key or stale:
        cache.pop("model", None)                          # same: released when no graph refers to it any more
        if fp == ():                                      # (a trusted call that has to pack takes the fingerprint once, so
            fp = NeRF.weights_version_of(network_fn, network_fine)[1]      # that later checking calls have one to compare with)
        with torch.cuda.device(dev):
            if layered:                                   # any network, any sample counts: include/nsr_wide.h
                from .wide import WideModel
                own = lambda net: {k: v.detach() for k, v in net.state_dict().items()}
                cache["model"] = WideModel(own(network_fn), own(network_fine) if network_fine is not None else None, device=dev,
                                           n_importance=n_importance, white_bkgd=white, lindisp=lindisp, n_samples=n_samples,
                                           mlp=forced)
                cache["model"].why_layered = layered
            else:
                cache["model"] = NsrModel(_native_sd(network_fn), _native_sd(network_fine) if network_fine is not None
                                          else None, device=dev, n_importance=n_importance, white_bkgd=white,
                                          lindisp=lindisp, mlp=forced, n_samples=n_samples)
            cache["model"].weights_version = ident
        cache["key"] = key
        cache["fp"] = fp
    elif fp != ():
        cache["fp"] = fp
    cache["model"].weights_fp = cache.get("fp", ())       # what render()'s deferred check of the patch form compares with
    return cache["model"]


def _layered_why(network_fn, network_fine, n_samples, n_importance, mlp=None, retraw=False):
    """None when the fused kernels serve this (networks, sample counts) pair, else the reason the layered renderer
    (wide.WideModel: one MFMA GEMM launch per layer, activations in HBM) takes it: a network that is not expressible as the
    kernels' 8 x 256 (NeRF.fused_why_not), or sample counts no fused kernel is built for.  NSR_LAYERED=1 sends everything
    there (cross-checks of the two renderers against each other)."""
    from .engine import IMPORTANCE_COUNTS, NATIVE_COUNTS, DEFAULT_MLP
    if os.environ.get("NSR_LAYERED") == "1":
        return "NSR_LAYERED=1"
    for net in (network_fn, network_fine):
        if net is not None and net.fused_why_not:
            return net.fused_why_not
    f16x2 = (mlp or DEFAULT_MLP) == "f16x2"
    if retraw:
        # RN:490-491: raw is [N, N_samples + N_importance, C].  The fused kernels return that where they evaluate exactly those
        # samples and the network has four output rows; elsewhere (duplicated importance samples, use_viewdirs=False with
        # output_ch != 4) the layered renderer does (r06; r05 refused)
        from .engine import NATIVE_IMPORTANCE
        last = network_fine if (n_importance > 0 and network_fine is not None) else network_fn
        if not last.use_viewdirs and last.output_ch != 4:
            return "retraw of a use_viewdirs=False network with output_ch=%d" % last.output_ch
        if n_samples == 64 and n_importance in IMPORTANCE_COUNTS and n_importance not in (0, 128) and \
                not (f16x2 and (n_importance in NATIVE_IMPORTANCE or n_importance == 96)):
            return "retraw with N_importance=%d (the %s kernels render it with duplicated samples)" % (n_importance, mlp or DEFAULT_MLP)
    if n_samples == 64 and n_importance in IMPORTANCE_COUNTS and (n_importance != 96 or f16x2):
        return None
    if f16x2 and (n_samples, n_importance) in NATIVE_COUNTS:
        return None
    return "N_samples=%r with N_importance=%r" % (n_samples, n_importance)


_RANGE_SWITCH_FRAC = 0.10         # more than this share of a handle's rays re-rendered by the fallback kernel: use bf16x3 outright
_RANGE_SWITCH_MLP = "bf16x3"      # ... the arithmetic with fp32's exponent range (no failure domain) at 1.7x the fp32-MFMA speed


def _note_range(model, network_fn=None):
    """f16x2 range safety net (include/nsr.h: NSR_FLAG_MLP_F16X2): called where the API has synchronised anyway.  Rays whose
    network evaluation left the fp16 range were rendered again by the bf16x3 kernel inside the same launch call -- the
    results are that kernel's (fp32-grade, fp32's exponent range) -- so this only says so, once per handle.  A network that
    sends more than a tenth of its rays down that route pays for both kernels: its later handles are built with the bf16x3
    kernels (`_model_for` reads the mark), which is the same arithmetic without the detour -- 157 against 94 Mray-samples/s
    for the fp32-MFMA kernels r04 switched to."""
    if network_fn is not None and not isinstance(network_fn, NeRF):
        network_fn = NeRF.adopt(network_fn)        # (the mark below lives on the wrapper _model_for reads it from)
    if getattr(model, "mlp", None) == "layered-f16x2":
        # the layered renderer's unit is a network PASS over a chunk of rays, re-run on bf16x3 as a whole (include/nsr_wide.h)
        st = model.range_status()
        if st["passes_rerun"] and not getattr(model, "_range_warned", False):
            import warnings
            model._range_warned = True
            switch = network_fn is not None and st["passes_rerun"] > _RANGE_SWITCH_FRAC * max(1, st["passes"])
            if switch:
                network_fn.__dict__["_nsr_force_mlp"] = (_RANGE_SWITCH_MLP, getattr(model, "weights_version", None))
            warnings.warn("neural_sim_nerf_amd: %d of %d network passes of the layered renderer left the fp16 range of its f16x2 GEMMs "
                          "and were run again on bf16x3 inside the call (the results are that arithmetic's).%s"
                          % (st["passes_rerun"], st["passes"], "  This network now gets the bf16x3 GEMMs outright." if switch else
                             "  NSR_WIDE_MLP=bf16x3 selects them outright."), RuntimeWarning)
        return
    if getattr(model, "mlp", None) != "f16x2":
        return
    st = model.range_status()
    if not st["points"]:
        return
    switch = network_fn is not None and st["rays"] > _RANGE_SWITCH_FRAC * max(1, model.rays_launched)
    if switch:
        network_fn.__dict__["_nsr_force_mlp"] = (_RANGE_SWITCH_MLP, getattr(model, "weights_version", None))
    if not getattr(model, "_range_warned", False):
        import warnings
        model._range_warned = True
        dropped = ("" if not st["dropped_items"] else
                   "; %d items (2 rays each) could NOT be -- a launch captured into a graph before NsrModel.reserve_range "
                   "-- and every output of their out-of-range rays is NaN" % st["dropped_items"])
        warnings.warn("neural_sim_nerf_amd: %d network evaluations left the fp16 range of the f16x2 kernels; %d of %d rays were "
                      "rendered again by the bf16x3 kernel%s.%s"
                      % (st["points"], st["rays"], model.rays_launched, dropped,
                         "  This network now gets the bf16x3 kernels outright." if switch else
                         "  NSR_MLP=bf16x3 selects the bf16x3 kernels outright."), RuntimeWarning)


# ------------------------------------------------------------------------------------------------------
# autograd glue
# ------------------------------------------------------------------------------------------------------
class _GetRays(torch.autograd.Function):
    """rays_d[p] = R dirs[p], rays_o[p] = t (RH:160-164): linear in c2w, so the VJP is a reduction over pixels."""

    @staticmethod
    def forward(ctx, c2w, H, W, K):
        m = _util_model(c2w.device if c2w.is_cuda else None)
        o, d = m.get_rays(H, W, K, c2w.detach())
        ctx.geom = (H, W, K, m)
        ctx.c2w_shape = c2w.shape
        return o, d

    @staticmethod
    def backward(ctx, go, gd):
        H, W, K, m = ctx.geom
        g = m.pose_grad(go.reshape(-1, 3), gd.reshape(-1, 3), H, W, K, H * W)[0]      # [3,4]
        out = torch.zeros(ctx.c2w_shape, dtype=torch.float32, device=g.device)
        out[:3, :4] = g
        return out, None, None, None


def _get_rays_autograd(H, W, K, c2w):
    c2w = torch.as_tensor(c2w, dtype=torch.float32)
    if not c2w.is_cuda:
        c2w = c2w.to(device)
    if c2w.requires_grad:
        return _GetRays.apply(c2w, int(H), int(W), K)
    return _util_model(c2w.device).get_rays(int(H), int(W), K, c2w)


class _RenderRays(torch.autograd.Function):
    """render(rays=...) with the input-side VJP the bilevel loop needs (RN:177: d rgb / d rays; network weights
    are frozen and z_samples is detached, RN:475, so nothing else carries gradient)."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, model, near, far, want_raw):
        out = model.render_rays(rays_o.detach(), rays_d.detach(), near, far, debug=want_raw)
        ctx.save_for_backward(rays_o.detach(), rays_d.detach())
        ctx.cfg = (model, near, far)
        fine = model.n_importance > 0
        keys = ["rgb_map", "disp_map", "acc_map"] + (["rgb0", "disp0", "acc0", "z_std"] if fine else [])
        if want_raw:
            keys.append("raw" if fine else "raw0")
        ctx.n_out = len(keys)
        ctx.mark_non_differentiable(*[out[k] for k in keys[1:]])
        return tuple(out[k] for k in keys)

    @staticmethod
    def backward(ctx, g_rgb, *others):
        model, near, far = ctx.cfg
        rays_o, rays_d = ctx.saved_tensors
        go, gd = model.render_rays_vjp(rays_o, rays_d, near, far, g_rgb)
        return go, gd, None, None, None, None


class _RenderRaysEx(torch.autograd.Function):
    """render_rays with the per-ray extras of the options beyond the deterministic test-time path (include/nsr.h:
    NsrRayExtras): given view directions (c2w_staticcam RN:91-96, ndc RN:101-103) -- a differentiable input of their own
    -- and the draws of perturb / raw_noise_std (constants of the backward pass)."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, viewdirs, model, near, far, want_raw, draws):
        ex = dict(draws or {})
        if viewdirs is not None:
            ex["viewdirs"] = viewdirs.detach()
        out = model.render_rays(rays_o.detach(), rays_d.detach(), near, far, debug=want_raw, extras=ex)
        ctx.save_for_backward(rays_o.detach(), rays_d.detach())
        ctx.cfg = (model, near, far, ex)
        fine = model.n_importance > 0
        keys = ["rgb_map", "disp_map", "acc_map"] + (["rgb0", "disp0", "acc0", "z_std"] if fine else [])
        if want_raw:
            keys.append("raw" if fine else "raw0")
        ctx.mark_non_differentiable(*[out[k] for k in keys[1:]])
        return tuple(out[k] for k in keys)

    @staticmethod
    def backward(ctx, g_rgb, *others):
        model, near, far, ex = ctx.cfg
        rays_o, rays_d = ctx.saved_tensors
        res = model.render_rays_vjp(rays_o, rays_d, near, far, g_rgb, extras=ex)
        gv = res[2] if "viewdirs" in ex else None
        return res[0], res[1], gv, None, None, None, None, None


class _NdcRays(torch.autograd.Function):
    """ndc_rays (RH:168-186) on the device, with its input-side VJP."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, model, H, W, focal, near):
        ctx.save_for_backward(rays_o.detach(), rays_d.detach())
        ctx.cfg = (model, H, W, focal, near)
        return model.ndc_rays(rays_o.detach(), rays_d.detach(), H, W, focal, near)

    @staticmethod
    def backward(ctx, g_o, g_d):
        model, H, W, focal, near = ctx.cfg
        rays_o, rays_d = ctx.saved_tensors
        go, gd = model.ndc_rays_vjp(rays_o, rays_d, H, W, focal, g_o.contiguous(), g_d.contiguous(), near)
        return go.reshape(rays_o.shape), gd.reshape(rays_d.shape), None, None, None, None, None


def _draws(kw, n, n_importance, dev, chunk=None, ni_kernel=128, n_samples=64, u_width=128):
    """The random draws of the stochastic options, in the reference's order: per `chunk` of rays (the reference draws inside
    render_rays, which batchify_rays calls once per chunk) t_rand (RN:451), the coarse density noise (RN:368), the
    resampling uniforms (RH:211), the fine density noise.  From torch's generator of the render device, by the same calls
    with the same shapes: a process that seeds -- or, like tests/golden/g21, replaces -- torch.rand / torch.randn gets the
    same numbers into the same rays as the reference; pinned comparisons can also hand the reference's own draws to
    NsrModel.render_rays(extras=...).
    pytest=True (the reference's test hook, RN:454-457, RH:214-222): every draw site reseeds numpy's GLOBAL generator
    with 0 and takes its numbers from it -- once per `chunk` of rays, so ray i gets row i mod chunk -- and the
    deterministic resampling uses NUMPY's linspace (a few ulp from torch's); reproduced exactly, side effect included.
    ni_kernel: importance samples per ray of the handle's kernels (NsrModel.ni_kernel): n_importance itself where they are
    specialised to it -- a row of u then holds the n draws followed by padding, the fine noise has 64 + n columns -- else 128
    with every draw repeated 128 / n times.  u_width: columns of a row of u as the handle reads it (128 for the fused
    kernels, n_importance for the layered renderer)."""
    native = n_importance > 0 and ni_kernel == n_importance and n_importance != 128

    def widen(u):               # [.., n_importance] -> the [.., 128] row the fused kernels read
        if u_width == n_importance:     # the layered renderer reads [.., n_importance] rows as they are
            return u
        if native:
            return torch.nn.functional.pad(u, (0, 128 - n_importance))
        return u.repeat_interleave(128 // n_importance, dim=-1)
    perturb = kw.get("perturb", 0.)
    perturbed = perturb not in (0, 0., False) and perturb > 0.
    std = float(kw.get("raw_noise_std", 0.) or 0.)
    d = {}
    if kw.get("pytest", False):
        if std > 0.:
            raise NotImplementedError("render: pytest=True with raw_noise_std > 0 -- the reference itself fails there (RN:371 "
                                      "calls .cuda() on a numpy array)")
        c = n if not chunk else max(1, min(int(chunk), n))
        rows = torch.arange(n) % c
        def seeded(width):
            np.random.seed(0)
            return torch.Tensor(np.random.rand(c, width))[rows]
        if perturbed:
            d["t_rand"] = seeded(n_samples).to(dev)
        if n_importance > 0:
            if perturbed:
                d["u"] = widen(seeded(n_importance)).to(dev)
            else:
                np.random.seed(0)                           # RH:216 reseeds on the deterministic branch too
                u = widen(torch.Tensor(np.linspace(0., 1., n_importance)))
                d["u"] = u[None].expand(n, u.shape[0]).contiguous().to(dev)
        return d
    # the reference draws inside render_rays, i.e. once per `chunk` of rays (batchify_rays RN:43-55), in the order t_rand,
    # coarse noise, u, fine noise: the same calls in the same order here, chunk by chunk, so that a caller who seeds (or
    # replaces) torch's generator the way it does for the reference gets the reference's stream for ANY number of rays
    c = n if not chunk else max(1, min(int(chunk), n))
    parts = {}
    for i0 in range(0, n, c):
        m = min(c, n - i0)
        if perturbed:
            parts.setdefault("t_rand", []).append(torch.rand(m, n_samples, device=dev))
        if std > 0.:
            parts.setdefault("noise0", []).append(torch.randn(m, n_samples, device=dev) * std)
        if n_importance > 0:
            if perturbed:                                   # det = (perturb == 0.), RN:474; fewer than 128: duplicated, as
                parts.setdefault("u", []).append(widen(torch.rand(m, n_importance, device=dev)))      # engine._host_tables
            if std > 0.:
                # (RN:371 draws [m, 64 + n_importance]; a 128-sample kernel rendering fewer importance samples through
                # duplicated uniforms takes 192 columns: the duplicates are zero-length intervals whose density is moot)
                parts.setdefault("noise1", []).append(torch.randn(m, n_samples + ni_kernel, device=dev) * std)
    for k in ("t_rand", "noise0", "u", "noise1"):           # (the order the tests and the docs name)
        if k in parts:
            d[k] = parts[k][0] if len(parts[k]) == 1 else torch.cat(parts[k], 0)
    return d


# ------------------------------------------------------------------------------------------------------
# the reference API
# ------------------------------------------------------------------------------------------------------
def batchify(fn, chunk):
    """RN:14-23.  Chunking bounds memory in the reference and does not change results (RN:67-68); the native
    path needs no chunking, so this returns fn."""
    return fn


def run_network(inputs, viewdirs, fn, embed_fn=None, embeddirs_fn=None, netchunk=1024 * 64):
    """RN:26-40: inputs [..., 3] points, viewdirs [N, 3] -> [..., 4], evaluated natively (encoding fused)."""
    flat = inputs.reshape(-1, 3)
    if viewdirs is None:                                    # use_viewdirs=False (RN:31): the network ignores directions
        if not (isinstance(fn, NeRF) and not fn.use_viewdirs):
            raise NotImplementedError("run_network: viewdirs=None needs a use_viewdirs=False NeRF module")
        dirs = torch.zeros_like(flat)
        dirs[:, 2] = 1.0
    else:
        dirs = viewdirs[:, None].expand(inputs.shape).reshape(-1, 3)
    if isinstance(fn, NeRF):
        out = fn.evaluate(flat, dirs)                       # straight to the native kernel: no [P,90] staging tensor
    else:                                                   # any other callable gets the reference's [P,90] layout
        x = torch.zeros(flat.shape[0], 90, dtype=torch.float32, device=flat.device)
        x[:, :3] = flat
        x[:, 63:66] = dirs
        out = fn(x)
    return out.reshape(list(inputs.shape[:-1]) + [out.shape[-1]])


def _check_viewdirs(name, use_viewdirs, kw):
    """use_viewdirs must say what the networks are (RN:60 / RH:92-96): a use_viewdirs=False network has no direction
    input at all -- it is served by the same kernels through NeRF.native_state_dict -- and a use_viewdirs=True network
    rendered without directions has no meaning."""
    for key in ("network_fn", "network_fine"):
        net = kw.get(key)
        if net is not None and bool(getattr(net, "use_viewdirs", True)) != bool(use_viewdirs):
            raise NotImplementedError("%s: use_viewdirs=%r with a %s built for use_viewdirs=%r"
                                      % (name, use_viewdirs, key, getattr(net, "use_viewdirs", True)))


def _check_kwargs(kw):
    """What NO renderer here serves (the fused kernels' own limits are checked on the handle that will run: _check_retraw)."""
    ns, ni = kw.get("N_samples", 64), kw.get("N_importance", 0)
    from .wide import MAX_SAMPLES
    if not (isinstance(ns, (int, np.integer)) and 3 <= ns <= MAX_SAMPLES) or not (isinstance(ni, (int, np.integer)) and 0 <= ni <= MAX_SAMPLES):
        raise NotImplementedError("render: unsupported option(s): N_samples=%r with N_importance=%r (N_samples 3..%d, N_importance "
                                  "0..%d)" % (ns, ni, MAX_SAMPLES, MAX_SAMPLES))


def _check_retraw(kw, model):
    """retraw wants the reference's [N, N_samples + N_importance, C]: the layered renderer returns exactly that; a fused handle
    does where its kernels evaluate exactly that many fine samples (128; 96 / 64 / 32 and the other native counts on f16x2
    handles) and the network has four output rows -- decided on the handle that will run, not on a guess about it."""
    if not kw.get("retraw", False) or getattr(model, "mlp", "").startswith("layered-"):
        return
    n_imp = kw.get("N_importance", 0)
    if n_imp not in (0, 128) and getattr(model, "ni_kernel", n_imp) != n_imp:
        raise NotImplementedError("render: retraw with N_importance=%r on a %s handle (its fine pass carries duplicated samples: "
                                  "raw would be [N,192,4]; the f16x2 kernels are specialised to N_importance 96, 64 and 32 and return "
                                  "the reference's raw there; NSR_LAYERED=1 serves any count)" % (n_imp, getattr(model, "mlp", "?")))
    net = kw.get("network_fine") if n_imp > 0 and kw.get("network_fine") is not None else kw.get("network_fn")
    if not getattr(net, "use_viewdirs", True) and getattr(net, "output_ch", 4) != 4:
        raise NotImplementedError("render: retraw with a use_viewdirs=False network of output_ch=%r on the fused kernels (the "
                                  "reference's raw is [N,S,%r]: RN:267, RH:119-120; they tap the four channels render_rays reads "
                                  "-- NeRF.evaluate / run_network return all of them, and so does the layered renderer: "
                                  "NSR_LAYERED=1)" % (net.output_ch, net.output_ch))


def render(H, W, K, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1., use_viewdirs=False,
           c2w_staticcam=None, **kwargs):
    """RN:58-123.  Returns [rgb_map, disp_map, acc_map, extras] with the reference's shapes: [H,W,...] when
    c2w is given, rays_d.shape[:-1] + ... for the rays form.  `chunk` does not change the deterministic render (RN:67-68; one
    persistent launch covers all rays); it only decides how the random draws of the stochastic options are laid out (_draws:
    the reference draws once per chunk of rays, and with pytest=True ray i gets row i mod chunk).
    ndc (RN:101-103), c2w_staticcam (RN:91-96), perturb > 0 (RN:447-459, RH:211) and raw_noise_std > 0 (RN:365-374) go
    through the per-ray extras of the native renderer (include/nsr.h: NsrRayExtras); see _draws for the random stream."""
    _check_viewdirs("render", use_viewdirs, kwargs)
    _check_kwargs(kwargs)
    n_imp = kwargs.get("N_importance", 0)
    from .run_nerf_helpers import DEFER_PATCH_CHECK, TRUST_PATCH_CALLS
    net_c, net_f = NeRF.adopt(kwargs["network_fn"]), NeRF.adopt(kwargs.get("network_fine", None) if n_imp > 0 else None)
    # the patch form (RN:168): the content fingerprint is enqueued AHEAD of the render and read while the render runs
    # (run_nerf_helpers: NSR_TRUST_VERSIONS); a mismatch renders again from repacked weights before this call returns
    token = NeRF.fingerprint_begin(net_c, net_f) if (rays is not None and DEFER_PATCH_CHECK and isinstance(net_c, NeRF)) else None
    model = _model_for(net_c, net_f, n_imp, kwargs, trust=rays is not None and (token is not None or TRUST_PATCH_CALLS))
    if token is not None:
        ret = _render_with(model, H, W, K, chunk, rays, c2w, ndc, near, far, use_viewdirs, c2w_staticcam, kwargs)
        if model.weights_fp not in ((), NeRF.fingerprint_end(token)):       # written through .data since they were packed
            model = _model_for(net_c, net_f, n_imp, kwargs)
            ret = _render_with(model, H, W, K, chunk, rays, c2w, ndc, near, far, use_viewdirs, c2w_staticcam, kwargs)
        return ret
    return _render_with(model, H, W, K, chunk, rays, c2w, ndc, near, far, use_viewdirs, c2w_staticcam, kwargs)


def _render_with(model, H, W, K, chunk, rays, c2w, ndc, near, far, use_viewdirs, c2w_staticcam, kwargs):
    """render() on the handle `model` (see there)."""
    per_ray_bounds = not (np.isscalar(near) and np.isscalar(far))          # RN:106-108: near / far may be arrays
    n_imp = kwargs.get("N_importance", 0)
    _check_retraw(kwargs, model)
    retraw = bool(kwargs.get("retraw", False))
    fine = n_imp > 0
    special = bool(ndc) or (c2w_staticcam is not None and use_viewdirs) or _stochastic(kwargs) or per_ray_bounds

    if c2w is not None:
        c2w = torch.as_tensor(c2w, dtype=torch.float32)
        if not c2w.requires_grad and not special:
            out = model.render_views(c2w.to(model.device), H, W, K, near, far, debug=retraw)
            sh = (int(H), int(W))
            ret = {k: v.reshape(sh + tuple(v.shape[1:])) for k, v in out.items()
                   if k in ("rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "z_std")}
            if retraw:
                r = out["raw" if fine else "raw0"]
                ret["raw"] = r.reshape(sh + tuple(r.shape[1:]))
            return _pack_ret(ret)
        rays_o, rays_d = _get_rays_autograd(H, W, K, c2w)
    else:
        rays_o, rays_d = rays
    rays_o = torch.as_tensor(rays_o, dtype=torch.float32)
    rays_d = torch.as_tensor(rays_d, dtype=torch.float32)
    viewdirs = None
    if not use_viewdirs:
        c2w_staticcam = None                                # RN:91-96 sits inside `if use_viewdirs:`: ignored without it
    if use_viewdirs and (ndc or c2w_staticcam is not None):     # RN:89-98: view directions from the rays BEFORE the next two
        v = rays_d.reshape(-1, 3).to(model.device)
        viewdirs = v / torch.norm(v, dim=-1, keepdim=True)
    if c2w_staticcam is not None:                           # RN:91-96
        rays_o, rays_d = _get_rays_autograd(H, W, K, c2w_staticcam)
    sh = tuple(rays_d.shape[:-1])
    ro = rays_o.reshape(-1, 3).to(model.device)
    rd = rays_d.reshape(-1, 3).to(model.device)
    if ndc:                                                 # RN:101-103 (the reference's callers pass near=0, far=1)
        ro, rd = _NdcRays.apply(ro, rd, model, int(H), int(W), float(K[0][0]), 1.0)
    if special:
        ex = _draws(kwargs, ro.shape[0], n_imp, model.device, chunk, model.ni_kernel, getattr(model, "n_samples", 64),
                    getattr(model, "u_width", 128))
        if per_ray_bounds:                                  # one bound per ray (the reference multiplies them into [N,1])
            for k, v in (("near", near), ("far", far)):
                t = torch.as_tensor(v, dtype=torch.float32).to(model.device).reshape(-1)
                if t.numel() not in (1, ro.shape[0]):
                    raise ValueError("render: %s has %d entries for %d rays" % (k, t.numel(), ro.shape[0]))
                ex[k] = t.expand(ro.shape[0]).contiguous()
            near = far = 0.0
        outs = _RenderRaysEx.apply(ro, rd, viewdirs, model, float(near), float(far), retraw, ex)
    else:
        outs = _RenderRays.apply(ro, rd, model, float(near), float(far), retraw)
    keys = ["rgb_map", "disp_map", "acc_map"] + (["rgb0", "disp0", "acc0", "z_std"] if fine else [])
    if retraw:
        keys.append("raw")
    ret = {k: v.reshape(sh + tuple(v.shape[1:])) for k, v in zip(keys, outs)}
    return _pack_ret(ret)


def _stochastic(kw):
    perturb = kw.get("perturb", 0.)
    return ((perturb not in (0, 0., False) and perturb > 0.) or float(kw.get("raw_noise_std", 0.) or 0.) > 0.
            or bool(kw.get("pytest", False)))


def _pack_ret(ret):
    k_extract = ["rgb_map", "disp_map", "acc_map"]        # RN:120-123
    return [ret[k] for k in k_extract] + [{k: v for k, v in ret.items() if k not in k_extract}]


def _scaled_hw(hwf, render_factor):
    H, W, focal = hwf
    if render_factor != 0:                                 # RN:217-221 (K itself is not rescaled there either)
        H, W, focal = H // render_factor, W // render_factor, focal / render_factor
    return int(H), int(W), focal


def _path_setup(name, hwf, render_factor, render_kwargs, need_fine=False):
    """-> (H, W, near, far, handle, general).  general: ndc=True / perturb > 0 / raw_noise_std > 0 are in the kwargs -- the
    reference forwards **render_kwargs to render() unchanged (RN:233, RN:168), so render_kwargs_train works there; here
    such a call goes through render() too (per pose), the deterministic one through the batched launches."""
    H, W, _ = _scaled_hw(hwf, render_factor)
    kw = dict(render_kwargs)
    near, far = kw.pop("near", 0.), kw.pop("far", 1.)
    general = bool(kw.pop("ndc", True)) or _stochastic(kw)
    _check_viewdirs(name, kw.pop("use_viewdirs", False), kw)
    _check_kwargs(kw)
    n_imp = kw.get("N_importance", 0)
    model = _model_for(kw["network_fn"], kw.get("network_fine", None) if n_imp > 0 else None, n_imp, kw)
    if need_fine and n_imp == 0 and not getattr(model, "mlp", "").startswith("layered-"):
        raise NotImplementedError("%s needs the coarse+fine configuration (N_importance > 0) on the fused kernels (their "
                                  "input-gradient kernels differentiate the fine pass; NSR_LAYERED=1 serves coarse-only)" % name)
    return H, W, near, far, model, general


def render_path(categorical_prob, render_poses, hwf, K, chunk, render_kwargs, gt_imgs=None, savedir=None,
                object_id=2, render_factor=0):
    """RN:213-255.  All poses are rendered by ONE persistent-kernel launch (the reference loops serially,
    RN:229); PNGs land in savedir/<object_id>/{i:03d}.png exactly as before.  `categorical_prob` is unused
    (as in the reference).  Returns (rgbs [K,H,W,3] float32, disps [K,H,W] float32) numpy arrays.

    Multi-GPU: when a torch.distributed process group with more than one rank is initialised (the
    script runs under torchrun), pose i is rendered by rank i mod world, every rank writes the PNGs of its own
    poses, the images are all-gathered (RCCL) and EVERY rank returns all K views in pose order -- the call site
    NM:128 needs no change.  NSR_AUTO_SHARD=0 disables it."""
    from . import dist as D
    H, W, near, far, model, general = _path_setup("render_path", hwf, render_factor, render_kwargs)
    if savedir is not None:
        os.makedirs(os.path.join(savedir, str(object_id)), exist_ok=True)
    poses = torch.as_tensor(render_poses, dtype=torch.float32).detach()

    def render_fn(p):
        if p.shape[0] == 0:
            return (torch.zeros((0, H, W, 3), device=model.device), torch.zeros((0, H, W), device=model.device))
        if general:               # ndc / stochastic options: pose by pose through render(), as the reference does (RN:229-235)
            r = [render(H, W, K, chunk=chunk, c2w=c[:3, :4], **render_kwargs)[:2] for c in p]
            return torch.stack([x[0] for x in r]), torch.stack([x[1] for x in r])
        out = model.render_views(p[:, :3, :4].to(model.device), H, W, K, near, far)
        return out["rgb_map"].reshape(-1, H, W, 3), out["disp_map"].reshape(-1, H, W)

    t = time.time()
    with torch.no_grad():
        if D.auto_shard_enabled():
            rgbs, disps = D.render_path_distributed(render_fn, poses, savedir=savedir, object_id=object_id)
            print("rendered %d views on %d ranks in %.3f s" % (rgbs.shape[0], D.world_info()[0], time.time() - t))
            return rgbs, disps
        rgb, disp = render_fn(poses)
        rgbs, disps = rgb.cpu().numpy(), disp.cpu().numpy()
        _note_range(model, render_kwargs.get("network_fn"))
    print("rendered %d views in %.3f s" % (rgbs.shape[0], time.time() - t))
    if savedir is not None:
        png.imwrite_many([os.path.join(savedir, str(object_id), "{:03d}.png".format(i)) for i in range(rgbs.shape[0])],
                         [to8b(rgbs[i]) for i in range(rgbs.shape[0])])
    return rgbs, disps


def _pose_patch_grads(model, c2w, cot, H, W, K, near, far, N_rand, render_kwargs=None):
    """One pose of render_path_grad on the device: (rgb [H,W,3], dL/d c2w[3,4] per patch [n_patches,3,4]).
    render_kwargs (given for ndc=True / perturb > 0 / raw_noise_std > 0, which RN:168 forwards to render() like everything
    else): the whole image goes through render() -- its autograd Functions carry the ndc projection, the given view
    directions and the draws -- as ONE forward and ONE input-gradient launch; with chunk = N_rand the draws are made patch
    by patch in the reference's order (one render() call = one chunk per patch there, RN:168 / _draws)."""
    with torch.no_grad():
        ro, rd = model.get_rays(H, W, K, c2w[:3, :4].detach().to(model.device))
    if render_kwargs is not None:
        rays = torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)], 0).requires_grad_(True)
        rgb = render(H, W, K, chunk=N_rand, rays=rays, **render_kwargs)[0]
        (g,) = torch.autograd.grad(rgb, rays, grad_outputs=cot.to(rgb.device))
        with torch.no_grad():
            g_pose = model.pose_grad(g[0].contiguous(), g[1].contiguous(), H, W, K, N_rand)
        return rgb.detach().reshape(H, W, 3), g_pose
    with torch.no_grad():
        go, gd, out = model.render_rays_vjp(ro.reshape(-1, 3), rd.reshape(-1, 3), near, far, cot, with_forward=True)
        g_pose = model.pose_grad(go, gd, H, W, K, N_rand)
    return out["rgb_map"].reshape(H, W, 3), g_pose


def render_path_grad(categorical_prob, render_poses, hwf, K, chunk, grad_E, render_kwargs, gt_imgs=None,
                     savedir=None, object_id=2, render_factor=0):
    """RN:126-210: per pose, the image and, per `chunk`-ray row-major patch, dL/d(categorical_prob) [8] for the
    detector cotangent grad_E[i]['grad_E'][0] ([3,H,W], used in its own channel order exactly like RN:154).

    The reference runs 313 forward+2 autograd calls per 400x400 image; here ONE forward+backward launch per pose
    returns dL/d(rays) for the whole image, a second tiny kernel contracts it per patch with d(rays)/d(c2w)
    (linear, RH:160-164) to [n_patches,3,4], and the 12x8 Jacobian d(c2w)/d(psi) of the caller's own pose graph
    (sample_pose, LL:202-247) finishes the chain -- the same numbers as the per-patch autograd.grad of RN:177-181.

    Multi-GPU: under an initialised process group pose i goes to rank i mod world; images and per-patch gradients
    are all-gathered, so every rank returns the reference's full (rgbs, dLdpsis) in pose-major order and NM:184-191
    run unchanged (the mean over the stacked list equals dist.mean_psi_grad's all-reduce)."""
    from . import dist as D
    H, W, near, far, model, general = _path_setup("render_path_grad", hwf, render_factor, render_kwargs, need_fine=True)
    n_rays = H * W
    N_rand = int(chunk)
    n_patches = (n_rays + N_rand - 1) // N_rand
    n_poses = min(len(render_poses), len(grad_E))              # RN:142
    shard = D.auto_shard_enabled()
    world, rank = D.world_info() if shard else (1, 0)
    if shard:
        D.check_distinct_devices()
        D.check_same_poses(torch.stack([torch.as_tensor(p, dtype=torch.float32).detach() for p in render_poses[:n_poses]])
                           if n_poses else torch.zeros(0, 4, 4))
    mine = D.shard_indices(n_poses, world, rank)
    jac_all = getattr(render_poses, "nsr_jac", None)           # [K,12,n_cat] when the poses come from the device sampler
    rgb_l, grad_l = [], []
    for i_pose in mine:
        c2w = render_poses[i_pose]
        pose = c2w[:3, :4]
        g = grad_E[i_pose]["grad_E"][0]
        g = torch.as_tensor(g.detach().cpu().numpy().transpose(1, 2, 0) if isinstance(g, torch.Tensor)
                            else np.asarray(g).transpose(1, 2, 0), dtype=torch.float32)      # RN:154 CHW -> HWC
        cot = g.reshape(-1, 3).to(model.device).contiguous()
        rgb, g_pose = _pose_patch_grads(model, pose, cot, H, W, K, near, far, N_rand,        # [n_patches,3,4]
                                        render_kwargs if general else None)
        if jac_all is not None:
            J = jac_all[i_pose]                    # the device sampler's own Jacobian (pose.sample_pose_device): no autograd
        else:
            # d vec(c2w[:3,:4]) / d psi through the caller's graph: 12 rows, one batched autograd call
            basis = torch.eye(12, dtype=pose.dtype, device=pose.device).reshape(12, 3, 4)
            (J,) = torch.autograd.grad(pose, categorical_prob, grad_outputs=basis, retain_graph=True,
                                       is_grads_batched=True)                                # [12, n_cat]
        grad_l.append((g_pose.reshape(n_patches, 12).to(J.device, J.dtype) @ J).detach().to(torch.float32).cpu())
        rgb_l.append(rgb.cpu())
        if savedir is not None:
            d = os.path.join(savedir, str(object_id), "withgrad")
            os.makedirs(d, exist_ok=True)
            png.imwrite(os.path.join(d, "{:03d}.png".format(i_pose)), to8b(rgb_l[-1].numpy()))  # RN:200-206
    n_cat = int(categorical_prob.numel())
    rgbs = torch.stack(rgb_l) if rgb_l else torch.zeros((0, H, W, 3))
    grads = torch.stack(grad_l) if grad_l else torch.zeros((0, n_patches, n_cat))
    if shard:
        rgbs = D.gather_views(rgbs, n_poses).cpu()
        grads = D.gather_patch_grads(grads, n_poses).cpu()
        if savedir is not None:
            torch.distributed.barrier()
    _note_range(model, render_kwargs.get("network_fn"))
    dLdpsis = [grads[i, p] for i in range(grads.shape[0]) for p in range(n_patches)]         # RN:190 order
    return rgbs.numpy(), dLdpsis


def create_nerf(args):
    """RN:258-340: builds the coarse/fine networks, loads `args.ft_path` (or the newest .tar under
    basedir/expname) and returns (render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer)."""
    embed_fn, input_ch = get_embedder(args.multires, args.i_embed)
    input_ch_views = 0
    embeddirs_fn = None
    if args.use_viewdirs:
        embeddirs_fn, input_ch_views = get_embedder(args.multires_views, args.i_embed)
    output_ch = 5 if args.N_importance > 0 else 4
    skips = [4]
    model = NeRF(D=args.netdepth, W=args.netwidth, input_ch=input_ch, output_ch=output_ch, skips=skips,
                 input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs).to(device)
    grad_vars = list(model.parameters())
    model_fine = None
    if args.N_importance > 0:
        model_fine = NeRF(D=args.netdepth_fine, W=args.netwidth_fine, input_ch=input_ch, output_ch=output_ch,
                          skips=skips, input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs).to(device)
        grad_vars += list(model_fine.parameters())

    network_query_fn = lambda inputs, viewdirs, network_fn: run_network(
        inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, netchunk=args.netchunk)
    optimizer = torch.optim.Adam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))

    start = 0
    basedir, expname = args.basedir, args.expname
    if args.ft_path is not None and args.ft_path != "None":
        ckpts = [args.ft_path]
    else:
        d = os.path.join(basedir, expname)
        ckpts = [os.path.join(d, f) for f in sorted(os.listdir(d)) if "tar" in f] if os.path.isdir(d) else []
    print("Found ckpts", ckpts)
    if len(ckpts) > 0 and not args.no_reload:
        ckpt_path = ckpts[-1]
        print("Reloading from", ckpt_path)
        ckpt = torch.load(ckpt_path, map_location=device, weights_only=False)
        start = ckpt["global_step"]
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
        model.load_state_dict(ckpt["network_fn_state_dict"])
        if model_fine is not None:
            model_fine.load_state_dict(ckpt["network_fine_state_dict"])

    render_kwargs_train = {
        "network_query_fn": network_query_fn, "perturb": args.perturb, "N_importance": args.N_importance,
        "network_fine": model_fine, "N_samples": args.N_samples, "network_fn": model,
        "use_viewdirs": args.use_viewdirs, "white_bkgd": args.white_bkgd, "raw_noise_std": args.raw_noise_std,
    }
    if args.dataset_type != "llff" or args.no_ndc:
        print("Not ndc!")
        render_kwargs_train["ndc"] = False
        render_kwargs_train["lindisp"] = args.lindisp
    render_kwargs_test = dict(render_kwargs_train)
    render_kwargs_test["perturb"] = False
    render_kwargs_test["raw_noise_std"] = 0.
    return render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer
