"""Relu-flip census of the input-gradient kernels: per-ray attribution of the difference between a kernel's
dL/d(rays_o, rays_d) and the oracle's float64 backprop at the same sample depths.  TEST INFRASTRUCTURE ONLY.

The gradient the bilevel loop takes (RN:174-181: autograd.grad(rgb_p, batch_rays, ...)) is piecewise smooth in the
network's pre-activations: relu'(x) = [x > 0] (RH:99-118, RN:356) switches a whole unit's contribution on or off at
x = 0.  Two evaluations of the forward pass that agree to fp32 rounding everywhere can disagree about the SIGN of a
pre-activation that sits within rounding of zero, and the gradient of that point then differs by that unit's whole
share -- far more than rounding.  The forward path got oracle/census.py for its discontinuities; this is the same for
the backward path.

Inputs: the kernel's gradients AND its debug taps (include/nsr.h: NsrVjpDebugOut) -- the relu patterns its backward
applied, dL/d raw as its compositing backward produced it, and its per-sample dL/d pts / dL/d viewdirs.  For every ray
whose relative gradient error against the oracle exceeds `thr`:

  (a) the hidden units whose on/off state differs from the oracle's float64 forward are counted, and each of them must sit
      AT the discontinuity: |pre-activation| <= margin * (sum_k |W_jk| |h_k| + |b_j|), the forward rounding scale of
      that unit (anything larger is a wrong forward pass, not a flip);
  (b) the oracle's backprop is replayed WITH the kernel's patterns (relu' as the kernel saw it, and sign(sigma) of RN:356
      where the oracle's sigma is itself within rounding of zero); the kernel's gradient must agree with the replay to `thr`;
  (c) anything else is `unattributed`.
Independently of the flags, per_point() measures the network backward alone -- oracle.network_vjp on the kernel's OWN
dL/d raw with the kernel's OWN patterns against the kernel's per-sample gradients -- which is where a defect of the
per-point gradient normalisation of csrc/nsr_h2_bwd.inc would show, flips excluded by construction."""
import numpy as np

import nerf_oracle as O

f64 = np.float64
MARGIN = 2e-5          # a flipped unit's |pre-activation| relative to sum |w||h| + |b| (fp32-grade forward: ~1e-6)


def decode_masks(masks, n_rays):
    """relu_masks [n_items,3,9,256,4] (uint32 bit patterns, include/nsr.h: NsrVjpDebugOut) -> (on_pre [N,192,8,256] bool,
    on_av [N,192,128] bool): True = unit ON.  Item t = rays 2t, 2t+1; fine pass p covers points q = 128 p + 32 w + j."""
    m = np.ascontiguousarray(masks).view(np.uint32).reshape(-1, 3, 9, 256, 4)
    n_items = m.shape[0]
    assert n_items == (n_rays + 1) // 2
    bits = ((m[..., None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool)      # [..., word, bit]
    e = 31 - np.arange(32)                                                            # bit -> element 16 (mo & 1) + r
    word = np.arange(4)[:, None]
    mo = 2 * word + (e[None, :] >> 4)                                                 # [4,32]
    r16 = np.broadcast_to(e[None, :] & 15, (4, 32))
    tid = np.arange(256)
    w, lane = tid >> 6, tid & 63
    j, h = lane & 31, lane >> 5
    unit = 32 * mo[None] + (r16[None] & 3) + 8 * (r16[None] >> 2) + 4 * h[:, None, None]     # [256,4,32]
    on = np.zeros((n_items * 2, 192, 9, 256), bool)
    for p in range(3):
        q = 128 * p + 32 * w + j                                                      # [256] point of the item
        ray, smp = q // 192, q % 192
        for L in range(9):
            off = bits[:, p, L]                                                       # [n_items,256,4,32] True = OFF
            for t in range(n_items):
                # lanes j and j + 32 hold complementary halves of the same point's units
                on[2 * t + ray[:, None, None], smp[:, None, None], L, unit] = ~off[t]
    on = on[:n_rays]
    return on[:, :, :8, :], on[:, :, 8, :128]


def _unit_margin(sd, fwd, e_p, e_d, layer, pt, unit):
    """|pre| / (sum |w||h| + |b|) of one unit of one point in the oracle's float64 forward (layer 8 = views_linears.0)."""
    W = lambda k: sd[k + ".weight"].astype(f64)
    B = lambda k: sd[k + ".bias"].astype(f64)
    if layer < 8:
        if layer == 0:
            x = e_p[pt]
        else:
            x = np.maximum(fwd["pre"][layer - 1][pt], 0)
            if layer - 1 == O.SKIP_AT:
                x = np.concatenate([e_p[pt], x])
        w, b, pre = W("pts_linears.%d" % layer)[unit], B("pts_linears.%d" % layer)[unit], fwd["pre"][layer][pt, unit]
    else:
        h = np.maximum(fwd["pre"][7][pt], 0)
        feat = h @ W("feature_linear").T + B("feature_linear")
        x = np.concatenate([feat, e_d[pt]])
        w, b, pre = W("views_linears.0")[unit], B("views_linears.0")[unit], fwd["av"][pt, unit]
    return abs(pre) / (np.abs(w) @ np.abs(x) + abs(b) + 1e-300)


def _rel(a, b):
    """per-ray relative error of [go | gd] rows"""
    return np.linalg.norm(a.astype(f64) - b.astype(f64), axis=1) / (np.linalg.norm(b.astype(f64), axis=1) + 1e-300)


def census(nets, rays_o, rays_d, near, far, cot, z_fine, got, thr, margin=MARGIN, white_bkgd=False, max_listed=8, **kw):
    """nets = (sd_coarse, sd_fine); rays [N,3]; cot [N,3]; z_fine [N,192]: the depths the kernel differentiated at;
    got: dict(grad_o, grad_d [N,3], relu_masks, grad_raw [N,192,4], grad_pts [N,192,6]) of the kernel; thr: relative
    per-ray tolerance (the tests pass 10 x the fp32 kernel's 95th percentile).  Returns a JSON-able dict; `unattributed`
    must be 0."""
    sd_c, sd_f = nets
    sd = sd_f if sd_f is not None else sd_c
    N = rays_o.shape[0]
    parts = {}
    go, gd, _ = O.render_rays_vjp(sd_c, sd_f, rays_o, rays_d, near, far, cot, z_fine=z_fine, white_bkgd=white_bkgd,
                                  parts=parts, **kw)
    g_or = np.concatenate([go, gd], 1)
    g_k = np.concatenate([got["grad_o"], got["grad_d"]], 1)
    e = _rel(g_k, g_or)
    on_pre, on_av = decode_masks(got["relu_masks"], N)
    fwd = parts["fwd"]
    P = N * 192
    k_pre = [on_pre[:, :, L].reshape(P, 256) for L in range(8)]
    k_av = on_av.reshape(P, 128)
    o_pre = [p > 0 for p in fwd["pre"]]
    o_av = fwd["av"] > 0
    flip_pre = [k_pre[L] != o_pre[L] for L in range(8)]
    flip_av = k_av != o_av
    flips_pt = sum(f.sum(1) for f in flip_pre) + flip_av.sum(1)                     # per point
    flips_ray = flips_pt.reshape(N, 192).sum(1)
    # sign(sigma) of RN:356 as the kernel saw it: dL/d sigma is exactly 0 where its relu is off
    sig = parts["sigma"]                                                            # [N,192] float64
    k_sig_on = got["grad_raw"][..., 3] != 0
    sig_scale = np.abs(sig).max() + 1e-300
    near0 = np.abs(sig) <= 1e-5 * sig_scale                                          # oracle's sigma within rounding of 0
    sigma_on = np.where(near0, k_sig_on, sig > 0)
    sig_flips_ray = ((sigma_on != (sig > 0)).sum(1))
    # (b) replay with the kernel's patterns
    go_r, gd_r, _ = O.render_rays_vjp(sd_c, sd_f, rays_o, rays_d, near, far, cot, z_fine=z_fine, white_bkgd=white_bkgd,
                                      relu_on=dict(pre=k_pre, av=k_av), sigma_on=sigma_on, **kw)
    e_replay = _rel(g_k, np.concatenate([go_r, gd_r], 1))
    flagged = e > thr
    # (a) every flipped unit of a flagged ray must sit at the discontinuity
    e_p = O.embed(parts["pts"], O.MULTIRES).astype(f64)
    e_d = O.embed(parts["dirs"], O.MULTIRES_VIEWS).astype(f64)
    worst_margin, off_cliff = 0.0, np.zeros(N, bool)
    for r in np.nonzero(flagged)[0]:
        pts_r = np.arange(r * 192, (r + 1) * 192)
        for L in range(9):
            fl = (flip_pre[L] if L < 8 else flip_av)[pts_r]
            for pi, u in zip(*np.nonzero(fl)):
                mg = _unit_margin(sd, fwd, e_p, e_d, L, pts_r[pi], u)
                worst_margin = max(worst_margin, mg)
                if mg > margin:
                    off_cliff[r] = True
    attributed = flagged & (e_replay <= thr) & ~off_cliff
    unattributed = flagged & ~attributed
    listed = [dict(ray=int(r), err=float(e[r]), err_replay=float(e_replay[r]), flipped_units=int(flips_ray[r]),
                   sigma_flips=int(sig_flips_ray[r]), off_cliff=bool(off_cliff[r]))
              for r in np.nonzero(flagged)[0][np.argsort(-e[flagged])][:max_listed]]
    q = lambda x, p: float(np.percentile(x, p)) if x.size else 0.0
    return dict(rays=int(N), thr=float(thr), rays_above_thr=int(flagged.sum()), attributed=int(attributed.sum()),
                unattributed=int(unattributed.sum()), rays_with_flips=int((flips_ray > 0).sum()),
                flipped_units_total=int(flips_ray.sum()), flagged_without_flips=int((flagged & (flips_ray == 0) & (sig_flips_ray == 0)).sum()),
                sigma_flips_total=int(sig_flips_ray.sum()), worst_flip_margin=float(worst_margin),
                err_p50=q(e, 50), err_p95=q(e, 95), err_max=float(e.max()),
                replay_p50=q(e_replay, 50), replay_p95=q(e_replay, 95), replay_max=float(e_replay.max()),
                max_err_unflagged=float(e[~flagged].max()) if (~flagged).any() else 0.0, worst=listed)


def per_point(nets, rays_o, rays_d, z_fine, got, viewdirs=None):
    """The network backward alone: oracle.network_vjp (float64) on the kernel's OWN dL/d raw with the kernel's OWN relu
    patterns, against the kernel's per-sample (dL/d pts, dL/d viewdirs).  Errors are relative to the largest per-sample
    gradient of the RAY (a sample whose own gradient is 1e-12 of its ray's does not matter to the ray).  Returns
    dict(p50, p99, max) -- flips cannot contribute by construction."""
    sd = nets[1] if nets[1] is not None else nets[0]
    N = rays_o.shape[0]
    z = z_fine.astype(np.float32)
    pts = O._add(rays_o[:, None, :].astype(np.float32), (rays_d[:, None, :].astype(np.float32) * z[:, :, None]).astype(np.float32)).reshape(-1, 3)
    vd = O.normalize_dirs(rays_d.astype(np.float32)) if viewdirs is None else viewdirs.astype(np.float32)
    dirs = np.broadcast_to(vd[:, None, :], (N, 192, 3)).reshape(-1, 3)
    on_pre, on_av = decode_masks(got["relu_masks"], N)
    P = N * 192
    relu_on = dict(pre=[on_pre[:, :, L].reshape(P, 256) for L in range(8)], av=on_av.reshape(P, 128))
    gp, gv = O.network_vjp(sd, pts, dirs, got["grad_raw"].reshape(P, 4), relu_on=relu_on)
    ref = np.concatenate([gp, gv], 1).reshape(N, 192, 6)
    k = got["grad_pts"].astype(f64)
    scale = np.abs(ref).reshape(N, -1).max(1)[:, None, None] + 1e-300
    err = (np.abs(k - ref) / scale).reshape(N, -1).max(1)                           # per ray: worst sample, relative to the ray
    return dict(p50=float(np.percentile(err, 50)), p90=float(np.percentile(err, 90)), p99=float(np.percentile(err, 99)),
                max=float(err.max()), argmax_ray=int(err.argmax()))


def attribute_without_taps(nets, rays_o, rays_d, near, far, cot, z_fine, grad_o, grad_d, rays, thr, margin=MARGIN,
                           max_candidates=48, white_bkgd=False):
    """The same attribution for a kernel that has NO debug taps (the x16 fp32 kernels): for each ray of `rays`, the hidden
    units of its 192 samples whose float64 pre-activation sits within `margin` of zero (relative to sum |w||h| + |b|) are
    the only ones whose relu' another evaluation can see differently.  A flip of one unit changes the gradient of ONE sample;
    its effect on (dL/d rays_o, dL/d rays_d) is computed by re-running the float64 backprop of that sample with the unit
    flipped.  The ray is attributed if the oracle's gradient plus the effect of ONE such flip -- or of a pair -- reproduces
    the kernel's gradient to `thr`.  Returns {ray: dict(flips=[(sample, layer, unit, margin), ...] or None, err, err_after)}."""
    sd_c, sd_f = nets
    sd = sd_f if sd_f is not None else sd_c
    out = {}
    for r in rays:
        r = int(r)
        ro, rd, ct, zf = rays_o[r:r + 1], rays_d[r:r + 1], cot[r:r + 1], z_fine[r:r + 1]
        parts = {}
        go, gd, _ = O.render_rays_vjp(sd_c, sd_f, ro, rd, near, far, ct, z_fine=zf, white_bkgd=white_bkgd, parts=parts)
        g0 = np.concatenate([go, gd], 1)[0].astype(f64)
        gk = np.concatenate([grad_o[r], grad_d[r]]).astype(f64)
        err = np.linalg.norm(gk - g0) / (np.linalg.norm(g0) + 1e-300)
        fwd, pts, dirs, g_raw = parts["fwd"], parts["pts"], parts["dirs"], parts["g_raw"].reshape(-1, 4)
        e_p = O.embed(pts, O.MULTIRES).astype(f64)
        e_d = O.embed(dirs, O.MULTIRES_VIEWS).astype(f64)
        # candidates: units within `margin` of the discontinuity, closest first
        cand = []
        W = lambda k: sd[k + ".weight"].astype(f64)
        B = lambda k: sd[k + ".bias"].astype(f64)
        for L in range(9):
            if L < 8:
                x = e_p if L == 0 else np.maximum(fwd["pre"][L - 1], 0)
                if L - 1 == O.SKIP_AT:
                    x = np.concatenate([e_p, x], 1)
                w, b, pre = W("pts_linears.%d" % L), B("pts_linears.%d" % L), fwd["pre"][L]
            else:
                feat = np.maximum(fwd["pre"][7], 0) @ W("feature_linear").T + B("feature_linear")
                x = np.concatenate([feat, e_d], 1)
                w, b, pre = W("views_linears.0"), B("views_linears.0"), fwd["av"]
            mag = np.abs(x) @ np.abs(w).T + np.abs(b)
            rel = np.abs(pre) / (mag + 1e-300)
            for s, u in zip(*np.nonzero(rel <= margin)):
                cand.append((float(rel[s, u]), int(s), L, int(u)))
        cand.sort()
        cand = cand[:max_candidates]
        v = O.normalize_dirs(rd.astype(np.float32)).astype(f64)[0]
        nrm = float(O.dir_norm(rd)[0])
        base = {}

        def point_grad(s, flip=None):
            on = dict(pre=[(fwd["pre"][L][s:s + 1] > 0).copy() for L in range(8)], av=(fwd["av"][s:s + 1] > 0).copy())
            if flip is not None:
                L, u = flip
                (on["pre"][L] if L < 8 else on["av"])[0, u] ^= True
            one = dict(pre=[p[s:s + 1] for p in fwd["pre"]], av=fwd["av"][s:s + 1], sigma=None, rgb_raw=None)
            gp, gv = O.network_vjp(sd, pts[s:s + 1], dirs[s:s + 1], g_raw[s:s + 1], one, relu_on=on)
            return gp[0], gv[0]

        effects = []
        for rel, s, L, u in cand:
            if s not in base:
                base[s] = point_grad(s)
            gp, gv = point_grad(s, (L, u))
            dgp, dgv = gp - base[s][0], gv - base[s][1]
            z = float(zf[0, s])
            effects.append(np.concatenate([dgp, z * dgp + (dgv - v * float(dgv @ v)) / nrm]))
        best, best_err = None, err
        rel_err = lambda g: np.linalg.norm(gk - g) / (np.linalg.norm(g) + 1e-300)
        for i, e1 in enumerate(effects):
            e_ = rel_err(g0 + e1)
            if e_ < best_err:
                best, best_err = [i], e_
        if best_err > thr:
            for i in range(len(effects)):
                for j in range(i + 1, len(effects)):
                    e_ = rel_err(g0 + effects[i] + effects[j])
                    if e_ < best_err:
                        best, best_err = [i, j], e_
        ok = best is not None and best_err <= thr
        out[r] = dict(err=float(err), err_after=float(best_err), candidates=len(cand),
                      flips=[(cand[i][1], cand[i][2], cand[i][3], cand[i][0]) for i in best] if ok else None)
    return out
