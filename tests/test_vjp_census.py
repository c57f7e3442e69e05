"""The relu-flip census of the input-gradient kernels (oracle/vjp_census.py) on the CPU.  The stand-in "kernel" is the
oracle's own float64 backprop with chosen relu patterns, its taps encoded in the kernels' bit layout (include/nsr.h:
NsrVjpDebugOut): a flip AT the discontinuity must come out attributed; a flip far from it, a fabricated gradient and a
wrong network backward must not.  The GPU tests run the same census on the HIP kernels' taps."""
import numpy as np
import pytest


def encode_masks(on_pre, on_av):
    """Inverse of vjp_census.decode_masks: (on_pre [N,192,8,256], on_av [N,192,128]) -> uint32 [ceil(N/2),3,9,256,4]."""
    N = on_pre.shape[0]
    n_items = (N + 1) // 2
    on = np.ones((n_items * 2, 192, 9, 256), bool)
    on[:N, :, :8] = on_pre
    on[:N, :, 8, :128] = on_av
    out = np.zeros((n_items, 3, 9, 256, 4), np.uint32)
    tid = np.arange(256)
    w, lane = tid >> 6, tid & 63
    j, h = lane & 31, lane >> 5
    for p in range(3):
        q = 128 * p + 32 * w + j
        ray, smp = q // 192, q % 192
        for L in range(9):
            for mo in range(8 if L < 8 else 4):
                for r in range(16):
                    unit = 32 * mo + (r & 3) + 8 * (r >> 2) + 4 * h                          # [256]
                    bit = 31 - (16 * (mo & 1) + r)
                    for t in range(n_items):
                        off = ~on[2 * t + ray, smp, L, unit]
                        out[t, p, L, :, mo >> 1] |= off.astype(np.uint32) << np.uint32(bit)
    return out


@pytest.fixture(scope="module")
def case(oracle, synth_nets):
    rng = np.random.RandomState(5)
    K = oracle.scaled_K(100.0)
    c2w = oracle.sweep_poses(1, seed=3)[0]
    ro, rd = oracle.get_rays(4, 4, K, c2w[:3, :4])
    ro, rd = ro.reshape(-1, 3)[:6].astype(np.float32), rd.reshape(-1, 3)[:6].astype(np.float32)
    cot = rng.standard_normal((6, 3)).astype(np.float32)
    zf = oracle.render_rays(synth_nets[0], synth_nets[1], ro, rd, oracle.normalize_dirs(rd), oracle.YCBV_NEAR,
                            oracle.YCBV_FAR, extras=True)["z_fine"]
    parts = {}
    go, gd, _ = oracle.render_rays_vjp(synth_nets[0], synth_nets[1], ro, rd, oracle.YCBV_NEAR, oracle.YCBV_FAR, cot,
                                       z_fine=zf, parts=parts)
    return dict(ro=ro, rd=rd, cot=cot, zf=zf, go=go, gd=gd, parts=parts)


def _kernel_like(oracle, nets, c, on_pre, on_av):
    """the 'kernel': the oracle's backprop under the given relu patterns, with its taps"""
    P = on_pre.shape[0] * 192
    parts = {}
    go, gd, _ = oracle.render_rays_vjp(nets[0], nets[1], c["ro"], c["rd"], oracle.YCBV_NEAR, oracle.YCBV_FAR, c["cot"],
                                       z_fine=c["zf"], parts=parts,
                                       relu_on=dict(pre=[on_pre[:, :, L].reshape(P, 256) for L in range(8)],
                                                    av=on_av.reshape(P, 128)))
    return dict(grad_o=go, grad_d=gd, relu_masks=encode_masks(on_pre, on_av),
                grad_raw=parts["g_raw"].astype(np.float32),
                grad_pts=np.concatenate([parts["g_pts"], parts["g_dirs"]], -1).astype(np.float32))


def _own_patterns(c):
    fwd = c["parts"]["fwd"]
    N = c["ro"].shape[0]
    on_pre = np.stack([p > 0 for p in fwd["pre"]], 1).reshape(N, 192, 8, 256)
    return on_pre, (fwd["av"] > 0).reshape(N, 192, 128)


def test_mask_codec_round_trip(case):
    import vjp_census as V
    on_pre, on_av = _own_patterns(case)
    a, b = V.decode_masks(encode_masks(on_pre, on_av), on_pre.shape[0])
    assert np.array_equal(a, on_pre) and np.array_equal(b, on_av)


def test_same_patterns_nothing_to_attribute(oracle, synth_nets, case):
    import vjp_census as V
    on_pre, on_av = _own_patterns(case)
    got = _kernel_like(oracle, synth_nets, case, on_pre, on_av)
    c = V.census(synth_nets, case["ro"], case["rd"], oracle.YCBV_NEAR, oracle.YCBV_FAR, case["cot"], case["zf"], got, thr=1e-5)
    assert c["rays_above_thr"] == 0 and c["unattributed"] == 0 and c["flipped_units_total"] == 0, c
    pp = V.per_point(synth_nets, case["ro"], case["rd"], case["zf"], got)
    assert pp["max"] < 1e-6, pp


def _flip(case, on_pre, want_small):
    """flip the unit of ray 2 with the smallest (or a large) |pre-activation| among the samples that carry gradient"""
    fwd = case["parts"]["fwd"]
    gp = np.abs(case["parts"]["g_pts"][2]).sum(-1)                                     # [192]
    smp = int(np.argmax(gp))
    pre = np.abs(fwd["pre"][6][2 * 192 + smp])
    unit = int(np.argmin(pre)) if want_small else int(np.argmax(pre))
    out = on_pre.copy()
    out[2, smp, 6, unit] ^= True
    return out, float(pre[unit])


def test_flip_at_the_discontinuity_is_attributed(oracle, synth_nets, case):
    import vjp_census as V
    on_pre, on_av = _own_patterns(case)
    flipped, pre = _flip(case, on_pre, want_small=True)
    got = _kernel_like(oracle, synth_nets, case, flipped, on_av)
    # the smallest of 256 pre-activations of one point is not within fp32 rounding of zero: give the census the margin the
    # fabricated flip needs, the mechanism under test is "flip -> replay agrees"
    c = V.census(synth_nets, case["ro"], case["rd"], oracle.YCBV_NEAR, oracle.YCBV_FAR, case["cot"], case["zf"], got,
                 thr=1e-7, margin=1.0)
    assert c["flipped_units_total"] == 1 and c["rays_with_flips"] == 1, c
    assert c["rays_above_thr"] >= 1 and c["unattributed"] == 0 and c["attributed"] == c["rays_above_thr"], c
    assert c["worst"][0]["ray"] == 2 and c["worst"][0]["flipped_units"] == 1


def test_flip_far_from_the_discontinuity_is_not_attributed(oracle, synth_nets, case):
    import vjp_census as V
    on_pre, on_av = _own_patterns(case)
    flipped, pre = _flip(case, on_pre, want_small=False)
    got = _kernel_like(oracle, synth_nets, case, flipped, on_av)
    c = V.census(synth_nets, case["ro"], case["rd"], oracle.YCBV_NEAR, oracle.YCBV_FAR, case["cot"], case["zf"], got, thr=1e-7)
    assert c["unattributed"] >= 1 and c["worst"][0]["off_cliff"], c


def test_fabricated_gradient_is_not_attributed(oracle, synth_nets, case):
    import vjp_census as V
    on_pre, on_av = _own_patterns(case)
    got = _kernel_like(oracle, synth_nets, case, on_pre, on_av)
    got["grad_d"] = got["grad_d"].copy()
    got["grad_d"][4] *= 1.01
    c = V.census(synth_nets, case["ro"], case["rd"], oracle.YCBV_NEAR, oracle.YCBV_FAR, case["cot"], case["zf"], got, thr=1e-4)
    assert c["rays_above_thr"] == 1 and c["unattributed"] == 1 and c["flagged_without_flips"] == 1, c


def test_wrong_network_backward_shows_per_point(oracle, synth_nets, case):
    import vjp_census as V
    on_pre, on_av = _own_patterns(case)
    got = _kernel_like(oracle, synth_nets, case, on_pre, on_av)
    got["grad_pts"] = got["grad_pts"].copy()
    smp = int(np.abs(got["grad_pts"][3]).max(-1).argmax())
    got["grad_pts"][3, smp] *= 1.001
    pp = V.per_point(synth_nets, case["ro"], case["rd"], case["zf"], got)
    assert pp["argmax_ray"] == 3 and pp["max"] > 5e-4, pp


def test_attribution_without_taps_finds_the_flipped_unit(oracle, synth_nets, case):
    """oracle/vjp_census.attribute_without_taps (for the x16 fp32 kernels, which have no debug taps): from the gradient
    alone it names the unit that was flipped -- and refuses a fabricated gradient."""
    import vjp_census as V
    on_pre, on_av = _own_patterns(case)
    flipped, pre = _flip(case, on_pre, want_small=True)
    got = _kernel_like(oracle, synth_nets, case, flipped, on_av)
    where = np.argwhere(flipped != on_pre)[0]                      # (ray 2, sample, layer 6, unit)
    a = V.attribute_without_taps(synth_nets, case["ro"], case["rd"], oracle.YCBV_NEAR, oracle.YCBV_FAR, case["cot"], case["zf"],
                                 got["grad_o"], got["grad_d"], rays=[2], thr=1e-6, margin=1.0, max_candidates=4000)
    # (margin 1.0 admits every unit; the candidate list is cut to the 4000 closest to zero -- the fabricated flip is the closest
    # unit of its sample's layer 6, far inside that)
    assert a[2]["flips"] is not None and a[2]["err_after"] <= 1e-6 < a[2]["err"], a[2]
    assert (int(where[1]), 6, int(where[3])) in [f[:3] for f in a[2]["flips"]], (a[2], where)
    bad_d = got["grad_d"].copy()
    bad_d[2] *= 1.01
    b = V.attribute_without_taps(synth_nets, case["ro"], case["rd"], oracle.YCBV_NEAR, oracle.YCBV_FAR, case["cot"], case["zf"],
                                 got["grad_o"], bad_d, rays=[2], thr=1e-6, margin=2e-5)
    assert b[2]["flips"] is None, b[2]
