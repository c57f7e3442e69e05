"""CPU oracle for the NeRF volumetric-render hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain numpy fp32, the algorithm of the reference's render path
(gyhandy/Neural-Sim-NeRF, files cited per function below; RN = optimization/utils/run_nerf_noscale.py,
RH = optimization/utils/run_nerf_helpers.py).  It exists so that tests can check the HIP path against
something that does not live in the product.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.  The product (`neural_sim_nerf_amd/`) never does.

Pinning: every function here is checked against golden vectors produced by importing and running the
reference itself (see oracle/gen_golden.py -> tests/golden/*.npz, tests/test_oracle_golden.py).

Numerics conventions (measured against torch 2.10 CPU, the reference's only runnable backend here):
  * everything is IEEE fp32, one rounding per op, NO fused multiply-add (torch evaluates op by op);
  * torch.cumprod / torch.cumsum on CPU fp32 accumulate sequentially in fp64 and round every prefix to
    fp32 (RN:376, RH:203) -> `_cumprod_f64`, `_cumsum_f64`;
  * torch.sum over a contiguous inner dim uses an 8-lane vectorised cascade (ATen SumKernel.cpp) whose
    association order is reproduced exactly by `_torch_sum_lastdim` -- needed because the pdf
    normaliser (RH:202) feeds the searchsorted indices that must be bit-exact;
  * torch.linspace(0,1,n) fp32 on CPU is NOT np.linspace (vectorised base + k*step with FMAs); the
    tables are constants of the path and are taken from host torch, as the reference does (RN:439);
  * sin/cos/exp/sigmoid come from numpy's fp32 libm here and from Sleef in torch: they agree to ~1 ulp,
    not bitwise -> the parity tests use a tolerance on everything downstream of them.
"""
import numpy as np

f32 = np.float32
f64 = np.float64

# ----------------------------------------------------------------------------------------------
# configuration of the path (reference: configs/nerf_param_ycbv_general.txt, NM:1232-1272 defaults)
# ----------------------------------------------------------------------------------------------
N_SAMPLES = 64          # CF:12
N_IMPORTANCE = 128      # CF:13
MULTIRES = 10           # NM:1268  -> 63 input channels
MULTIRES_VIEWS = 4      # NM:1270  -> 27 view channels
NET_DEPTH = 8           # NM:1232
NET_WIDTH = 256         # NM:1234
SKIP_AT = 4             # RN:269 skips=[4]
IN_CH = 3 + 3 * 2 * MULTIRES            # 63
IN_CH_VIEWS = 3 + 3 * 2 * MULTIRES_VIEWS  # 27

# camera of the YCB-V object-2 data (logs/nerfdata/nerf_traindata_info.json:2-5,35-51; LL:197-198)
YCBV_K = [[1333.3333740234375, 0.0, 195.4293212890625],
          [0.0, 1334.2196044921875, 200.63180541992188],
          [0.0, 0.0, 1.0]]
YCBV_NEAR = 0.8103964843749999 - 0.5
YCBV_FAR = 1.4297681884765627 + 0.5
YCBV_HW = 400


def scaled_K(scale):
    """Intrinsics for an (400/scale)x(400/scale) image, the way LL:185-192 rescales them."""
    K = [list(r) for r in YCBV_K]
    K[0] = [v / scale for v in K[0]]
    K[1] = [v / scale for v in K[1]]
    return K


# ----------------------------------------------------------------------------------------------
# torch-CPU arithmetic emulation helpers
# ----------------------------------------------------------------------------------------------
def torch_linspace01(n):
    """torch.linspace(0., 1., n) fp32 as computed on the CPU (RN:439, RH:208: the reference builds both
    tables on the host and then moves them).  ATen's vectorised kernel is not bit-equal to
    np.linspace (its lanes are base + k*step with fused multiply-adds), so the table is taken from the
    host's own torch -- the same call the reference makes -- and only falls back to a closed form (equal
    to ~93 % of the entries, 1 ulp off elsewhere) when torch is not importable."""
    try:
        import torch
        return torch.linspace(0., 1., steps=n).numpy().astype(f32)
    except ImportError:  # pragma: no cover
        step = f32(f32(1.0) / f32(n - 1))
        i = np.arange(n)
        lo = (i.astype(f32) * step).astype(f32)
        hi = (f32(1.0) - ((n - 1 - i).astype(f32) * step).astype(f32)).astype(f32)
        return np.where(i < n // 2, lo, hi).astype(f32)


def _add(a, b):
    return (a + b).astype(f32)


def _torch_sum_lastdim(x):
    """torch.sum(x, -1) for contiguous fp32 [N, n] -- ATen's vectorised inner cascade sum, 8 lanes,
    4-way ILP.  Exact association order (valid for n < 8*4*16 = 512)."""
    x = np.ascontiguousarray(x, dtype=f32)
    n_rows, n = x.shape
    W, ILP = 8, 4
    nvec = n // W
    size_ilp = nvec // ILP
    assert size_ilp < 16, "cascade levels beyond the first are not emulated"
    vec = lambda v: x[:, v * W:(v + 1) * W]
    zero = np.zeros((n_rows, W), f32)
    ps = [zero.copy() for _ in range(ILP)]
    for i in range(size_ilp):
        for k in range(ILP):
            ps[k] = _add(ps[k], vec(i * ILP + k))
    for v in range(size_ilp * ILP, nvec):
        ps[0] = _add(ps[0], vec(v))
    for k in range(1, ILP):
        ps[0] = _add(ps[0], ps[k])
    total = np.zeros(n_rows, f32)
    for k in range(nvec * W, n):
        total = _add(total, x[:, k])
    for k in range(W):
        total = _add(total, ps[0][:, k])
    return total


def _cumsum_f64(x):
    acc = np.zeros(x.shape[:-1], f64)
    out = np.empty_like(x, dtype=f32)
    for k in range(x.shape[-1]):
        acc = acc + x[..., k].astype(f64)
        out[..., k] = acc.astype(f32)
    return out


def _cumprod_f64(x):
    acc = np.ones(x.shape[:-1], f64)
    out = np.empty_like(x, dtype=f32)
    for k in range(x.shape[-1]):
        acc = acc * x[..., k].astype(f64)
        out[..., k] = acc.astype(f32)
    return out


# ----------------------------------------------------------------------------------------------
# synthetic weights (no pretrained YCB-V checkpoints are available: reference .gitignore:7)
# ----------------------------------------------------------------------------------------------
LAYER_SHAPES = (
    [("pts_linears.0", NET_WIDTH, IN_CH)]
    + [("pts_linears.%d" % i, NET_WIDTH, NET_WIDTH + (IN_CH if i == SKIP_AT + 1 else 0))
       for i in range(1, NET_DEPTH)]
    + [("feature_linear", NET_WIDTH, NET_WIDTH),
       ("alpha_linear", 1, NET_WIDTH),
       ("views_linears.0", NET_WIDTH // 2, NET_WIDTH + IN_CH_VIEWS),
       ("rgb_linear", 3, NET_WIDTH // 2)]
)
N_PARAMS = sum(o * i + o for _, o, i in LAYER_SHAPES)   # 595844 (RH:70-122)
MACS_PER_POINT = sum(o * i for _, o, i in LAYER_SHAPES)  # 593408


def synth_weights(seed, fine_of=None):
    """Deterministic non-degenerate weights with the state_dict names/shapes of RH:70-122.

    nn.Linear-style U(-1/sqrt(in), 1/sqrt(in)) init, then: trunk weights x1.6, alpha weight x50,
    alpha bias -0.5 (so rays carry a mix of empty and opaque space: acc in (0.4,1), non-trivial z_std).
    `fine_of`: derive a *different but consistent* fine net from a coarse one (x(1+0.05 N(0,1)))."""
    rng = np.random.RandomState(seed)
    sd = {}
    if fine_of is not None:
        for k, v in fine_of.items():
            sd[k] = (v * (1.0 + 0.05 * rng.standard_normal(v.shape))).astype(f32)
        return sd
    for name, o, i in LAYER_SHAPES:
        bound = 1.0 / np.sqrt(i)
        w = rng.uniform(-bound, bound, size=(o, i))
        b = rng.uniform(-bound, bound, size=(o,))
        if name.startswith("pts_linears"):
            w = w * 1.6
        if name == "alpha_linear":
            w = w * 50.0
            b = b * 0.0 - 0.5
        sd[name + ".weight"] = w.astype(f32)
        sd[name + ".bias"] = b.astype(f32)
    return sd


def synth_weights_noviews(seed, fine_of=None, output_ch=5):
    """The use_viewdirs=False network of RH:70-122 (create_nerf RN:267: output_ch = 5 with a fine pass) from the recipe
    above: the same trunk, `output_linear` = three colour rows + the density row of synth_weights (+ an unused fifth
    row), and the reference's unused `views_linears.0` of shape [128, 0 + 256]."""
    if fine_of is not None:
        rng = np.random.RandomState(seed)
        return {k: (v * (1.0 + 0.05 * rng.standard_normal(v.shape))).astype(f32) for k, v in fine_of.items()}
    base = synth_weights(seed)
    rng = np.random.RandomState(seed + 77)
    sd = {k: v for k, v in base.items() if k.startswith("pts_linears")}
    bound = 1.0 / np.sqrt(NET_WIDTH)
    w = rng.uniform(-bound, bound, size=(output_ch, NET_WIDTH)) * 4.0
    b = rng.uniform(-bound, bound, size=(output_ch,))
    w[3], b[3] = base["alpha_linear.weight"][0], base["alpha_linear.bias"][0]
    sd["output_linear.weight"], sd["output_linear.bias"] = w.astype(f32), b.astype(f32)
    sd["views_linears.0.weight"] = rng.uniform(-bound, bound, size=(NET_WIDTH // 2, NET_WIDTH)).astype(f32)
    sd["views_linears.0.bias"] = rng.uniform(-bound, bound, size=(NET_WIDTH // 2,)).astype(f32)
    return sd


# ----------------------------------------------------------------------------------------------
# the path
# ----------------------------------------------------------------------------------------------
def get_rays(H, W, K, c2w):
    """RH:156-165.  K is the nested python-float list the reference reads from JSON (LL:177); the
    scalars enter fp32 tensor ops, i.e. are rounded to fp32 first.  Returns rays_o, rays_d [H,W,3]."""
    c2w = np.asarray(c2w, dtype=f32)
    col = np.arange(W, dtype=f32)[None, :].repeat(H, 0)   # "i" after .t(): x / column index
    row = np.arange(H, dtype=f32)[:, None].repeat(W, 1)   # "j": y / row index
    dx = ((col - f32(K[0][2])) / f32(K[0][0])).astype(f32)
    dy = (-((row - f32(K[1][2])) / f32(K[1][1]))).astype(f32)
    dz = -np.ones_like(dx)
    dirs = np.stack([dx, dy, dz], -1)
    prod = (dirs[..., None, :] * c2w[:3, :3]).astype(f32)              # [H,W,3(a),3(k)]
    rays_d = _add(_add(prod[..., 0], prod[..., 1]), prod[..., 2])       # sum over k, in order
    rays_o = np.broadcast_to(c2w[:3, 3], rays_d.shape).astype(f32)
    return rays_o, rays_d


def normalize_dirs(rays_d):
    """RN:97: viewdirs = d / ||d||  (torch.norm: sqrt of the sequential fp32 sum of squares)."""
    d = rays_d.astype(f32)
    sq = (d * d).astype(f32)
    n = np.sqrt(_add(_add(sq[..., 0], sq[..., 1]), sq[..., 2])).astype(f32)
    return (d / n[..., None]).astype(f32)


def dir_norm(rays_d):
    d = rays_d.astype(f32)
    sq = (d * d).astype(f32)
    return np.sqrt(_add(_add(sq[..., 0], sq[..., 1]), sq[..., 2])).astype(f32)


def embed(x, n_freqs):
    """RH:18-48: [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)], 3 channels each."""
    x = x.astype(f32)
    out = [x]
    for l in range(n_freqs):
        xs = (x * f32(2.0 ** l)).astype(f32)
        out.append(np.sin(xs).astype(f32))
        out.append(np.cos(xs).astype(f32))
    return np.concatenate(out, -1)


def net_shape(sd):
    """(D, W, input_ch, input_ch_views, skips, use_viewdirs) of a state dict with the parameter names of RH:70-97, read off
    the weight shapes: layer i + 1 takes W + input_ch inputs exactly when i is a skip (RH:82-83)."""
    D = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("pts_linears."))
    W, input_ch = sd["pts_linears.0.weight"].shape
    skips = [i for i in range(D - 1) if sd["pts_linears.%d.weight" % (i + 1)].shape[1] == W + input_ch]
    use_viewdirs = "output_linear.weight" not in sd
    input_ch_views = sd["views_linears.0.weight"].shape[1] - W
    return D, W, input_ch, input_ch_views, skips, use_viewdirs


def mlp(sd, x_embedded, keep=None, all_rows=False, views=None):
    """RH:99-122 for the network the state dict describes (net_shape): x_embedded [P, 63 + 27] = the FULL encodings
    (10 / 4 frequencies) -> [P, 4] = (rgb logits, sigma).  A network built for fewer frequencies reads the leading
    3 + 6 L columns of each (the encoder emits its bands in increasing order, RH:35-48); a state dict with `output_linear`
    is the use_viewdirs=False network (RH:95-96, RH:119-120): outputs = output_linear(h), directions ignored.
    views given: x_embedded is the position encoding alone ([P, input_ch], any number of frequencies) and views the direction
    encoding [P, input_ch_views]."""
    lin = lambda name, h: (h @ sd[name + ".weight"].T + sd[name + ".bias"]).astype(f32)
    D, W, input_ch, input_ch_views, skips, use_viewdirs = net_shape(sd)
    if views is None:
        pts, views = x_embedded[:, :input_ch], x_embedded[:, IN_CH:IN_CH + max(input_ch_views, 0)]
    else:
        pts = x_embedded
    h = pts
    for i in range(D):
        h = np.maximum(lin("pts_linears.%d" % i, h), f32(0))
        if keep is not None:
            keep["h%d" % i] = h
        if i in skips:
            h = np.concatenate([pts, h], -1)
    if not use_viewdirs:                                  # RH:119-120: every row of output_linear (5 when N_importance > 0,
        out = lin("output_linear", h)                     # RN:267); render_rays reads the first four (RN:363-374)
        return out if all_rows else out[:, :4]
    alpha = lin("alpha_linear", h)
    feature = lin("feature_linear", h)
    hv = np.maximum(lin("views_linears.0", np.concatenate([feature, views], -1)), f32(0))
    rgb = lin("rgb_linear", hv)
    return np.concatenate([rgb, alpha], -1)


def synth_weights_shape(seed, D, W, L, Lv, skips, use_viewdirs=True, output_ch=5):
    """Seeded weights for a NeRF of another shape (RH:70-97) in the spirit of synth_weights: nn.Linear-style init, trunk
    x 1.6 * sqrt(256 / W) (keeps the activations of a narrower net alive), density row x 50 * 256 / W, density bias -0.5."""
    rng = np.random.RandomState(seed)
    in_ch, in_v = 3 + 6 * L, (3 + 6 * Lv if use_viewdirs else 0)
    sd = {}

    def lin(name, o, i, scale=1.0):
        bound = 1.0 / np.sqrt(i)
        sd[name + ".weight"] = (rng.uniform(-bound, bound, size=(o, i)) * scale).astype(f32)
        sd[name + ".bias"] = rng.uniform(-bound, bound, size=(o,)).astype(f32)
    g = 1.6 * np.sqrt(256.0 / W)
    lin("pts_linears.0", W, in_ch, g)
    for i in range(D - 1):
        lin("pts_linears.%d" % (i + 1), W, W + in_ch if i in skips else W, g)
    lin("views_linears.0", W // 2, in_v + W)
    if use_viewdirs:
        lin("feature_linear", W, W)
        lin("alpha_linear", 1, W, 50.0 * 256.0 / W)
        sd["alpha_linear.bias"] = np.full(1, -0.5, f32)
        lin("rgb_linear", 3, W // 2)
    else:
        lin("output_linear", output_ch, W, 4.0)
        sd["output_linear.weight"][3] *= f32(12.5 * 256.0 / W)
        sd["output_linear.bias"][3] = f32(-0.5)
    return sd


_BACKEND = "numpy"


def set_backend(name):
    """"numpy" (default, what the parity tests use) or "torch": the embedding + MLP of run_network executed with
    multi-threaded torch CPU ops -- the closest stand-in for the reference's own CPU PyTorch path when the
    oracle is timed as bench.py's cpu_baseline.  Everything else (compositing, resampling) stays numpy."""
    global _BACKEND
    assert name in ("numpy", "torch")
    _BACKEND = name


def _run_network_torch(sd, pts, viewdirs):
    import torch
    N, S, _ = pts.shape
    with torch.no_grad():
        t = lambda k: torch.from_numpy(sd[k])
        x = torch.from_numpy(np.ascontiguousarray(pts.reshape(-1, 3)))
        d = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(viewdirs[:, None, :], pts.shape).reshape(-1, 3)))

        def enc(v, L):
            fr = 2.0 ** torch.arange(L, dtype=torch.float32)
            a = v[:, None, :] * fr[None, :, None]                                   # [P,L,3]
            return torch.cat([v, torch.stack([torch.sin(a), torch.cos(a)], 2).reshape(v.shape[0], -1)], -1)
        e, ed = enc(x, MULTIRES), enc(d, MULTIRES_VIEWS)
        h = e
        for i in range(NET_DEPTH):
            h = torch.relu_(torch.addmm(t("pts_linears.%d.bias" % i), h, t("pts_linears.%d.weight" % i).T))
            if i == SKIP_AT:
                h = torch.cat([e, h], -1)
        alpha = torch.addmm(t("alpha_linear.bias"), h, t("alpha_linear.weight").T)
        feat = torch.addmm(t("feature_linear.bias"), h, t("feature_linear.weight").T)
        hv = torch.relu_(torch.addmm(t("views_linears.0.bias"), torch.cat([feat, ed], -1), t("views_linears.0.weight").T))
        rgb = torch.addmm(t("rgb_linear.bias"), hv, t("rgb_linear.weight").T)
        return torch.cat([rgb, alpha], -1).numpy().reshape(N, S, 4)


def run_network(sd, pts, viewdirs):
    """RN:26-40: pts [N,S,3], viewdirs [N,3] -> raw [N,S,4]."""
    if _BACKEND == "torch":
        return _run_network_torch(sd, pts, viewdirs)
    N, S, _ = pts.shape
    _, _, input_ch, input_ch_views, _, _ = net_shape(sd)         # the network's own frequencies (get_embedder RH:51-66)
    e = embed(pts.reshape(-1, 3), (input_ch - 3) // 6)
    ed = embed(np.broadcast_to(viewdirs[:, None, :], pts.shape).reshape(-1, 3), max(input_ch_views - 3, 0) // 6)
    return mlp(sd, e, views=ed).reshape(N, S, 4)


def raw2outputs(raw, z_vals, rays_d, white_bkgd=False, noise=None):
    """RN:343-387.  noise [N,S] (or None): the term raw_noise_std * randn the reference adds to the density before the
    relu (RN:365-374), drawn by the caller.
    Returns rgb_map [N,3], disp_map [N], acc_map [N], weights [N,S], depth_map [N]."""
    raw = raw.astype(f32)
    z = z_vals.astype(f32)
    N, S = z.shape
    dists = np.concatenate([(z[:, 1:] - z[:, :-1]).astype(f32), np.full((N, 1), 1e10, f32)], -1)
    dists = (dists * dir_norm(rays_d)[:, None]).astype(f32)
    with np.errstate(over="ignore"):
        rgb = (f32(1) / (f32(1) + np.exp(-raw[..., :3]).astype(f32))).astype(f32)
        dens = raw[..., 3] if noise is None else _add(raw[..., 3], np.asarray(noise, f32))     # RN:374
        sig = np.maximum(dens, f32(0))
        alpha = (f32(1) - np.exp((-sig * dists).astype(f32)).astype(f32)).astype(f32)
    one_minus = _add((f32(1) - alpha).astype(f32), f32(1e-10))
    T = _cumprod_f64(np.concatenate([np.ones((N, 1), f32), one_minus], -1))[:, :-1]
    weights = (alpha * T).astype(f32)
    rgb_map = np.sum((weights[..., None] * rgb).astype(f32), -2, dtype=f32)
    depth_map = np.sum((weights * z).astype(f32), -1, dtype=f32)
    acc_map = np.sum(weights, -1, dtype=f32)
    with np.errstate(invalid="ignore", divide="ignore"):
        q = (depth_map / acc_map).astype(f32)
        disp_map = (f32(1) / np.where(np.isnan(q), q, np.maximum(f32(1e-10), q))).astype(f32)
    if white_bkgd:                                                        # RN:384-385
        rgb_map = _add(rgb_map, (f32(1) - acc_map[..., None]).astype(f32))
    return rgb_map, disp_map, acc_map, weights, depth_map


def sample_pdf(bins, weights, n_samples=N_IMPORTANCE, u=None):
    """RH:199-243.  u None: the deterministic branch (det=True because perturb==0, RN:474), u = linspace(0, 1, n);
    u [n] or [N,n]: the uniforms of the det=False branch (RH:211), drawn by the caller.
    bins [N,63] (z mid-points), weights [N,62] (coarse weights[1:-1]).
    Returns samples [N,n], inds int64 [N,n] (searchsorted right=True), cdf [N,63]."""
    w = _add(weights.astype(f32), f32(1e-5))
    pdf = (w / _torch_sum_lastdim(w)[:, None]).astype(f32)
    cdf = np.concatenate([np.zeros((w.shape[0], 1), f32), _cumsum_f64(pdf)], -1)
    if u is None:
        u = torch_linspace01(n_samples)
    u = np.broadcast_to(np.asarray(u, f32), (cdf.shape[0], np.shape(u)[-1]))
    nb = cdf.shape[-1]
    inds = (cdf[:, None, :] <= u[:, :, None]).sum(-1).astype(np.int64)      # first idx with cdf > u
    below = np.maximum(inds - 1, 0)
    above = np.minimum(inds, nb - 1)
    take = lambda a, i: np.take_along_axis(a, i, -1)
    c0, c1 = take(cdf, below), take(cdf, above)
    b0, b1 = take(bins.astype(f32), below), take(bins.astype(f32), above)
    denom = (c1 - c0).astype(f32)
    denom = np.where(denom < f32(1e-5), f32(1), denom)
    t = ((u - c0).astype(f32) / denom).astype(f32)
    samples = _add(b0, (t * (b1 - b0).astype(f32)).astype(f32))
    return samples, inds, cdf


def coarse_z(near, far, n=N_SAMPLES, lindisp=False):
    """RN:439-443: z = near*(1-t) + far*t, or 1/(1/near*(1-t) + 1/far*t) with lindisp, per ray (near/far [N])."""
    t = torch_linspace01(n)
    near = np.asarray(near, f32).reshape(-1, 1)
    far = np.asarray(far, f32).reshape(-1, 1)
    if lindisp:
        inv = _add(((f32(1) / near).astype(f32) * (f32(1) - t).astype(f32)).astype(f32),
                   ((f32(1) / far).astype(f32) * t).astype(f32))
        return (f32(1) / inv).astype(f32)
    return _add((near * (f32(1) - t).astype(f32)).astype(f32), (far * t).astype(f32))


def perturb_z(z, t_rand):
    """RN:447-459: stratified samples, one per interval between the mid-points; t_rand [N,S] in [0, 1) from the caller."""
    z = z.astype(f32)
    mids = (f32(0.5) * _add(z[:, 1:], z[:, :-1])).astype(f32)
    upper = np.concatenate([mids, z[:, -1:]], -1)
    lower = np.concatenate([z[:, :1], mids], -1)
    return _add(lower, ((upper - lower).astype(f32) * np.asarray(t_rand, f32)).astype(f32))


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """RH:168-186 in torch's fp32 arithmetic: python scalars (H, W, focal, near) combine in double and are rounded to
    fp32 when they meet a tensor."""
    o, d = rays_o.astype(f32), rays_d.astype(f32)
    near32 = f32(near)
    t = (-(_add(near32, o[..., 2])) / d[..., 2]).astype(f32)
    o = _add(o, (t[..., None] * d).astype(f32))
    cw = f32(-1.0 / (W / (2.0 * focal)))
    ch = f32(-1.0 / (H / (2.0 * focal)))
    two_near = f32(2.0 * near)
    o0 = ((cw * o[..., 0]).astype(f32) / o[..., 2]).astype(f32)
    o1 = ((ch * o[..., 1]).astype(f32) / o[..., 2]).astype(f32)
    o2 = _add(f32(1), (two_near / o[..., 2]).astype(f32))
    d0 = (cw * ((d[..., 0] / d[..., 2]).astype(f32) - (o[..., 0] / o[..., 2]).astype(f32)).astype(f32)).astype(f32)
    d1 = (ch * ((d[..., 1] / d[..., 2]).astype(f32) - (o[..., 1] / o[..., 2]).astype(f32)).astype(f32)).astype(f32)
    d2 = (f32(-2.0 * near) / o[..., 2]).astype(f32)
    return np.stack([o0, o1, o2], -1), np.stack([d0, d1, d2], -1)


def ndc_rays_vjp(H, W, focal, near, rays_o, rays_d, g_o, g_d):
    """Input-side VJP of ndc_rays in float64: (dL/d o', dL/d d') -> (dL/d rays_o, dL/d rays_d)."""
    o, d = rays_o.astype(f64), rays_d.astype(f64)
    g_o, g_d = g_o.astype(f64), g_d.astype(f64)
    cw, ch = -1.0 / (W / (2.0 * focal)), -1.0 / (H / (2.0 * focal))
    t = -(near + o[..., 2]) / d[..., 2]
    s = o + t[..., None] * d                                   # shifted origin
    sx, sy, sz = s[..., 0], s[..., 1], s[..., 2]
    dx, dy, dz = d[..., 0], d[..., 1], d[..., 2]
    # outputs: o0 = cw sx/sz, o1 = ch sy/sz, o2 = 1 + 2 near/sz, d0 = cw (dx/dz - sx/sz), d1 = ch (dy/dz - sy/sz), d2 = -2 near/sz
    gs = np.zeros_like(s)
    gd = np.zeros_like(d)
    gs[..., 0] = (g_o[..., 0] - g_d[..., 0]) * cw / sz
    gs[..., 1] = (g_o[..., 1] - g_d[..., 1]) * ch / sz
    gs[..., 2] = (-(g_o[..., 0] - g_d[..., 0]) * cw * sx - (g_o[..., 1] - g_d[..., 1]) * ch * sy
                  - 2.0 * near * g_o[..., 2] + 2.0 * near * g_d[..., 2]) / (sz * sz)
    gd[..., 0] = g_d[..., 0] * cw / dz
    gd[..., 1] = g_d[..., 1] * ch / dz
    gd[..., 2] = -(g_d[..., 0] * cw * dx + g_d[..., 1] * ch * dy) / (dz * dz)
    # s = o + t d, t = -(near + oz)/dz
    gt = (gs * d).sum(-1)
    go = gs.copy()
    go[..., 2] += gt * (-1.0 / dz)
    gd = gd + gs * t[..., None]
    gd[..., 2] += gt * (near + o[..., 2]) / (dz * dz)
    return go.astype(f32), gd.astype(f32)


def render_rays(sd_coarse, sd_fine, rays_o, rays_d, viewdirs, near, far,
                n_samples=N_SAMPLES, n_importance=N_IMPORTANCE, extras=False, white_bkgd=False, lindisp=False,
                t_rand=None, u=None, noise0=None, noise1=None):
    """RN:390-501.  The random draws of the stochastic options are the caller's: t_rand [N,S] (perturb > 0, RN:447-459),
    u [N,n_importance] (sample_pdf with det=False, RH:211), noise0 [N,S] / noise1 [N,S+n_importance] (raw_noise_std *
    randn of the coarse / fine raw2outputs, RN:365-372); None = the deterministic path (perturb=0, raw_noise_std=0).
    rays_o/rays_d/viewdirs [N,3], near/far scalars or [N]."""
    rays_o, rays_d, viewdirs = (a.astype(f32) for a in (rays_o, rays_d, viewdirs))
    N = rays_o.shape[0]
    near = np.broadcast_to(np.asarray(near, f32), (N,))
    far = np.broadcast_to(np.asarray(far, f32), (N,))
    z = coarse_z(near, far, n_samples, lindisp)
    if t_rand is not None:
        z = perturb_z(z, t_rand)
    pts = _add(rays_o[:, None, :], (rays_d[:, None, :] * z[:, :, None]).astype(f32))
    raw = raw0 = run_network(sd_coarse, pts, viewdirs)
    rgb_map, disp_map, acc_map, weights, _ = raw2outputs(raw, z, rays_d, white_bkgd, noise0)
    ret = {}
    if n_importance > 0:
        ret.update(rgb0=rgb_map, disp0=disp_map, acc0=acc_map)
        z_mid = (f32(0.5) * _add(z[:, 1:], z[:, :-1])).astype(f32)
        z_samples, inds, cdf = sample_pdf(z_mid, weights[:, 1:-1], n_importance, u)
        z_fine = np.sort(np.concatenate([z, z_samples], -1), -1)
        pts = _add(rays_o[:, None, :], (rays_d[:, None, :] * z_fine[:, :, None]).astype(f32))
        raw = run_network(sd_fine if sd_fine is not None else sd_coarse, pts, viewdirs)
        rgb_map, disp_map, acc_map, weights_f, _ = raw2outputs(raw, z_fine, rays_d, white_bkgd, noise1)
        ret["z_std"] = np.std(z_samples.astype(f64), -1).astype(f32)   # RN:495 (unbiased=False)
        if extras:
            ret.update(z_samples=z_samples, inds=inds, cdf=cdf, z_fine=z_fine, weights0=weights,
                       weights=weights_f, raw=raw, raw0=raw0, z_coarse=z)
    elif extras:
        ret.update(raw0=raw0, weights0=weights, z_coarse=z)
    ret.update(rgb_map=rgb_map, disp_map=disp_map, acc_map=acc_map)
    return ret


def render(sd_coarse, sd_fine, H, W, K, c2w=None, rays=None, near=0.0, far=1.0,
           n_samples=N_SAMPLES, n_importance=N_IMPORTANCE, chunk=4096, extras=False, white_bkgd=False, lindisp=False,
           ndc=False, c2w_staticcam=None, randoms=None, use_viewdirs=True):
    """RN:58-123 with use_viewdirs=True.  ndc: RN:101-103 (the caller passes near=0, far=1 as the reference's callers
    do); c2w_staticcam: RN:91-96 (view directions from c2w, rays from the static camera; ignored with use_viewdirs=False,
    as upstream); randoms: dict with any of
    t_rand, u, noise0, noise1 for ALL rays (see render_rays), sliced per chunk here.
    Returns dict of [H,W,...] (c2w form) or [N,...]."""
    if c2w is not None:
        rays_o, rays_d = get_rays(H, W, K, c2w)
    else:
        rays_o, rays_d = rays
    viewdirs = normalize_dirs(rays_d.reshape(-1, 3).astype(f32))              # RN:89-98: from the rays BEFORE the next two
    if c2w_staticcam is not None and use_viewdirs:                            # RN:91-96 sits inside `if use_viewdirs:`
        rays_o, rays_d = get_rays(H, W, K, c2w_staticcam)
    sh = rays_d.shape[:-1]
    if ndc:
        rays_o, rays_d = ndc_rays(H, W, K[0][0], 1.0, rays_o, rays_d)
    rays_o = rays_o.reshape(-1, 3).astype(f32)
    rays_d = rays_d.reshape(-1, 3).astype(f32)
    randoms = randoms or {}
    outs = []
    for i in range(0, rays_o.shape[0], chunk):
        s = slice(i, i + chunk)
        rnd = {k: np.asarray(v).reshape(rays_o.shape[0], -1)[s] for k, v in randoms.items() if v is not None}
        outs.append(render_rays(sd_coarse, sd_fine, rays_o[s], rays_d[s], viewdirs[s], near, far,
                                n_samples, n_importance, extras, white_bkgd, lindisp, **rnd))
    ret = {k: np.concatenate([o[k] for o in outs], 0) for k in outs[0]}
    return {k: v.reshape(sh + v.shape[1:]) for k, v in ret.items()}


def to8b(x):
    """RH:14 (truncating conversion)."""
    return (255 * np.clip(x, 0, 1)).astype(np.uint8)


def psnr(a, b):
    """RH:12-13 convention."""
    mse = np.mean((a.astype(f64) - b.astype(f64)) ** 2)
    return float("inf") if mse == 0 else float(-10.0 * np.log10(mse))


# ----------------------------------------------------------------------------------------------
# poses (LL:89-94 pose_spherical_nograd; LL:250-301 sampling is replaced by a seeded draw)
# ----------------------------------------------------------------------------------------------
def pose_spherical(theta_deg, phi_deg, radius):
    """LL:89-94 pose_spherical_nograd, fp32 4x4 camera-to-world."""
    th = theta_deg / 180.0 * np.pi
    ph = phi_deg / 180.0 * np.pi
    trans = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]], f32)
    rphi = np.array([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0],
                     [0, np.sin(ph), np.cos(ph), 0], [0, 0, 0, 1]], f32)
    rth = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0],
                    [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]], f32)
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], f32)
    return (flip @ (rth @ (rphi @ trans))).astype(f32)


def sweep_poses(n_views, seed=0):
    """Synthetic stand-in for the Gumbel-softmax pose sweep (LL:250-301, NM:1164-1165): theta~U(85,95),
    phi drawn around the dominant 45-degree bin, radius 1.01 (LL:292-293)."""
    rng = np.random.RandomState(seed)
    poses = []
    for _ in range(n_views):
        theta = rng.uniform(85, 95)
        phi = rng.uniform(0, 360)
        poses.append(pose_spherical(theta, phi - 180.0, 1.01))
    return np.stack(poses, 0)


# ----------------------------------------------------------------------------------------------
# network backward (inputs only)
# ----------------------------------------------------------------------------------------------
def _network_forward64(sd, pts, dirs):
    """fp64 forward of RH:99-122 (any shape: net_shape) on fp32 encodings, keeping what the backward needs."""
    W = lambda k: sd[k + ".weight"].astype(f64)
    B = lambda k: sd[k + ".bias"].astype(f64)
    D, _, input_ch, input_ch_views, skips, use_viewdirs = net_shape(sd)
    e_p = embed(pts, (input_ch - 3) // 6).astype(f64)
    pre = []
    h = e_p
    for i in range(D):
        a = h @ W("pts_linears.%d" % i).T + B("pts_linears.%d" % i)
        pre.append(a)
        h = np.maximum(a, 0)
        if i in skips:
            h = np.concatenate([e_p, h], -1)
    if not use_viewdirs:                                # RH:119-120
        out = h @ W("output_linear").T + B("output_linear")
        return dict(pre=pre, av=None, sigma=out[:, 3], rgb_raw=out[:, :3])
    e_d = embed(dirs, (input_ch_views - 3) // 6).astype(f64)
    sigma = (h @ W("alpha_linear").T + B("alpha_linear"))[:, 0]
    feat = h @ W("feature_linear").T + B("feature_linear")
    av = np.concatenate([feat, e_d], -1) @ W("views_linears.0").T + B("views_linears.0")
    rgb_raw = np.maximum(av, 0) @ W("rgb_linear").T + B("rgb_linear")
    return dict(pre=pre, av=av, sigma=sigma, rgb_raw=rgb_raw)


def _embed_bwd(x, G, n_freqs):
    x = x.astype(f64)
    out = G[:, :3].copy()
    for l in range(n_freqs):
        fr = 2.0 ** l
        gs, gc = G[:, 3 + 6 * l:6 + 6 * l], G[:, 6 + 6 * l:9 + 6 * l]
        out += fr * (gs * np.cos(x * fr) - gc * np.sin(x * fr))
    return out


def network_vjp(sd, pts, dirs, g_raw, fwd=None, relu_on=None):
    """Input-side VJP of run_network: g_raw [P,4] = dL/d(rgb logits, sigma) -> dL/d pts [P,3], dL/d dirs [P,3]
    (weights are constants).  Manual backprop of RH:99-122 in float64, for a network of any shape (net_shape).
    relu_on (counterfactual replay, oracle/vjp_census.py): dict(pre=[D x [P,W] bool], av=[P,W/2] bool) -- the relu
    patterns to apply INSTEAD of the forward's own (pre > 0), i.e. relu' as another evaluation of the forward saw it."""
    W = lambda k: sd[k + ".weight"].astype(f64)
    D, width, input_ch, input_ch_views, skips, use_viewdirs = net_shape(sd)
    if fwd is None:
        fwd = _network_forward64(sd, pts, dirs)
    pre, av = fwd["pre"], fwd["av"]
    on = [p > 0 for p in pre] if relu_on is None else relu_on["pre"]
    g_raw = g_raw.astype(f64)
    if av is None:                                      # use_viewdirs=False: outputs = output_linear(h)
        G_h = g_raw @ W("output_linear")[:4]
        G_ed = np.zeros((pts.shape[0], 3), f64)
    else:
        G_av = (g_raw[:, :3] @ W("rgb_linear")) * ((av > 0) if relu_on is None else relu_on["av"])
        G_cat = G_av @ W("views_linears.0")
        G_feat, G_ed = G_cat[:, :width], G_cat[:, width:]
        G_h = G_feat @ W("feature_linear") + g_raw[:, 3:4] @ W("alpha_linear")
    G_ep = np.zeros((pts.shape[0], input_ch), f64)
    for i in reversed(range(D)):
        if i in skips:                                  # h after layer i was cat[e_p, relu(pre_i)] (RH:105-106)
            G_ep += G_h[:, :input_ch]
            G_h = G_h[:, input_ch:]
        G_h = (G_h * on[i]) @ W("pts_linears.%d" % i)
    G_ep += G_h
    return (_embed_bwd(pts, G_ep, (input_ch - 3) // 6),
            _embed_bwd(dirs, G_ed, (input_ch_views - 3) // 6 if use_viewdirs else 0))


# ----------------------------------------------------------------------------------------------
# input-side VJP of the fine render (what render_path_grad needs, RN:168-178)
# ----------------------------------------------------------------------------------------------
def render_rays_vjp(sd_coarse, sd_fine, rays_o, rays_d, near, far, grad_rgb,
                    n_samples=N_SAMPLES, n_importance=N_IMPORTANCE, z_fine=None, white_bkgd=False, lindisp=False,
                    viewdirs=None, noise1=None, randoms=None, relu_on=None, sigma_on=None, parts=None):
    """d(sum(rgb_map * grad_rgb)) / d(rays_o, rays_d), the quantity torch.autograd.grad(rgb_p, batch_rays,
    grad_outputs=patch_grad_E) returns at RN:177.  Network weights are constants and z_samples is detached
    (RN:475), so gradient reaches the rays only through the FINE pass: pts = o + d*z (RN:478), the view
    direction d/|d| (RN:97) and dists*|d| (RN:361).  Manual backprop in float64 on the fp32 forward's
    activations pattern.  Returns grad_o [N,3], grad_d [N,3] (float32) and the forward rgb_map.
    viewdirs [N,3]: view directions that are an input of their own (c2w_staticcam RN:91-96, ndc RN:101-103) -- the
    return value then has a fourth entry, dL/d viewdirs, and grad_d holds no view-direction term.  noise1 [N,S]: the fine
    pass's density noise (RN:365-374); randoms: the forward's draws when z_fine is to be recomputed here."""
    rays_o = rays_o.astype(f32)
    rays_d = rays_d.astype(f32)
    N = rays_o.shape[0]
    sd = sd_fine if sd_fine is not None else sd_coarse
    vd = normalize_dirs(rays_d) if viewdirs is None else viewdirs.astype(f32)
    if randoms and noise1 is None:
        noise1 = randoms.get("noise1")
    if z_fine is None:
        z_fine = render_rays(sd_coarse, sd_fine, rays_o, rays_d, vd, near, far, n_samples, n_importance,
                             extras=True, white_bkgd=white_bkgd, lindisp=lindisp, **(randoms or {}))["z_fine"]
    S = z_fine.shape[1]
    z = z_fine.astype(f32)
    pts = _add(rays_o[:, None, :], (rays_d[:, None, :] * z[:, :, None]).astype(f32)).reshape(-1, 3)
    dirs = np.broadcast_to(vd[:, None, :], (N, S, 3)).reshape(-1, 3)
    # ---- forward, keeping pre-activations ----
    fwd = _network_forward64(sd, pts, dirs)
    sigma, rgb_raw = fwd["sigma"], fwd["rgb_raw"]
    # ---- compositing forward (float64) ----
    sigma = sigma.reshape(N, S)
    if noise1 is not None:
        sigma = _add(sigma.astype(f32), np.asarray(noise1, f32)).astype(f64)     # RN:374: the relu sees raw + noise
    c = 1.0 / (1.0 + np.exp(-rgb_raw.reshape(N, S, 3)))
    nrm = dir_norm(rays_d).astype(f64)
    dz = np.concatenate([np.diff(z.astype(f64), axis=1), np.full((N, 1), 1e10)], 1)
    delta = dz * nrm[:, None]
    rs = np.maximum(sigma, 0)
    alpha = 1.0 - np.exp(-rs * delta)
    om = 1.0 - alpha + 1e-10
    T = np.cumprod(np.concatenate([np.ones((N, 1)), om], 1), 1)[:, :-1]
    w = alpha * T
    rgb_map = (w[..., None] * c).sum(1)
    if white_bkgd:
        rgb_map = rgb_map + (1.0 - w.sum(1))[:, None]                   # RN:384-385
    # ---- compositing backward ----
    g = grad_rgb.astype(f64)
    A = (c * g[:, None, :]).sum(-1)                                     # dL/dw_i
    if white_bkgd:
        A = A - g.sum(-1)[:, None]                                      # d(1 - acc)/dw_i = -1 per channel
    Aw = A * w
    suffix = np.concatenate([np.cumsum(Aw[:, ::-1], 1)[:, ::-1][:, 1:], np.zeros((N, 1))], 1)   # sum_{k>i}
    d_alpha = A * T - suffix / om
    d_sigma = d_alpha * delta * (1.0 - alpha) * ((sigma > 0) if sigma_on is None else sigma_on)
    d_delta = d_alpha * rs * (1.0 - alpha)
    d_nrm = (d_delta * dz).sum(1)
    d_rgb_raw = (w[..., None] * g[:, None, :]) * c * (1.0 - c)
    # ---- network backward (inputs only) ----
    g_raw = np.concatenate([d_rgb_raw.reshape(-1, 3), d_sigma.reshape(-1, 1)], -1)
    G_pts, G_dirs = network_vjp(sd, pts, dirs, g_raw, fwd, relu_on)
    if parts is not None:                               # the intermediates oracle/vjp_census.py attributes with
        parts.update(fwd=fwd, pts=pts, dirs=dirs, g_raw=g_raw.reshape(N, S, 4), sigma=sigma,
                     g_pts=G_pts.reshape(N, S, 3), g_dirs=G_dirs.reshape(N, S, 3))
    G_pts = G_pts.reshape(N, S, 3)
    G_v = G_dirs.reshape(N, S, 3).sum(1)
    grad_o = G_pts.sum(1)
    v = normalize_dirs(rays_d).astype(f64)
    grad_d = (G_pts * z.astype(f64)[:, :, None]).sum(1)
    grad_d += d_nrm[:, None] * v                                                 # d|d|/dd = d/|d|
    if viewdirs is not None:
        return grad_o.astype(f32), grad_d.astype(f32), rgb_map.astype(f32), G_v.astype(f32)
    grad_d += (G_v - v * (G_v * v).sum(-1, keepdims=True)) / nrm[:, None]        # d(d/|d|)
    return grad_o.astype(f32), grad_d.astype(f32), rgb_map.astype(f32)
