"""Weight packer: NeRF state_dict (RH:70-122 names) -> the layout the gfx950 kernel streams through LDS.

The kernel (csrc/nsr_kernels.hip) computes every layer as  H_out^T = W * H_in^T  on v_mfma_f32_32x32x2_f32
with the weights as the A operand.  One MFMA consumes, per lane l, A[i = l&31][k = l>>5]; the C/D fragment
holds, in lane l register r, output row (r&3) + 8*(r>>2) + 4*(l>>5) of column l&31.  Because layer L's C/D
registers are used *as they are* as layer L+1's B operands, k-step t of a 256-wide layer reads, in lane half
h, input feature

    kappa(t, h) = 32*(t>>4) + (t&3) + 8*((t&15)>>2) + 4*h            t = 0..127, h = 0..1

so the packer stores W[:, kappa(t, h)] where a plain GEMM would store W[:, 2t+h].  The position / direction
encodings (RH:18-48) are produced in-register in the order eps / eps_d below (sin in half 0, cos in half 1).

Stream layout (floats): 290 steps of 2048 floats (two steps = one 16 KiB LDS slab).  A step is 8 chunks of
[64 lanes][4] floats, read back by one ds_read_b128 per lane and chunk:
    8-block layers : step = k-quad tq, chunk = output block mo, float kk -> W[32*mo + (l&31)][col(4*tq+kk, l>>5)]
    views layer    : step s, chunk c -> k-quad 2*s + (c>>2), output block c&3
followed by the aux block (biases in C-fragment order, alpha/rgb heads), see nsr_device.h.
"""
import numpy as np

N_STEPS = 290
STEP_FLOATS = 2048
SLAB_FLOATS = 4096
STREAM_SLABS = 145
AUX_FLOATS = 3328
PACKED_FLOATS = STREAM_SLABS * SLAB_FLOATS + AUX_FLOATS

AUX_BIAS, AUX_BIAS_V, AUX_W_ALPHA, AUX_W_RGB, AUX_B_ALPHA, AUX_B_RGB = 0, 2304, 2432, 2688, 3072, 3073


def kappa(t, h):
    t = np.asarray(t)
    return 32 * (t >> 4) + (t & 3) + 8 * ((t & 15) >> 2) + 4 * h


def eps(t, h, n_freq):
    """Reference embedding column produced in encoding register t of lane half h (-1 = zero padding).
    Reference order (RH:39-48): [x y z | sin(2^0 .)(3) cos(2^0 .)(3) | sin(2^1 .) ...]."""
    n_trig = 3 * n_freq
    if t < n_trig:
        return 3 + 6 * (t // 3) + 3 * h + (t % 3)
    if t == n_trig:
        return 0 if h == 0 else 2
    if t == n_trig + 1:
        return 1 if h == 0 else -1
    return -1


def _cfrag_index(n_feat):
    """feature f -> position in the C-fragment-ordered vectors of the aux block."""
    f = np.arange(n_feat)
    mo, i = f >> 5, f & 31
    h, ri, rq = (i >> 2) & 1, i & 3, i >> 3
    return ((mo * 4 + rq) * 2 + h) * 4 + ri


def _pack_steps8(W, cols):
    """W [256, K]; cols [n_ksteps, 2] (reference column or -1) -> [n_ksteps/4, 8, 64, 4] floats."""
    n_k = cols.shape[0]
    assert W.shape[0] == 256 and n_k % 4 == 0
    Wp = np.concatenate([W, np.zeros((W.shape[0], 1), W.dtype)], 1)       # column -1 -> zeros
    lane = np.arange(64)
    i, h = lane & 31, lane >> 5
    out = np.empty((n_k // 4, 8, 64, 4), np.float32)
    for tq in range(n_k // 4):
        for kk in range(4):
            c = cols[4 * tq + kk][h]                                        # [64]
            for mo in range(8):
                out[tq, mo, :, kk] = Wp[32 * mo + i, c]
    return out


def _pack_steps_views(W, cols):
    """W [128, K]; cols [144, 2] -> [18 steps, 8 chunks, 64, 4]."""
    assert W.shape[0] == 128 and cols.shape[0] == 144
    Wp = np.concatenate([W, np.zeros((W.shape[0], 1), W.dtype)], 1)
    lane = np.arange(64)
    i, h = lane & 31, lane >> 5
    out = np.empty((18, 8, 64, 4), np.float32)
    for s in range(18):
        for c in range(8):
            tq, mo = 2 * s + (c >> 2), c & 3
            for kk in range(4):
                out[s, c, :, kk] = Wp[32 * mo + i, cols[4 * tq + kk][h]]
    return out


def pack_network(sd):
    """sd: mapping name -> array-like with the reference's state_dict keys (RH:82-97).  Returns float32
    [PACKED_FLOATS].  Raises on any shape that is not the 8x256 / skip-4 / use_viewdirs architecture."""
    g = lambda k: np.asarray(sd[k], dtype=np.float32)
    shapes = {"pts_linears.0.weight": (256, 63), "pts_linears.5.weight": (256, 319),
              "feature_linear.weight": (256, 256), "alpha_linear.weight": (1, 256),
              "views_linears.0.weight": (128, 283), "rgb_linear.weight": (3, 128)}
    for i in (1, 2, 3, 4, 6, 7):
        shapes["pts_linears.%d.weight" % i] = (256, 256)
    for k, shp in shapes.items():
        if k not in sd or tuple(np.shape(sd[k])) != shp:
            raise ValueError("pack_network: %s must have shape %s (8x256 NeRF with skip at 4 and view "
                             "directions, RH:70-122); got %s" % (k, shp, np.shape(sd[k]) if k in sd else None))
    t = np.arange(128)
    cols_main = np.stack([kappa(t, 0), kappa(t, 1)], 1)                                     # [128,2]
    cols_enc = np.array([[eps(tt, 0, 10), eps(tt, 1, 10)] for tt in range(32)])
    cols_dir = np.array([[eps(tt, 0, 4), eps(tt, 1, 4)] for tt in range(16)])
    steps = [_pack_steps8(g("pts_linears.0.weight"), cols_enc)]
    for i in range(1, 8):
        W = g("pts_linears.%d.weight" % i)
        if i == 5:                                   # cat([input_pts, h]) (RH:105): input columns first
            steps.append(_pack_steps8(W[:, :63], cols_enc))
            W = W[:, 63:]
        steps.append(_pack_steps8(W, cols_main))
    steps.append(_pack_steps8(g("feature_linear.weight"), cols_main))
    cols_views = np.concatenate([cols_main, np.where(cols_dir >= 0, cols_dir + 256, -1)], 0)  # cat([feature, views])
    steps.append(_pack_steps_views(g("views_linears.0.weight"), cols_views))
    stream = np.concatenate([s.reshape(-1) for s in steps])
    assert stream.size == N_STEPS * STEP_FLOATS == STREAM_SLABS * SLAB_FLOATS

    aux = np.zeros(AUX_FLOATS, np.float32)
    ci256, ci128 = _cfrag_index(256), _cfrag_index(128)
    for L in range(8):
        aux[AUX_BIAS + L * 256 + ci256] = g("pts_linears.%d.bias" % L)
    aux[AUX_BIAS + 8 * 256 + ci256] = g("feature_linear.bias")
    aux[AUX_BIAS_V + ci128] = g("views_linears.0.bias")
    wa = g("alpha_linear.weight")[0]
    tq, hh, kk = np.meshgrid(np.arange(32), np.arange(2), np.arange(4), indexing="ij")
    aux[AUX_W_ALPHA + ((tq * 2 + hh) * 4 + kk).ravel()] = wa[kappa(4 * tq + kk, hh).ravel()]
    wr = g("rgb_linear.weight")
    for c in range(3):
        aux[AUX_W_RGB + c * 128 + ci128] = wr[c]
    aux[AUX_B_ALPHA] = g("alpha_linear.bias")[0]
    aux[AUX_B_RGB:AUX_B_RGB + 3] = g("rgb_linear.bias")
    return np.concatenate([stream, aux]).astype(np.float32)


# ----------------------------------------------------------------------------------------------------------
# backward (input-gradient) stream: the same GEMM machinery with W^T as the A operand
# ----------------------------------------------------------------------------------------------------------
def _enc_row(i):
    """Row i (0..31) of an 'encoding' output block <-> (register tl, lane half h_e) of the encoding array:
    the C fragment puts row (r&3)+8(r>>2)+4h in register r of half h, so row i is register
    tl = (i&3) + 4*(i>>3) of half (i>>2)&1."""
    return (i & 3) + 4 * (i >> 3), (i >> 2) & 1


def _pack_T(M, n_mo, n_tq):
    """M [n_mo*32 rows, K] (already transposed: rows = the GEMM's outputs, columns = its k features in
    reference order); k-step t / half h reads column kappa(t, h).  Chunk order [tq][mo] -> [n_tq, n_mo, 64, 4]."""
    assert M.shape[0] == n_mo * 32
    lane = np.arange(64)
    i, h = lane & 31, lane >> 5
    out = np.empty((n_tq, n_mo, 64, 4), np.float32)
    for tq in range(n_tq):
        for kk in range(4):
            c = kappa(4 * tq + kk, h)
            for mo in range(n_mo):
                out[tq, mo, :, kk] = M[32 * mo + i, c]
    return out


def _enc_rows_T(W_cols, n_freq, n_blocks):
    """Rows of the transposed matrix that produce d/d(encoding register): W_cols [K_out, n_in] holds the
    layer's columns for the encoding inputs (reference order); returns [n_blocks*32, K_out]."""
    rows = np.zeros((n_blocks * 32, W_cols.shape[0]), np.float32)
    for b in range(n_blocks):
        for i in range(32):
            tl, he = _enc_row(i)
            col = eps(16 * b + tl, he, n_freq)
            if col >= 0:
                rows[32 * b + i] = W_cols[:, col]
    return rows


def pack_network_backward(sd):
    """Transposed stream of one network for the input-side VJP kernel (k_render_vjp).  Order: views^T encoding rows (1
    block x 16 quads) | views^T (8 blocks) | feature^T | L7^T | L6^T | L5^T encoding rows (2 blocks) | L5^T (8 blocks) |
    L4^T..L1^T | L0^T (2 encoding blocks): the encoding rows are their own segments, run before the 8-block GEMM of the same
    layer.  Returns float32 [STREAM_SLABS * SLAB_FLOATS] (the aux block of pack_network is shared)."""
    g = lambda k: np.asarray(sd[k], dtype=np.float32)
    segs = []
    Wv = g("views_linears.0.weight")                                     # [128, 256 + 27]
    segs.append(_pack_T(_enc_rows_T(Wv[:, 256:], 4, 1), 1, 16))
    segs.append(_pack_T(np.ascontiguousarray(Wv[:, :256].T), 8, 16))
    segs.append(_pack_T(g("feature_linear.weight").T, 8, 32))
    for l in (7, 6):
        segs.append(_pack_T(g("pts_linears.%d.weight" % l).T, 8, 32))
    W5 = g("pts_linears.5.weight")                                       # [256, 63 + 256], input columns first
    segs.append(_pack_T(_enc_rows_T(W5[:, :63], 10, 2), 2, 32))
    segs.append(_pack_T(np.ascontiguousarray(W5[:, 63:].T), 8, 32))
    for l in (4, 3, 2, 1):
        segs.append(_pack_T(g("pts_linears.%d.weight" % l).T, 8, 32))
    segs.append(_pack_T(_enc_rows_T(g("pts_linears.0.weight"), 10, 2), 2, 32))
    stream = np.concatenate([x.reshape(-1) for x in segs]).astype(np.float32)
    assert stream.size == STREAM_SLABS * SLAB_FLOATS
    return stream


# ----------------------------------------------------------------------------------------------------------
# "x16" forward layout: v_mfma_f32_16x16x4_f32, 16 points per wave, TWO workgroups per CU (k_render16)
# ----------------------------------------------------------------------------------------------------------
# One MFMA consumes, per lane l, A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; its C/D fragment holds, in
# register r of lane l, output row 4*(l>>4) + r of column l&15.  With the weights as A and 16-row output blocks,
# register (block mo, r) of lane group g = l>>4 holds feature 16*mo + 4*g + r -- which is also the feature k-step
# t = 4*mo + r reads in group g.  The K permutation of the 32x32 layout disappears: kappa16(t, g) = 16*(t>>2) + 4g +
# (t&3), biases and head weights stay in natural order.  A chunk is [64 lanes][4 k-steps] for one 16-row block; a
# 256-wide layer is 16 k-quads x 16 blocks = 256 chunks = 16 slabs, the stream is 145 slabs as before.
def kappa16(t, g):
    t = np.asarray(t)
    return 16 * (t >> 2) + 4 * g + (t & 3)


def eps16(t, g, n_freq):
    """Reference embedding column held in encoding register t of lane group g (-1 = padding).  The 6*n_freq sin/cos
    columns (reference order after the 3 identity columns: per frequency L, sin xyz then cos xyz) are dealt
    6*n_freq/4 per group; the next register holds the identity column g (x, y, z, pad).
      directions (n_freq = 4): group g holds frequency g: t = 3*sc + axis;
      positions (n_freq = 10): group g holds sin (g even) or cos (g odd) of octaves 5*(g >> 1) + t // 3, axis t % 3 --
        the kernel then needs two per-lane constants instead of fifteen (frequency, sin/cos, axis per register)."""
    per_group = 6 * n_freq // 4
    if t < per_group:
        if n_freq == 10:
            L, sc, ax = 5 * (g >> 1) + t // 3, g & 1, t % 3
            return 3 + 6 * L + 3 * sc + ax
        return 3 + per_group * g + t
    if t == per_group:
        return g if g < 3 else -1
    return -1


def _pack16(W, cols, n_mo):
    """W [16*n_mo, K]; cols [n_ksteps, 4 groups] -> [n_ksteps/4, n_mo, 64, 4]."""
    n_k = cols.shape[0]
    assert W.shape[0] == 16 * n_mo and n_k % 4 == 0
    Wp = np.concatenate([W, np.zeros((W.shape[0], 1), W.dtype)], 1)
    lane = np.arange(64)
    i, g = lane & 15, lane >> 4
    out = np.empty((n_k // 4, n_mo, 64, 4), np.float32)
    for tq in range(n_k // 4):
        for kk in range(4):
            c = cols[4 * tq + kk][g]
            for mo in range(n_mo):
                out[tq, mo, :, kk] = Wp[16 * mo + i, c]
    return out


def pack_network16(sd):
    """Forward stream + natural-order aux block for k_render16.  float32 [PACKED_FLOATS]."""
    g = lambda k: np.asarray(sd[k], dtype=np.float32)
    t = np.arange(64)
    cols_main = np.stack([kappa16(t, gg) for gg in range(4)], 1)                              # [64,4]
    cols_enc = np.array([[eps16(tt, gg, 10) for gg in range(4)] for tt in range(16)])
    cols_dir = np.array([[eps16(tt, gg, 4) for gg in range(4)] for tt in range(8)])
    segs = [_pack16(g("pts_linears.0.weight"), cols_enc, 16)]
    for i in range(1, 8):
        W = g("pts_linears.%d.weight" % i)
        if i == 5:
            segs.append(_pack16(W[:, :63], cols_enc, 16))
            W = W[:, 63:]
        segs.append(_pack16(W, cols_main, 16))
    segs.append(_pack16(g("feature_linear.weight"), cols_main, 16))
    cols_views = np.concatenate([cols_main, np.where(cols_dir >= 0, cols_dir + 256, -1)], 0)   # [72,4]
    segs.append(_pack16(g("views_linears.0.weight"), cols_views, 8))
    stream = np.concatenate([x.reshape(-1) for x in segs])
    assert stream.size == STREAM_SLABS * SLAB_FLOATS
    aux = np.zeros(AUX_FLOATS, np.float32)
    for L in range(8):
        aux[AUX_BIAS + L * 256:AUX_BIAS + (L + 1) * 256] = g("pts_linears.%d.bias" % L)
    aux[AUX_BIAS + 8 * 256:AUX_BIAS + 9 * 256] = g("feature_linear.bias")
    aux[AUX_BIAS_V:AUX_BIAS_V + 128] = g("views_linears.0.bias")
    aux[AUX_W_ALPHA:AUX_W_ALPHA + 256] = g("alpha_linear.weight")[0]
    aux[AUX_W_RGB:AUX_W_RGB + 384] = g("rgb_linear.weight").reshape(-1)
    aux[AUX_B_ALPHA] = g("alpha_linear.bias")[0]
    aux[AUX_B_RGB:AUX_B_RGB + 3] = g("rgb_linear.bias")
    return np.concatenate([stream, aux]).astype(np.float32)


# ----------------------------------------------------------------------------------------------------------
# x16 backward (input-gradient) stream for k_render_vjp16: the x16 machinery with W^T as the A operand
# ----------------------------------------------------------------------------------------------------------
def _enc_rows16(W_cols, n_freq, n_blocks):
    """Rows of a transposed matrix that produce d/d(encoding register): output block b, register r of lane group g is
    row 16b + 4g + r and must be the gradient w.r.t. encoding register t = 4b + r of that group, i.e. reference column
    eps16(t, g).  W_cols [K_out, n_in] (reference column order) -> [16 * n_blocks, K_out]."""
    rows = np.zeros((16 * n_blocks, W_cols.shape[0]), np.float32)
    for b in range(n_blocks):
        for gg in range(4):
            for r in range(4):
                col = eps16(4 * b + r, gg, n_freq)
                if col >= 0:
                    rows[16 * b + 4 * gg + r] = W_cols[:, col]
    return rows


def pack_network_backward16(sd):
    """Transposed stream of one network in the x16 layout.  Order and slab counts as pack_network_backward:
    views^T (18 blocks x 8 quads = 9 slabs) | feature^T 16 | L7^T 16 | L6^T 16 | L5^T (20 blocks: 16 hidden + 4
    encoding) 20 | L4^T..L1^T 64 | L0^T (4 encoding blocks) 4 = 145 slabs.  k-step t of lane group g contracts
    over output feature kappa16(t, g) of the forward layer.  Returns float32 [STREAM_SLABS * SLAB_FLOATS]."""
    g = lambda k: np.asarray(sd[k], dtype=np.float32)
    cols = lambda n_k: np.stack([kappa16(np.arange(n_k), gg) for gg in range(4)], 1)
    segs = []
    Wv = g("views_linears.0.weight")                                     # [128, 256 + 27]
    segs.append(_pack16(np.concatenate([Wv[:, :256].T, _enc_rows16(Wv[:, 256:], 4, 2)], 0), cols(32), 18))
    segs.append(_pack16(g("feature_linear.weight").T, cols(64), 16))
    for l in (7, 6):
        segs.append(_pack16(g("pts_linears.%d.weight" % l).T, cols(64), 16))
    W5 = g("pts_linears.5.weight")                                       # [256, 63 + 256], input columns first
    segs.append(_pack16(np.concatenate([W5[:, 63:].T, _enc_rows16(W5[:, :63], 10, 4)], 0), cols(64), 20))
    for l in (4, 3, 2, 1):
        segs.append(_pack16(g("pts_linears.%d.weight" % l).T, cols(64), 16))
    segs.append(_pack16(_enc_rows16(g("pts_linears.0.weight"), 10, 4), cols(64), 4))
    stream = np.concatenate([x.reshape(-1) for x in segs]).astype(np.float32)
    assert stream.size == STREAM_SLABS * SLAB_FLOATS
    return stream


# ----------------------------------------------------------------------------------------------------------
# "bf16x3" forward layout (csrc/nsr_b3.inc, k_render_b3): v_mfma_f32_32x32x16_bf16, every fp32 weight as three bf16 pieces
# ----------------------------------------------------------------------------------------------------------
# One MFMA consumes, per lane l, A[row l&31][8 k-slots of lane half l>>5] (8 bf16 = 16 bytes).  Slot i of k16 block kb is
# k-step t = 8*kb + i of the fp32 x32 layout above, for both lane halves, so the column tables (kappa / eps) are shared.
# A chunk (1 KiB = [64 lanes][8] bf16) is ONE piece of ONE (k16 block, output block) fragment; a step is 4 chunks, a
# group 12 steps = 3 slabs; the tables mirror B3Sched<8> / B3Sched<4> in csrc/nsr_b3.inc.
STREAM_SLABS_B3 = 219
PACKED_B3_FLOATS = STREAM_SLABS_B3 * SLAB_FLOATS + AUX_FLOATS
B3_STEPS8 = ([(0, 0), (1, 1), (1, 2), (2, 3)], [(0, 1), (0, 3), (2, 0), (2, 2)], [(0, 2), (1, 3), (1, 4), (2, 5)],
             [(0, 4), (1, 5), (1, 6), (2, 7)], [(0, 5), (0, 7), (2, 4), (2, 6)], [(0, 6), (1, 7), (1, 0), (2, 1)])
B3_STEPS4 = ([(0, 0), (1, 1), (1, 2), (2, 3)], [(0, 1), (0, 3), (2, 0), (2, 2)], [(0, 2), (1, 3), (1, 0), (2, 1)])


def bf16_round(x):
    """fp32 -> nearest-even bf16, returned as fp32 (low 16 bits zero)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32)
    return r.view(np.float32)


def split_bf16x3(x):
    """x (fp32) -> three bf16-valued fp32 arrays with x == p0 + p1 + p2 exactly (|x| well inside the fp32 range)."""
    x = np.ascontiguousarray(x, np.float32)
    p0 = bf16_round(x)
    r1 = (x - p0).astype(np.float32)
    p1 = bf16_round(r1)
    r2 = (r1 - p1).astype(np.float32)
    p2 = bf16_round(r2)
    return p0, p1, p2


def _pack_b3(W, cols, n_mo):
    """W [32*n_mo, K]; cols [n_ksteps, 2] (reference column or -1).  Returns uint16 [n_groups*48, 64, 8]: the chunks of
    the segment in stream order.  n_ksteps is padded with zero blocks to whole groups."""
    steps = B3_STEPS8 if n_mo == 8 else B3_STEPS4
    spk = len(steps)                         # steps per k16 block
    nb = 12 // spk                           # k16 blocks per group
    assert W.shape[0] == 32 * n_mo and cols.shape[0] % 8 == 0
    n_kb = cols.shape[0] // 8
    n_groups = -(-n_kb // nb)
    cols = np.concatenate([cols, np.full((8 * (n_groups * nb - n_kb), 2), -1, cols.dtype)], 0)
    Wp = np.concatenate([W, np.zeros((W.shape[0], 1), W.dtype)], 1)       # column -1 -> zeros
    lane = np.arange(64)
    row, h = lane & 31, lane >> 5
    frag = np.empty((n_groups * nb, n_mo, 64, 8), np.float32)             # [k16 block][output block][lane][slot]
    for kb in range(n_groups * nb):
        for i in range(8):
            c = cols[8 * kb + i][h]
            for mo in range(n_mo):
                frag[kb, mo, :, i] = Wp[32 * mo + row, c]
    pieces = [(p.view(np.uint32) >> 16).astype(np.uint16) for p in split_bf16x3(frag)]
    out = np.empty((n_groups * 48, 64, 8), np.uint16)
    n = 0
    for g in range(n_groups):
        for G in range(12):
            kb = nb * g + G // spk
            for piece, mo in steps[G % spk]:
                out[n] = pieces[piece][kb, mo]
                n += 1
    return out


def pack_network_b3(sd):
    """The forward stream of pack_network with every weight as three bf16 pieces, in the chunk order of k_render_b3,
    followed by the SAME fp32 aux block.  Returns float32 [PACKED_B3_FLOATS] (the stream part is packed bf16 pairs
    viewed as floats)."""
    g = lambda k: np.asarray(sd[k], dtype=np.float32)
    aux = pack_network(sd)[STREAM_SLABS * SLAB_FLOATS:]                  # also validates the shapes
    t = np.arange(128)
    cols_main = np.stack([kappa(t, 0), kappa(t, 1)], 1)
    cols_enc = np.array([[eps(tt, 0, 10), eps(tt, 1, 10)] for tt in range(32)])
    cols_dir = np.array([[eps(tt, 0, 4), eps(tt, 1, 4)] for tt in range(16)])
    segs = [_pack_b3(g("pts_linears.0.weight"), cols_enc, 8)]
    for i in range(1, 8):
        W = g("pts_linears.%d.weight" % i)
        if i == 5:
            segs.append(_pack_b3(W[:, :63], cols_enc, 8))
            W = W[:, 63:]
        segs.append(_pack_b3(W, cols_main, 8))
    segs.append(_pack_b3(g("feature_linear.weight"), cols_main, 8))
    cols_views = np.concatenate([cols_main, np.where(cols_dir >= 0, cols_dir + 256, -1)], 0)
    segs.append(_pack_b3(g("views_linears.0.weight"), cols_views, 4))
    stream = np.concatenate([x.reshape(-1) for x in segs])
    assert stream.size * 2 == STREAM_SLABS_B3 * SLAB_FLOATS * 4
    return np.concatenate([stream.view(np.float32), aux]).astype(np.float32, copy=False)


STREAM_SLABS_B3_BWD = 234


def pack_network_backward_b3(sd):
    """Transposed stream of one network for k_render_vjp_b3, bf16x3 layout.  Every segment of pack_network_backward that
    has encoding rows is split into a 4-block GEMM for those rows (1 or 2 real blocks, zero-padded) FOLLOWED by the
    8-block GEMM for the 256 hidden features (the kernel runs the small one first so that its accumulators are dead
    when the big one starts).  Order: views^T enc | views^T | feature^T | L7^T | L6^T | L5^T enc | L5^T | L4^T..L1^T |
    L0^T enc.  Returns float32 [STREAM_SLABS_B3_BWD * SLAB_FLOATS] (packed bf16 pairs viewed as floats)."""
    g = lambda k: np.asarray(sd[k], dtype=np.float32)
    t128, t64 = np.arange(128), np.arange(64)
    cols256 = np.stack([kappa(t128, 0), kappa(t128, 1)], 1)              # K = 256: 16 k16 blocks
    cols128 = np.stack([kappa(t64, 0), kappa(t64, 1)], 1)                # K = 128 (views layer's outputs): 8 blocks
    pad4 = lambda M: np.concatenate([M, np.zeros((128 - M.shape[0], M.shape[1]), np.float32)], 0)
    segs = []
    Wv = g("views_linears.0.weight")                                     # [128, 256 + 27]
    segs.append(_pack_b3(pad4(_enc_rows_T(Wv[:, 256:], 4, 1)), cols128, 4))
    segs.append(_pack_b3(np.ascontiguousarray(Wv[:, :256].T), cols128, 8))
    segs.append(_pack_b3(np.ascontiguousarray(g("feature_linear.weight").T), cols256, 8))
    for l in (7, 6):
        segs.append(_pack_b3(np.ascontiguousarray(g("pts_linears.%d.weight" % l).T), cols256, 8))
    W5 = g("pts_linears.5.weight")                                       # [256, 63 + 256], input columns first
    segs.append(_pack_b3(pad4(_enc_rows_T(W5[:, :63], 10, 2)), cols256, 4))
    segs.append(_pack_b3(np.ascontiguousarray(W5[:, 63:].T), cols256, 8))
    for l in (4, 3, 2, 1):
        segs.append(_pack_b3(np.ascontiguousarray(g("pts_linears.%d.weight" % l).T), cols256, 8))
    segs.append(_pack_b3(pad4(_enc_rows_T(g("pts_linears.0.weight"), 10, 2)), cols256, 4))
    stream = np.concatenate([x.reshape(-1) for x in segs])
    assert stream.size * 2 == STREAM_SLABS_B3_BWD * SLAB_FLOATS * 4
    return stream.view(np.float32)


# ----------------------------------------------------------------------------------------------------------
# "f16x2" forward layout (csrc/nsr_h2.inc, k_render_h2): v_mfma_f32_32x32x16_f16, every fp32 weight as two fp16 pieces
# ----------------------------------------------------------------------------------------------------------
# Same MFMA operand layout as bf16x3 (8 k-slots per lane, slot i of k16 block kb = k-step 8 kb + i of the fp32 x32 layout),
# 2 pieces x 2 bytes per weight: the stream has the fp32 stream's size and segment order (STREAM_SLABS).  Step st of a k16
# block is the 4 chunks [hi(2st), hi(2st+1), lo(2(st^1)), lo(2(st^1)+1)].
# Range management (all exact powers of two):
#   sw[l]   weights of layer l are stored as W * 2^sw[l] with max|W| * 2^sw[l] in [2^14, 2^15)          (l = 0..9, 8 =
#           feature_linear, 9 = views_linears.0; the two column groups of L5 and of the views layer share one sw)
#   ca[l]   the activations entering layer l are multiplied so that the fp16 pieces are those of x * 2^ca[l]  (0 unless
#           `act_scale_log2` says otherwise)
#   the accumulators of layer l then hold 2^(sw[l] + ca[l]) * (W x + b): biases are stored times that factor, and the
#   registers the NEXT layer reads carry it (P[l+1] = sw[l] + ca[l]); the multiplier applied before its split is
#   2^(ca[l+1] - P[l+1]).  The two VALU heads read scaled registers too: their weights are stored times 2^-P.
AUX_H2_SCALE = 3080
AUX_H2_BWD = 3096
STREAM_SLABS_H2_BWD = 146
H2_WEIGHT_TOP_LOG2 = 15          # scaled weights stay below 2^15 (fp16 max is 65504 = 2^16 - 32)


def f16_round(x):
    """fp32 -> nearest-even fp16 (subnormals kept), returned as fp32"""
    with np.errstate(over="ignore"):
        return np.ascontiguousarray(x, np.float32).astype(np.float16).astype(np.float32)


def split_f16x2(x):
    """x (fp32) -> (hi, lo) fp16-valued fp32 arrays, hi = fp16(x), lo = fp16(x - hi)"""
    x = np.ascontiguousarray(x, np.float32)
    hi = f16_round(x)
    lo = f16_round((x - hi).astype(np.float32))
    return hi, lo


def h2_weight_scale_log2(*mats):
    """sw: the power of two that puts the largest |entry| of the given matrices into [2^14, 2^15)"""
    top = max(float(np.abs(np.asarray(m, np.float32)).max()) for m in mats)
    if not np.isfinite(top):
        raise ValueError("pack_network_h2: non-finite weight")
    if top == 0.0:
        return 0
    return int(H2_WEIGHT_TOP_LOG2 - 1 - np.floor(np.log2(top)))


def _pack_h2(W, cols, n_mo):
    """W [32*n_mo, K] (ALREADY scaled); cols [n_ksteps, 2] (reference column or -1), n_ksteps a multiple of 8.  Returns uint16
    [n_kb * n_mo * 2 chunks, 64, 8]: per k16 block, per step st, the chunks hi(2st), hi(2st+1), lo(2(st^1)), lo(2(st^1)+1)
    (csrc/nsr_h2.inc: the six MFMAs of a step then touch four different accumulators)."""
    assert W.shape[0] == 32 * n_mo and cols.shape[0] % 8 == 0 and n_mo % 2 == 0
    n_kb = cols.shape[0] // 8
    Wp = np.concatenate([W, np.zeros((W.shape[0], 1), W.dtype)], 1)       # column -1 -> zeros
    lane = np.arange(64)
    row, h = lane & 31, lane >> 5
    frag = np.empty((n_kb, n_mo, 64, 8), np.float32)                      # [k16 block][output block][lane][slot]
    for kb in range(n_kb):
        for i in range(8):
            c = cols[8 * kb + i][h]
            for mo in range(n_mo):
                frag[kb, mo, :, i] = Wp[32 * mo + row, c]
    hi, lo = (p.astype(np.float16).view(np.uint16) for p in split_f16x2(frag))
    out = np.empty((n_kb * n_mo * 2, 64, 8), np.uint16)
    n = 0
    for kb in range(n_kb):
        for st in range(n_mo // 2):                      # step st: hi pieces of blocks 2st, 2st+1, lo pieces of the partner pair
            hb, lb = 2 * st, (2 * (st ^ 1) if n_mo > 2 else 0)       # (a 2-block GEMM has one step: hi(0) hi(1) lo(0) lo(1))
            for piece, mo in ((hi, hb), (hi, hb + 1), (lo, lb), (lo, lb + 1)):
                out[n] = piece[kb, mo]
                n += 1
    return out


def h2_scales(sd, act_scale_log2=None):
    """(sw[10], ca[10]) of pack_network_h2 for this network; act_scale_log2: None (all 0) or 10 integers ca[l]."""
    g = lambda k: np.asarray(sd[k], dtype=np.float32)
    sw = [h2_weight_scale_log2(g("pts_linears.%d.weight" % l)) for l in range(8)]
    sw.append(h2_weight_scale_log2(g("feature_linear.weight")))
    sw.append(h2_weight_scale_log2(g("views_linears.0.weight")))
    ca = [0] * 10 if act_scale_log2 is None else [int(v) for v in act_scale_log2]
    assert len(ca) == 10
    return sw, ca


def h2_report(sd, coord_max=4.0, act_scale_log2=None):
    """Pack-time range report of the f16x2 layout for one network: per layer (pts_linears.0-7, feature_linear,
    views_linears.0) the weight scale sw (weights are stored x 2^sw, largest entry in [2^14, 2^15)), the activation scale ca,
    and two bounds on the largest hidden activation ENTERING the layer, log2, against the fp16 ceiling 2^16 (a scaled
    activation >= 65504 sends the point to the fp32 kernel: include/nsr.h, range safety net):
      worst_log2  -- guaranteed: interval arithmetic with the infinity-norm of every layer (sum_k |W_jk| x_max + |b_j|),
                     inputs |sin|, |cos| <= 1 and |coordinate| <= coord_max.  Loose by construction (it assumes every
                     term of every dot product aligned), but a network whose worst case fits can NEVER overflow;
      typical_log2 -- the same recursion with the 2-norm of the rows and an RMS input: what a random-sign sum gives.
    headroom_* = 16 + ... - that bound - ca, in bits; negative worst-case head-room is normal for real networks (the kernels
    check every point at run time), negative TYPICAL head-room means most points will take the fp32 route: use mlp="fp32"."""
    g = lambda k: np.asarray(sd[k], dtype=np.float64)
    sw, ca = h2_scales(sd, act_scale_log2)
    names = ["pts_linears.%d" % l for l in range(8)] + ["feature_linear", "views_linears.0"]
    x_inf = np.concatenate([np.full(3, coord_max), np.ones(60)])            # the 63 encoding channels (RH:39-48)
    x_rms = np.concatenate([np.full(3, coord_max / np.sqrt(3.0)), np.full(60, np.sqrt(0.5))])
    d_inf, d_rms = np.ones(27), np.concatenate([np.full(3, 1 / np.sqrt(3.0)), np.full(24, np.sqrt(0.5))])
    rows = []
    h_inf = h_rms = None
    for l, name in enumerate(names):
        W, b = g(name + ".weight"), g(name + ".bias")
        if l == 0:
            in_inf, in_rms = x_inf, x_rms
        elif l == 5:
            in_inf, in_rms = np.concatenate([x_inf, h_inf]), np.concatenate([x_rms, h_rms])
        elif l == 9:
            in_inf, in_rms = np.concatenate([h_inf, d_inf]), np.concatenate([h_rms, d_rms])
        else:
            in_inf, in_rms = h_inf, h_rms
        # the HIDDEN part of what enters the layer (the encodings are bounded by construction): cat([x(63), h(256)]) for the
        # skip layer, cat([feature(256), dirs(27)]) for the view layer (a trailing slice dropped 27 features there)
        hid_inf = in_inf[63:] if l == 5 else (in_inf[:256] if l == 9 else in_inf)
        hid_rms = in_rms[63:] if l == 5 else (in_rms[:256] if l == 9 else in_rms)
        ent_inf, ent_rms = float(hid_inf.max() if l else 1.0), float(np.abs(hid_rms).max() if l else 1.0)
        rows.append(dict(layer=name, sw=int(sw[l]), ca=int(ca[l]), max_abs_weight=float(np.abs(W).max()),
                         worst_log2=float(np.log2(max(ent_inf, 1e-300))), typical_log2=float(np.log2(max(ent_rms, 1e-300))),
                         headroom_worst_bits=float(16 - ca[l] - np.log2(max(ent_inf, 1e-300))),
                         headroom_typical_bits=float(16 - ca[l] - np.log2(max(ent_rms, 1e-300)))))
        h_inf = np.abs(W) @ in_inf + np.abs(b)
        h_rms = np.sqrt((W * W) @ (in_rms * in_rms) + b * b)                # relu / identity: magnitude carried forward
    return rows


def pack_network_h2(sd, act_scale_log2=None):
    """The forward stream of pack_network with every (scaled) weight as two fp16 pieces, in the chunk order of k_render_h2,
    followed by the aux block with scaled biases / head weights and the activation multipliers.  Returns float32
    [PACKED_FLOATS] (the stream part is packed fp16 pairs viewed as floats)."""
    g = lambda k: np.asarray(sd[k], dtype=np.float32)
    aux = pack_network(sd)[STREAM_SLABS * SLAB_FLOATS:].copy()          # also validates the shapes
    sw, ca = h2_scales(sd, act_scale_log2)
    p2 = lambda e: np.float32(2.0) ** np.float32(e)
    t = np.arange(128)
    cols_main = np.stack([kappa(t, 0), kappa(t, 1)], 1)
    cols_enc = np.array([[eps(tt, 0, 10), eps(tt, 1, 10)] for tt in range(32)])
    cols_dir = np.array([[eps(tt, 0, 4), eps(tt, 1, 4)] for tt in range(16)])
    segs = [_pack_h2(g("pts_linears.0.weight") * p2(sw[0]), cols_enc, 8)]
    for i in range(1, 8):
        W = g("pts_linears.%d.weight" % i) * p2(sw[i])
        if i == 5:
            segs.append(_pack_h2(W[:, :63], cols_enc, 8))
            W = W[:, 63:]
        segs.append(_pack_h2(np.ascontiguousarray(W), cols_main, 8))
    segs.append(_pack_h2(g("feature_linear.weight") * p2(sw[8]), cols_main, 8))
    cols_views = np.concatenate([cols_main, np.where(cols_dir >= 0, cols_dir + 256, -1)], 0)
    segs.append(_pack_h2(g("views_linears.0.weight") * p2(sw[9]), cols_views, 4))
    stream = np.concatenate([x.reshape(-1) for x in segs])
    assert stream.size * 2 == STREAM_SLABS * SLAB_FLOATS * 4
    # aux: biases times 2^(sw + ca); head weights times 2^-P of the registers they read
    P = [0] + [sw[l] + ca[l] for l in range(9)]                         # P[l]: scale carried by the hidden input of layer l (1..9)
    for L in range(9):
        aux[AUX_BIAS + L * 256:AUX_BIAS + (L + 1) * 256] *= p2(sw[L] + ca[L])
    aux[AUX_BIAS_V:AUX_BIAS_V + 128] *= p2(sw[9] + ca[9])
    aux[AUX_W_ALPHA:AUX_W_ALPHA + 256] *= p2(-P[8])                     # alpha_linear reads h7 = the input of feature_linear
    aux[AUX_W_RGB:AUX_W_RGB + 384] *= p2(-(sw[9] + ca[9]))              # rgb_linear reads the views layer's output
    sc = np.zeros(16, np.float32)
    sc[0] = p2(ca[0])                                                   # encoding @ L0
    for L in range(1, 9):
        sc[L] = p2(ca[L] - P[L])                                        # hidden input of L1..L7, feature_linear
    sc[9] = p2(ca[5])                                                   # encoding @ L5 (shares L5's sw and ca)
    sc[10] = p2(ca[9] - P[9])                                           # feature @ views layer
    sc[11] = p2(ca[9])                                                  # direction encoding @ views layer
    aux[AUX_H2_SCALE:AUX_H2_SCALE + 16] = sc
    # multipliers of the backward chain (csrc/nsr_h2_bwd.inc; gradients are normalised per point, their own scale cb = 0):
    # the source of transposed layer j carries the weight scale of the GEMM before it (the rgb head's scaled weights first)
    tb = np.zeros(16, np.float32)
    tb[0] = p2(sw[9] + ca[9])                                           # views^T: gv was built from rgb weights x 2^-(sw9+ca9)
    tb[1] = p2(-sw[9])                                                  # feature^T reads the output of views^T
    for j, l in enumerate((8, 7, 6, 5, 4, 3, 2, 1)):                    # L7^T .. L1^T, L0^T read the output of layer l's transpose
        tb[2 + j] = p2(-sw[l])
    tb[10] = p2(sw[8] + P[8])                                           # alpha_linear^T: stored weights x 2^-P8, accumulator x 2^sw8
    tb[11], tb[12], tb[13] = p2(-sw[9]), p2(-sw[5]), p2(-sw[0])         # encoding-gradient rows of views^T, L5^T, L0^T
    aux[AUX_H2_BWD:AUX_H2_BWD + 16] = tb
    if not np.isfinite(aux).all():
        raise ValueError("pack_network_h2: a scaled bias / head weight left the fp32 range")
    return np.concatenate([stream.view(np.float32), aux]).astype(np.float32, copy=False)


def pack_network_backward_h2(sd):
    """Transposed stream of one network for k_render_vjp_h2, f16x2 layout, weights scaled like the forward stream (2^sw of
    their layer).  The encoding rows of pack_network_backward's 9-, 10- and 2-block segments are 2-block GEMMs run BEFORE the
    8-block GEMM of the same layer.  Order: views^T enc | views^T | feature^T | L7^T | L6^T | L5^T enc | L5^T | L4^T..L1^T |
    L0^T enc.  Returns float32 [STREAM_SLABS_H2_BWD * SLAB_FLOATS] (packed fp16 pairs viewed as floats); the multipliers the
    kernel needs are in the aux block of pack_network_h2."""
    g = lambda k: np.asarray(sd[k], dtype=np.float32)
    sw, _ = h2_scales(sd)
    p2 = lambda e: np.float32(2.0) ** np.float32(e)
    t128, t64 = np.arange(128), np.arange(64)
    cols256 = np.stack([kappa(t128, 0), kappa(t128, 1)], 1)              # K = 256: 16 k16 blocks
    cols128 = np.stack([kappa(t64, 0), kappa(t64, 1)], 1)                # K = 128 (views layer's outputs): 8 blocks
    pad2 = lambda M: np.concatenate([M, np.zeros((64 - M.shape[0], M.shape[1]), np.float32)], 0)
    segs = []
    Wv = g("views_linears.0.weight") * p2(sw[9])                         # [128, 256 + 27]
    segs.append(_pack_h2(pad2(_enc_rows_T(Wv[:, 256:], 4, 1)), cols128, 2))
    segs.append(_pack_h2(np.ascontiguousarray(Wv[:, :256].T), cols128, 8))
    segs.append(_pack_h2(np.ascontiguousarray(g("feature_linear.weight").T) * p2(sw[8]), cols256, 8))
    for l in (7, 6):
        segs.append(_pack_h2(np.ascontiguousarray(g("pts_linears.%d.weight" % l).T) * p2(sw[l]), cols256, 8))
    W5 = g("pts_linears.5.weight") * p2(sw[5])                           # [256, 63 + 256], input columns first
    segs.append(_pack_h2(pad2(_enc_rows_T(W5[:, :63], 10, 2)), cols256, 2))
    segs.append(_pack_h2(np.ascontiguousarray(W5[:, 63:].T), cols256, 8))
    for l in (4, 3, 2, 1):
        segs.append(_pack_h2(np.ascontiguousarray(g("pts_linears.%d.weight" % l).T) * p2(sw[l]), cols256, 8))
    segs.append(_pack_h2(pad2(_enc_rows_T(g("pts_linears.0.weight") * p2(sw[0]), 10, 2)), cols256, 2))
    stream = np.concatenate([x.reshape(-1) for x in segs])
    assert stream.size * 2 == STREAM_SLABS_H2_BWD * SLAB_FLOATS * 4
    return stream.view(np.float32)
