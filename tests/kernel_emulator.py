"""Lane-level numpy emulation of ONE wave of the kernel's network pass (csrc/nsr_kernels.hip: mlp_pass).

Purpose: check the packer (pack.py) and the kernel's fragment indexing against the oracle on the CPU, before
any GPU time is spent.  It follows the HIP code statement for statement (ring steps, fragment reads,
register-chained activations) and models v_mfma_f32_32x32x2_f32 as documented:
    A[i][k] = a[lane = i + 32k],  B[k][j] = b[lane = j + 32k],
    D[(r&3) + 8(r>>2) + 4(lane>>5)][lane&31] = acc[lane][r].
The MFMA model itself is verified on hardware by nsr_selftest."""
import numpy as np

LANE = np.arange(64)
ROW_OF = ((np.arange(16) & 3) + 8 * (np.arange(16) >> 2))[None, :] + 4 * (LANE >> 5)[:, None]   # [64,16]
COL_OF = (LANE & 31)[:, None].repeat(16, 1)                                                      # [64,16]


def mfma(a, b, acc):
    """a, b: [64] floats (one VGPR each); acc: [64,16]."""
    A = a.reshape(2, 32).T            # [i, k]
    B = b.reshape(2, 32)              # [k, j]
    D = (A.astype(np.float64) @ B.astype(np.float64))
    out = acc.copy()
    out += D[ROW_OF, COL_OF].astype(np.float32)
    return out


class Stream:
    """The packed stream as the kernel sees it: consecutive 1 KiB chunks of [64 lanes][4]."""

    def __init__(self, stream_floats):
        self.chunks = stream_floats.reshape(-1, 64, 4)
        self.pos = 0

    def seg(self, nmo, ntq, bop, acc):
        """csrc/nsr_kernels.hip `seg<NMO,NTQ>`: chunk n <-> (k-quad n // NMO, output block n % NMO)."""
        for n in range(nmo * ntq):
            frag = self.chunks[self.pos]
            self.pos += 1
            tq, mo = divmod(n, nmo)
            for kk in range(4):
                acc[mo] = mfma(frag[:, kk], bop(4 * tq + kk), acc[mo])


def load_bias(aux, off, nmo):
    h = LANE >> 5
    acc = np.zeros((nmo, 64, 16), np.float32)
    for mo in range(nmo):
        for rq in range(4):
            base = off + ((mo * 4 + rq) * 2 + h) * 4
            for ri in range(4):
                acc[mo, :, rq * 4 + ri] = aux[base + ri]
    return acc


def _encode(X, nfreq, n):
    h = LANE >> 5
    e = np.zeros((n, 64), np.float32)
    for L in range(nfreq):
        for ax in range(3):
            arg = (X[:, ax] * np.float32(2 ** L)).astype(np.float32)
            e[3 * L + ax] = np.where(h == 1, np.cos(arg), np.sin(arg)).astype(np.float32)
    e[3 * nfreq] = np.where(h == 1, X[:, 2], X[:, 0])
    e[3 * nfreq + 1] = np.where(h == 1, 0.0, X[:, 1])
    return e


def mlp_pass(packed, pts, dirs, masks=None):
    """pts, dirs: [32,3] -> raw [32,4], emulating one wave (lanes j and j+32 both own point j).
    `masks`: optional dict filled with the relu patterns ([nmo,64,16] bool per layer), like mask capture."""
    from neural_sim_nerf_amd import pack as PK
    st = Stream(packed[:PK.STREAM_SLABS * PK.SLAB_FLOATS])
    aux = packed[PK.STREAM_SLABS * PK.SLAB_FLOATS:]
    h = LANE >> 5
    P = np.concatenate([pts, pts], 0).astype(np.float32)
    V = np.concatenate([dirs, dirs], 0).astype(np.float32)
    e = _encode(P, 10, 32)
    ed = _encode(V, 4, 16)
    regs = lambda arr: (lambda t: arr[t >> 4][:, t & 15])
    acc = load_bias(aux, PK.AUX_BIAS, 8)
    st.seg(8, 8, lambda t: e[t], acc)
    if masks is not None:
        masks[0] = acc > 0
    inp = np.maximum(acc, 0)
    alpha_part = np.zeros(64, np.float32)
    for L in range(1, 9):
        acc = load_bias(aux, PK.AUX_BIAS + L * 256, 8)
        if L == 5:
            st.seg(8, 8, lambda t: e[t], acc)
        if L == 8:
            for tq in range(32):
                for kk in range(4):
                    w = aux[PK.AUX_W_ALPHA + (tq * 2 + h) * 4 + kk]
                    alpha_part = alpha_part + w * inp[(4 * tq + kk) >> 4][:, (4 * tq + kk) & 15]
        st.seg(8, 32, regs(inp), acc)
        if masks is not None and L < 8:
            masks[L] = acc > 0
        inp = np.maximum(acc, 0) if L < 8 else acc.copy()
    av = load_bias(aux, PK.AUX_BIAS_V, 4)
    st.seg(4, 36, lambda t: inp[t >> 4][:, t & 15] if t < 128 else ed[t - 128], av)
    if masks is not None:
        masks[8] = av > 0
    assert st.pos == 145 * 16
    part = np.zeros((4, 64), np.float32)
    part[3] = alpha_part
    for c in range(3):
        for mo in range(4):
            for rq in range(4):
                for ri in range(4):
                    w = aux[PK.AUX_W_RGB + c * 128 + ((mo * 4 + rq) * 2 + h) * 4 + ri]
                    part[c] = part[c] + w * np.maximum(av[mo][:, rq * 4 + ri], 0)
    raw = np.zeros((32, 4), np.float32)
    for c in range(4):
        bias = aux[PK.AUX_B_RGB + c] if c < 3 else aux[PK.AUX_B_ALPHA]
        raw[:, c] = part[c][:32] + part[c][32:] + bias
    return raw


def _embed_bwd(X, G, nfreq):
    """csrc embed_bwd: per-lane-half contribution, then the two halves are added."""
    h = LANE >> 5
    out = np.zeros((3, 64), np.float64)
    out[0] = np.where(h == 1, 0.0, G[3 * nfreq])
    out[1] = np.where(h == 1, 0.0, G[3 * nfreq + 1])
    out[2] = np.where(h == 1, G[3 * nfreq], 0.0)
    for L in range(nfreq):
        for ax in range(3):
            f = np.float32(2 ** L)
            arg = (X[:, ax] * f).astype(np.float32)
            out[ax] += f * G[3 * L + ax] * np.where(h == 1, -np.sin(arg), np.cos(arg))
    return (out[:, :32] + out[:, 32:]).T          # [32,3]


def mlp_bwd_pass(packed_fwd, stream_bwd, masks, pts, dirs, g_raw):
    """Emulates csrc mlp_bwd_pass for one wave: g_raw [32,4] -> (dL/dpts [32,3], dL/ddirs [32,3])."""
    from neural_sim_nerf_amd import pack as PK
    st = Stream(stream_bwd)
    aux = packed_fwd[PK.STREAM_SLABS * PK.SLAB_FLOATS:]
    h = LANE >> 5
    P = np.concatenate([pts, pts], 0).astype(np.float32)
    V = np.concatenate([dirs, dirs], 0).astype(np.float32)
    G = np.concatenate([g_raw, g_raw], 0).astype(np.float32)      # per lane
    regs = lambda arr: (lambda t: arr[t >> 4][:, t & 15])
    gv = np.zeros((4, 64, 16), np.float32)
    for mo in range(4):
        for rq in range(4):
            for ri in range(4):
                idx = ((mo * 4 + rq) * 2 + h) * 4 + ri
                v = (aux[PK.AUX_W_RGB + idx] * G[:, 0] + aux[PK.AUX_W_RGB + 128 + idx] * G[:, 1]
                     + aux[PK.AUX_W_RGB + 256 + idx] * G[:, 2])
                gv[mo][:, rq * 4 + ri] = np.where(masks[8][mo][:, rq * 4 + ri], v, 0)
    ae = np.zeros((1, 64, 16), np.float32)
    st.seg(1, 16, regs(gv), ae)                                    # the direction-encoding rows: their own 1-block segment
    dv = _embed_bwd(V, [ae[0][:, t] for t in range(16)], 4)
    acc = np.zeros((8, 64, 16), np.float32)
    st.seg(8, 16, regs(gv), acc)
    gin = acc.copy()
    dp5 = None
    for idx in range(8):
        if idx == 3:                                               # L5^T: its 64 encoding rows first
            genc = np.zeros((2, 64, 16), np.float32)
            st.seg(2, 32, regs(gin), genc)
            dp5 = _embed_bwd(P, [genc[t >> 4][:, t & 15] for t in range(32)], 10)
        acc[:] = 0
        st.seg(8, 32, regs(gin), acc)
        if idx == 0:
            for tq in range(32):
                for kk in range(4):
                    t = 4 * tq + kk
                    acc[t >> 4][:, t & 15] += aux[PK.AUX_W_ALPHA + (tq * 2 + h) * 4 + kk] * G[:, 3]
        gin = np.where(masks[7 - idx], acc, 0).astype(np.float32)
    genc = np.zeros((2, 64, 16), np.float32)
    st.seg(2, 32, regs(gin), genc)
    assert st.pos == 145 * 16
    dp = _embed_bwd(P, [genc[t >> 4][:, t & 15] for t in range(32)], 10) + dp5
    return dp, dv


# ----------------------------------------------------------------------------------------------------------
# x16 variant: v_mfma_f32_16x16x4_f32, 16 points per wave (csrc mlp_pass16)
#   A[i][k] = a[lane = i + 16k], B[k][j] = b[lane = j + 16k], D[4*(lane>>4) + r][lane&15] = acc[lane][r]
# ----------------------------------------------------------------------------------------------------------
ROW16 = 4 * (LANE >> 4)[:, None] + np.arange(4)[None, :]          # [64,4]
COL16 = (LANE & 15)[:, None].repeat(4, 1)


def mfma16(a, b, acc):
    A = a.reshape(4, 16).T            # [i, k]
    B = b.reshape(4, 16)              # [k, j]
    D = A.astype(np.float64) @ B.astype(np.float64)
    return acc + D[ROW16, COL16].astype(np.float32)


class Stream16(Stream):
    def seg(self, nmo, ntq, bop, acc):
        for n in range(nmo * ntq):
            frag = self.chunks[self.pos]
            self.pos += 1
            tq, mo = divmod(n, nmo)
            for kk in range(4):
                acc[mo] = mfma16(frag[:, kk], bop(4 * tq + kk), acc[mo])


def _enc16_column(t, g, nfreq):
    """(octave, sin/cos, axis) of encoding register t in lane group g, as csrc mlp_pass16 assigns them."""
    if nfreq == 10:
        return 5 * (g >> 1) + t // 3, g & 1, np.full_like(g, t % 3)
    q = (6 * nfreq // 4) * g + t                       # directions: group g = octave g, t = 3 sc + axis
    return q // 6, (q % 6) // 3, q % 3


def _encode16(X, nfreq, n):
    g = LANE >> 4
    per = 6 * nfreq // 4
    e = np.zeros((n, 64), np.float32)
    for t in range(per):
        L, sc, ax = _enc16_column(t, g, nfreq)
        arg = (X[LANE, ax] * np.exp2(L).astype(np.float32)).astype(np.float32)
        e[t] = np.where(sc == 1, np.cos(arg), np.sin(arg)).astype(np.float32)
    e[per] = np.where(g < 3, X[LANE, np.minimum(g, 2)], 0.0)
    return e


def mlp_pass16(packed16, pts, dirs, masks=None):
    """pts, dirs [16,3] -> raw [16,4]; lanes j, j+16, j+32, j+48 all own point j.
    `masks`: optional dict filled with the relu patterns ([nmo,64,4] bool per layer), like mask capture."""
    from neural_sim_nerf_amd import pack as PK
    st = Stream16(packed16[:PK.STREAM_SLABS * PK.SLAB_FLOATS])
    aux = packed16[PK.STREAM_SLABS * PK.SLAB_FLOATS:]
    g = LANE >> 4
    P = np.tile(pts, (4, 1)).astype(np.float32)
    V = np.tile(dirs, (4, 1)).astype(np.float32)
    e, ed = _encode16(P, 10, 16), _encode16(V, 4, 8)
    regs = lambda arr: (lambda t: arr[t >> 2][:, t & 3])

    def bias(off, nmo):
        acc = np.zeros((nmo, 64, 4), np.float32)
        for mo in range(nmo):
            for r in range(4):
                acc[mo][:, r] = aux[off + 16 * mo + 4 * g + r]
        return acc
    acc = bias(PK.AUX_BIAS, 16)
    st.seg(16, 4, lambda t: e[t], acc)
    if masks is not None:
        masks[0] = acc > 0
    inp = np.maximum(acc, 0)
    alpha_part = np.zeros(64, np.float32)
    for L in range(1, 9):
        acc = bias(PK.AUX_BIAS + L * 256, 16)
        if L == 5:
            st.seg(16, 4, lambda t: e[t], acc)
        if L == 8:
            for t in range(64):
                alpha_part = alpha_part + aux[PK.AUX_W_ALPHA + 16 * (t >> 2) + 4 * g + (t & 3)] * inp[t >> 2][:, t & 3]
        st.seg(16, 16, regs(inp), acc)
        if masks is not None and L < 8:
            masks[L] = acc > 0
        inp = np.maximum(acc, 0) if L < 8 else acc.copy()
    av = bias(PK.AUX_BIAS_V, 8)
    st.seg(8, 18, lambda t: inp[t >> 2][:, t & 3] if t < 64 else ed[t - 64], av)
    if masks is not None:
        masks[8] = av > 0
    assert st.pos == 145 * 16
    part = np.zeros((4, 64), np.float32)
    part[3] = alpha_part
    for c in range(3):
        for mo in range(8):
            for r in range(4):
                part[c] = part[c] + aux[PK.AUX_W_RGB + c * 128 + 16 * mo + 4 * g + r] * np.maximum(av[mo][:, r], 0)
    raw = np.zeros((16, 4), np.float32)
    for c in range(4):
        b = aux[PK.AUX_B_RGB + c] if c < 3 else aux[PK.AUX_B_ALPHA]
        raw[:, c] = part[c].reshape(4, 16).sum(0) + b
    return raw


def _embed_bwd16(X, G, nfreq):
    """csrc embed_bwd16: lane group g holds the gradients G[t] of its own encoding registers (t < 6 nfreq / 4: column
    q = per*g + t -> frequency q // 6, sin/cos, axis; t = per: identity column g); the four groups are added."""
    g = LANE >> 4
    per = 6 * nfreq // 4
    out = np.zeros((3, 64), np.float64)
    for ax in range(3):
        out[ax] = np.where(g == ax, G[per], 0.0)
    for t in range(per):
        L, sc, ax = _enc16_column(t, g, nfreq)
        f = np.exp2(L).astype(np.float32)
        arg = (X[LANE, ax] * f).astype(np.float32)
        c = f * G[t] * np.where(sc == 1, -np.sin(arg), np.cos(arg))
        for a in range(3):
            out[a] += np.where(ax == a, c, 0.0)
    return out.reshape(3, 4, 16).sum(1).T          # [16,3]


def mlp_bwd_pass16(packed16_fwd, stream_bwd16, masks, pts, dirs, g_raw):
    """Emulates csrc mlp_bwd_pass16 for one wave: g_raw [16,4] -> (dL/dpts [16,3], dL/ddirs [16,3])."""
    from neural_sim_nerf_amd import pack as PK
    st = Stream16(stream_bwd16)
    aux = packed16_fwd[PK.STREAM_SLABS * PK.SLAB_FLOATS:]
    g = LANE >> 4
    P = np.tile(pts, (4, 1)).astype(np.float32)
    V = np.tile(dirs, (4, 1)).astype(np.float32)
    G = np.tile(g_raw, (4, 1)).astype(np.float32)      # per lane
    regs = lambda arr: (lambda t: arr[t >> 2][:, t & 3])
    gv = np.zeros((8, 64, 4), np.float32)
    for mo in range(8):
        for r in range(4):
            f = 16 * mo + 4 * g + r
            v = (aux[PK.AUX_W_RGB + f] * G[:, 0] + aux[PK.AUX_W_RGB + 128 + f] * G[:, 1] + aux[PK.AUX_W_RGB + 256 + f] * G[:, 2])
            gv[mo][:, r] = np.where(masks[8][mo][:, r], v, 0)
    accv = np.zeros((18, 64, 4), np.float32)
    st.seg(18, 8, regs(gv), accv)
    dv = _embed_bwd16(V, [accv[16 + (t >> 2)][:, t & 3] for t in range(8)], 4)
    gin = accv[:16].copy()
    acc = np.zeros((20, 64, 4), np.float32)
    for idx in range(8):
        acc[:16] = 0
        st.seg(20 if idx == 3 else 16, 16, regs(gin), acc)
        if idx == 0:
            for mo in range(16):
                for r in range(4):
                    acc[mo][:, r] += aux[PK.AUX_W_ALPHA + 16 * mo + 4 * g + r] * G[:, 3]
        gin = np.where(masks[7 - idx], acc[:16], 0).astype(np.float32)
    genc = acc[16:]
    st.seg(4, 16, regs(gin), genc)
    assert st.pos == 145 * 16
    dp = _embed_bwd16(P, [genc[t >> 2][:, t & 3] for t in range(16)], 10)
    return dp, dv


# ----------------------------------------------------------------------------------------------------------
# bf16x3 forward pass (csrc/nsr_b3.inc, mlp_pass<.., B3 = true>): v_mfma_f32_32x32x16_bf16 modelled as
#     A[i][k] = a[lane = i + 32 (k // 8)][slot k % 8],  B[k][j] = b[lane = j + 32 (k // 8)][slot k % 8],
# same C/D fragment as the fp32 32x32 MFMA.  (Any other pairing of slots with k is equivalent as long as it is
# the same for A and B, which is what the kernel relies on.)
# ----------------------------------------------------------------------------------------------------------
B3_PIECE8 = [[0, 1, 1, 2], [0, 0, 2, 2], [0, 1, 1, 2], [0, 1, 1, 2], [0, 0, 2, 2], [0, 1, 1, 2]]      # B3Sched<8>
B3_BLOCK8 = [[0, 1, 2, 3], [1, 3, 0, 2], [2, 3, 4, 5], [4, 5, 6, 7], [5, 7, 4, 6], [6, 7, 0, 1]]
B3_PIECE4 = [[0, 1, 1, 2], [0, 0, 2, 2], [0, 1, 1, 2]]                                                # B3Sched<4>
B3_BLOCK4 = [[0, 1, 2, 3], [1, 3, 0, 2], [2, 3, 0, 1]]


def _bf16_bits_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def mfma_b3(a, b, acc):
    """a, b: [64, 8] bf16-valued floats (4 VGPRs each); acc: [64,16]."""
    A = a.reshape(2, 32, 8).transpose(1, 0, 2).reshape(32, 16)       # [i, k = 8 half + slot]
    B = b.reshape(2, 32, 8).transpose(0, 2, 1).reshape(16, 32)       # [k, j]
    D = A.astype(np.float64) @ B.astype(np.float64)
    out = acc.copy()
    out += D[ROW_OF, COL_OF].astype(np.float32)
    return out


def split_trunc(x):
    """split_pair of nsr_b3.inc: three truncation pieces of an fp32 array (exact sum), as bf16-valued floats."""
    x = np.ascontiguousarray(x, np.float32)
    m = np.uint32(0xffff0000)
    p0 = (x.view(np.uint32) & m).view(np.float32)
    r1 = (x - p0).astype(np.float32)
    p1 = (r1.view(np.uint32) & m).view(np.float32)
    r2 = (r1 - p1).astype(np.float32)
    p2 = (r2.view(np.uint32) & m).view(np.float32)
    assert np.array_equal(p2, r2)                                     # the third residual IS a bf16
    return p0, p1, p2


class StreamB3:
    """The bf16x3 stream as the kernel sees it: consecutive 1 KiB chunks of [64 lanes][8 bf16]."""

    def __init__(self, stream_floats):
        self.chunks = _bf16_bits_to_f32(stream_floats.view(np.uint16).reshape(-1, 64, 8))
        self.pos = 0

    def gemm(self, nmo, ngroups, src, acc, zero=False):
        """gemm_b3<NMO, NGROUPS, ZERO>: src(kb) -> [64, 8] fp32 (this lane's slots of k16 block kb)."""
        if zero:
            acc[:nmo] = 0
        piece, block = (B3_PIECE8, B3_BLOCK8) if nmo == 8 else (B3_PIECE4, B3_BLOCK4)
        spk = len(piece)
        nb = 12 // spk
        for g in range(ngroups):
            b = [split_trunc(src(nb * g + i)) for i in range(nb)]
            for G in range(12):
                s, bb = G % spk, b[G // spk]
                A = [self.chunks[self.pos + c] for c in range(4)]
                self.pos += 4
                for j in range(3):                                   # consume_b3: rounds, then chunks
                    for c in range(4):
                        if piece[s][c] + j <= 2:
                            mo = block[s][c]
                            acc[mo] = mfma_b3(A[c], bb[j], acc[mo])


def mlp_pass_b3(packed_b3, pts, dirs, masks=None):
    """pts, dirs: [32,3] -> raw [32,4]: one wave of mlp_pass<CAPTURE, true> (`masks`: filled like the fp32 emulation)."""
    from neural_sim_nerf_amd import pack as PK
    st = StreamB3(packed_b3[:PK.STREAM_SLABS_B3 * PK.SLAB_FLOATS])
    aux = packed_b3[PK.STREAM_SLABS_B3 * PK.SLAB_FLOATS:]
    h = LANE >> 5
    P = np.concatenate([pts, pts], 0).astype(np.float32)
    V = np.concatenate([dirs, dirs], 0).astype(np.float32)
    e = _encode(P, 10, 32)
    ed = _encode(V, 4, 16)
    enc_src = lambda kb: np.stack([e[8 * kb + i] for i in range(8)], 1)
    acc = load_bias(aux, PK.AUX_BIAS, 8)
    st.gemm(8, 2, enc_src, acc)
    if masks is not None:
        masks[0] = acc > 0
    inp = np.maximum(acc, 0)
    in_src = lambda kb: inp[kb >> 1][:, 8 * (kb & 1):8 * (kb & 1) + 8]
    alpha_part = np.zeros(64, np.float32)
    for L in range(1, 9):
        acc = load_bias(aux, PK.AUX_BIAS + L * 256, 8)
        if L == 5:
            st.gemm(8, 2, enc_src, acc)
        if L == 8:
            for tq in range(32):
                for kk in range(4):
                    w = aux[PK.AUX_W_ALPHA + (tq * 2 + h) * 4 + kk]
                    alpha_part = alpha_part + w * inp[(4 * tq + kk) >> 4][:, (4 * tq + kk) & 15]
        st.gemm(8, 8, in_src, acc)
        if masks is not None and L < 8:
            masks[L] = acc > 0
        inp = np.maximum(acc, 0) if L < 8 else acc.copy()
    av = load_bias(aux, PK.AUX_BIAS_V, 4)

    def v_src(kb):
        if kb < 16:
            return in_src(kb)
        if kb < 18:
            return np.stack([ed[8 * (kb - 16) + i] for i in range(8)], 1)
        return np.zeros((64, 8), np.float32)
    st.gemm(4, 5, v_src, av)
    if masks is not None:
        masks[8] = av > 0
    assert st.pos == PK.STREAM_SLABS_B3 * 16
    part = np.zeros((4, 64), np.float32)
    part[3] = alpha_part
    for c in range(3):
        for mo in range(4):
            for rq in range(4):
                for ri in range(4):
                    w = aux[PK.AUX_W_RGB + c * 128 + ((mo * 4 + rq) * 2 + h) * 4 + ri]
                    part[c] = part[c] + w * np.maximum(av[mo][:, rq * 4 + ri], 0)
    raw = np.zeros((32, 4), np.float32)
    for c in range(4):
        bias = aux[PK.AUX_B_RGB + c] if c < 3 else aux[PK.AUX_B_ALPHA]
        raw[:, c] = part[c][:32] + part[c][32:] + bias
    return raw


def mlp_bwd_pass_b3(packed_b3_fwd, stream_bwd_b3, masks, pts, dirs, g_raw):
    """One wave of mlp_bwd_pass<true>: g_raw [32,4] -> (dL/dpts [32,3], dL/ddirs [32,3])."""
    from neural_sim_nerf_amd import pack as PK
    st = StreamB3(stream_bwd_b3)
    aux = packed_b3_fwd[PK.STREAM_SLABS_B3 * PK.SLAB_FLOATS:]
    h = LANE >> 5
    P = np.concatenate([pts, pts], 0).astype(np.float32)
    V = np.concatenate([dirs, dirs], 0).astype(np.float32)
    G = np.concatenate([g_raw, g_raw], 0).astype(np.float32)
    gv = np.zeros((4, 64, 16), np.float32)
    for mo in range(4):
        for rq in range(4):
            for ri in range(4):
                idx = ((mo * 4 + rq) * 2 + h) * 4 + ri
                v = (aux[PK.AUX_W_RGB + idx] * G[:, 0] + aux[PK.AUX_W_RGB + 128 + idx] * G[:, 1]
                     + aux[PK.AUX_W_RGB + 256 + idx] * G[:, 2])
                gv[mo][:, rq * 4 + ri] = np.where(masks[8][mo][:, rq * 4 + ri], v, 0)
    frag_src = lambda arr: (lambda kb: arr[kb >> 1][:, 8 * (kb & 1):8 * (kb & 1) + 8])
    ae = np.zeros((4, 64, 16), np.float32)
    st.gemm(4, 2, frag_src(gv), ae, zero=True)
    dv = _embed_bwd(V, [ae[0][:, t] for t in range(16)], 4)
    acc = np.zeros((8, 64, 16), np.float32)
    st.gemm(8, 4, frag_src(gv), acc, zero=True)
    gin = acc.copy()
    dp5 = None
    for idx in range(8):
        if idx == 3:
            st.gemm(4, 4, frag_src(gin), ae, zero=True)
            dp5 = _embed_bwd(P, [ae[t >> 4][:, t & 15] for t in range(32)], 10)
        st.gemm(8, 8, frag_src(gin), acc, zero=True)
        if idx == 0:
            for tq in range(32):
                for kk in range(4):
                    t = 4 * tq + kk
                    acc[t >> 4][:, t & 15] += aux[PK.AUX_W_ALPHA + (tq * 2 + h) * 4 + kk] * G[:, 3]
        gin = np.where(masks[7 - idx], acc, 0).astype(np.float32)
    st.gemm(4, 4, frag_src(gin), ae, zero=True)
    assert st.pos == PK.STREAM_SLABS_B3_BWD * 16
    dp = _embed_bwd(P, [ae[t >> 4][:, t & 15] for t in range(32)], 10) + dp5
    return dp, dv


# ----------------------------------------------------------------------------------------------------------
# f16x2 (csrc/nsr_h2.inc): fp16 MFMAs on two-piece split operands with power-of-two range management
# ----------------------------------------------------------------------------------------------------------
H2_MAX = np.float32(65504.0 * (1.0 - 1.0 / 4096.0))


def split_h2(x, m):
    """split_pair_h2 of nsr_h2.inc: (hi, lo) fp16-valued floats of x * m, and max |x * m|"""
    t = (np.ascontiguousarray(x, np.float32) * np.float32(m)).astype(np.float32)
    with np.errstate(over="ignore", invalid="ignore"):
        hi = t.astype(np.float16).astype(np.float32)
        lo = (t - hi).astype(np.float32).astype(np.float16).astype(np.float32)
    return hi, lo, float(np.abs(t).max())


class StreamH2:
    """The f16x2 stream as the kernel sees it: consecutive 1 KiB chunks of [64 lanes][8 fp16]."""

    def __init__(self, stream_floats):
        self.chunks = stream_floats.view(np.float16).reshape(-1, 64, 8).astype(np.float32)
        self.pos = 0
        self.amax = np.zeros(64, np.float32)

    def gemm(self, nmo, nkb, src, scale, acc, zero=False):
        """gemm_h2<NMO, NKB, ZERO>: src(kb) -> [64, 8] fp32 (this lane's slots of k16 block kb), scale(kb) -> multiplier."""
        if zero:
            acc[:nmo] = 0
        for kb in range(nkb):
            t = (src(kb) * np.float32(scale(kb))).astype(np.float32)
            self.amax = np.maximum(self.amax, np.abs(t).max(1))
            hi, lo, _ = split_h2(src(kb), scale(kb))
            for st in range(nmo // 2):                               # one step: hi(HB), hi(HB+1), lo(LB), lo(LB+1)
                hb, lb = 2 * st, (2 * (st ^ 1) if nmo > 2 else 0)
                A = [self.chunks[self.pos + c] for c in range(4)]
                self.pos += 4
                for a, bb, mo in ((A[0], lo, hb), (A[1], lo, hb + 1), (A[2], hi, lb), (A[3], hi, lb + 1),
                                  (A[0], hi, hb), (A[1], hi, hb + 1)):
                    acc[mo] = mfma_b3(a, bb, acc[mo])                # same operand layout as the bf16 MFMA


def mlp_pass_h2(packed_h2, pts, dirs, masks=None):
    """pts, dirs: [32,3] -> raw [32,4]: one wave of mlp_pass<CAPTURE, kMlpH2>."""
    from neural_sim_nerf_amd import pack as PK
    st = StreamH2(packed_h2[:PK.STREAM_SLABS * PK.SLAB_FLOATS])
    aux = packed_h2[PK.STREAM_SLABS * PK.SLAB_FLOATS:]
    sc = aux[PK.AUX_H2_SCALE:PK.AUX_H2_SCALE + 16]
    h = LANE >> 5
    P = np.concatenate([pts, pts], 0).astype(np.float32)
    V = np.concatenate([dirs, dirs], 0).astype(np.float32)
    e = _encode(P, 10, 32)
    ed = _encode(V, 4, 16)
    enc_src = lambda kb: np.stack([e[8 * kb + i] for i in range(8)], 1)
    acc = load_bias(aux, PK.AUX_BIAS, 8)
    st.gemm(8, 4, enc_src, lambda kb: sc[0], acc)
    if masks is not None:
        masks[0] = acc > 0
    inp = np.maximum(acc, 0)
    in_src = lambda kb: inp[kb >> 1][:, 8 * (kb & 1):8 * (kb & 1) + 8]
    alpha_part = np.zeros(64, np.float32)
    for L in range(1, 9):
        acc = load_bias(aux, PK.AUX_BIAS + L * 256, 8)
        if L == 5:
            st.gemm(8, 4, enc_src, lambda kb: sc[9], acc)
        if L == 8:
            for tq in range(32):
                for kk in range(4):
                    w = aux[PK.AUX_W_ALPHA + (tq * 2 + h) * 4 + kk]
                    alpha_part = alpha_part + w * inp[(4 * tq + kk) >> 4][:, (4 * tq + kk) & 15]
        st.gemm(8, 16, in_src, lambda kb, L=L: sc[L], acc)
        if masks is not None and L < 8:
            masks[L] = acc > 0
        inp = np.maximum(acc, 0) if L < 8 else acc.copy()
    av = load_bias(aux, PK.AUX_BIAS_V, 4)
    v_src = lambda kb: in_src(kb) if kb < 16 else np.stack([ed[8 * (kb - 16) + i] for i in range(8)], 1)
    st.gemm(4, 18, v_src, lambda kb: sc[10] if kb < 16 else sc[11], av)
    if masks is not None:
        masks[8] = av > 0
    assert st.pos == PK.STREAM_SLABS * 16
    ov = np.where(st.amax > H2_MAX, np.float32(np.nan), np.float32(0))
    part = np.zeros((4, 64), np.float32)
    part[3] = alpha_part
    part += ov[None, :]
    for c in range(3):
        for mo in range(4):
            for rq in range(4):
                for ri in range(4):
                    w = aux[PK.AUX_W_RGB + c * 128 + ((mo * 4 + rq) * 2 + h) * 4 + ri]
                    part[c] = part[c] + w * np.maximum(av[mo][:, rq * 4 + ri], 0)
    raw = np.zeros((32, 4), np.float32)
    for c in range(4):
        bias = aux[PK.AUX_B_RGB + c] if c < 3 else aux[PK.AUX_B_ALPHA]
        raw[:, c] = part[c][:32] + part[c][32:] + bias
    return raw


def mlp_bwd_pass_h2(packed_h2_fwd, stream_bwd_h2, masks, pts, dirs, g_raw):
    """One wave of mlp_bwd_pass_h2 (csrc/nsr_h2_bwd.inc): g_raw [32,4] -> (dL/dpts [32,3], dL/ddirs [32,3])."""
    from neural_sim_nerf_amd import pack as PK
    st = StreamH2(stream_bwd_h2)
    aux = packed_h2_fwd[PK.STREAM_SLABS * PK.SLAB_FLOATS:]
    tb = aux[PK.AUX_H2_BWD:PK.AUX_H2_BWD + 16]
    h = LANE >> 5
    P = np.concatenate([pts, pts], 0).astype(np.float32)
    V = np.concatenate([dirs, dirs], 0).astype(np.float32)
    G = np.concatenate([g_raw, g_raw], 0).astype(np.float32)
    # per-point normalisation by a power of two (h2_norm_scale)
    m = np.abs(G).max(1)
    ef = (m.view(np.uint32) >> 23) & 0xff
    K = 7                                            # kH2GradLog2: the largest input lands in [2^7, 2^8)
    ok = (ef >= 1 + K) & (ef <= 253)
    s = np.where(ok, ((254 - ef.astype(np.int64) + K) << 23).astype(np.uint32).view(np.float32), np.float32(1))
    inv = np.where(ok, ((ef.astype(np.int64) - K) << 23).astype(np.uint32).view(np.float32), np.float32(1))
    G = (G * s[:, None]).astype(np.float32)
    gv = np.zeros((4, 64, 16), np.float32)
    for mo in range(4):
        for rq in range(4):
            for ri in range(4):
                idx = ((mo * 4 + rq) * 2 + h) * 4 + ri
                v = (aux[PK.AUX_W_RGB + idx] * G[:, 0] + aux[PK.AUX_W_RGB + 128 + idx] * G[:, 1]
                     + aux[PK.AUX_W_RGB + 256 + idx] * G[:, 2])
                gv[mo][:, rq * 4 + ri] = np.where(masks[8][mo][:, rq * 4 + ri], v, 0)
    frag_src = lambda arr: (lambda kb: arr[kb >> 1][:, 8 * (kb & 1):8 * (kb & 1) + 8])
    ae = np.zeros((2, 64, 16), np.float32)
    st.gemm(2, 8, frag_src(gv), lambda kb: tb[0], ae, zero=True)
    dv = _embed_bwd(V, [ae[0][:, t] for t in range(16)], 4) * (tb[11] * inv)[:32, None]
    acc = np.zeros((8, 64, 16), np.float32)
    st.gemm(8, 8, frag_src(gv), lambda kb: tb[0], acc, zero=True)
    gin = acc.copy()
    dp5 = None
    for idx in range(8):
        mb = tb[1 + idx]
        if idx == 3:
            st.gemm(2, 16, frag_src(gin), lambda kb: mb, ae, zero=True)
            dp5 = _embed_bwd(P, [ae[t >> 4][:, t & 15] for t in range(32)], 10) * tb[12]
        st.gemm(8, 16, frag_src(gin), lambda kb: mb, acc, zero=True)
        if idx == 0:
            for tq in range(32):
                for kk in range(4):
                    t = 4 * tq + kk
                    acc[t >> 4][:, t & 15] += aux[PK.AUX_W_ALPHA + (tq * 2 + h) * 4 + kk] * (G[:, 3] * tb[10])
        gin = np.where(masks[7 - idx], acc, 0).astype(np.float32)
    st.gemm(2, 16, frag_src(gin), lambda kb: tb[9], ae, zero=True)
    assert st.pos == PK.STREAM_SLABS_H2_BWD * 16
    dp = (_embed_bwd(P, [ae[t >> 4][:, t & 15] for t in range(32)], 10) * tb[13] + dp5) * inv[:32, None]
    ov = np.where(st.amax > H2_MAX, np.float32(np.nan), np.float32(0))
    ov = (ov[:32] + ov[32:])[:, None]
    return dp + ov, dv + ov
