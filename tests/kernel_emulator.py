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
    def __init__(self, packed):
        self.steps = packed[:290 * 2048].reshape(290, 8, 64, 4)
        self.aux = packed[290 * 2048:]
        self.pos = 0

    def next_step(self):
        s = self.steps[self.pos]
        self.pos += 1
        return s                       # [8 chunks][64 lanes][4]


def load_bias(aux, off, nmo):
    h = LANE >> 5
    acc = np.zeros((nmo, 64, 16), np.float32)
    for mo in range(nmo):
        for rq in range(4):
            base = off + ((mo * 4 + rq) * 2 + h) * 4
            for ri in range(4):
                acc[mo, :, rq * 4 + ri] = aux[base + ri]
    return acc


def step8(frag, bvals, acc):
    """frag [8][64][4]; bvals: 4 arrays [64]."""
    for kk in range(4):
        for mo in range(8):
            acc[mo] = mfma(frag[mo][:, kk], bvals[kk], acc[mo])


def step4(frag, bvals, acc):
    for q in range(2):
        for kk in range(4):
            for mo in range(4):
                acc[mo] = mfma(frag[q * 4 + mo][:, kk], bvals[q * 4 + kk], acc[mo])


def mlp_pass(packed, pts, dirs):
    """pts, dirs: [32,3] -> raw [32,4], emulating one wave (lane j and j+32 both own point j)."""
    from_aux = __import__("importlib").import_module("neural_sim_nerf_amd.pack")
    st = Stream(packed)
    aux = st.aux
    h = LANE >> 5
    P = np.concatenate([pts, pts], 0).astype(np.float32)      # per lane
    V = np.concatenate([dirs, dirs], 0).astype(np.float32)

    def encode(X, nfreq, n):
        e = np.zeros((n, 64), np.float32)
        for L in range(nfreq):
            for ax in range(3):
                arg = (X[:, ax] * np.float32(2 ** L)).astype(np.float32)
                e[3 * L + ax] = np.where(h == 1, np.cos(arg), np.sin(arg)).astype(np.float32)
        e[3 * nfreq] = np.where(h == 1, X[:, 2], X[:, 0])
        e[3 * nfreq + 1] = np.where(h == 1, 0.0, X[:, 1])
        return e

    e = encode(P, 10, 32)
    ed = encode(V, 4, 16)

    def seg_enc(acc):
        for tq in range(8):
            step8(st.next_step(), [e[4 * tq + kk] for kk in range(4)], acc)

    def seg_main(inp, acc):
        for tq in range(32):
            step8(st.next_step(), [inp[(4 * tq + kk) >> 4][:, (4 * tq + kk) & 15] for kk in range(4)], acc)

    acc = load_bias(aux, from_aux.AUX_BIAS, 8)
    seg_enc(acc)
    inp = np.maximum(acc, 0)
    alpha_part = np.zeros(64, np.float32)
    for L in range(1, 9):
        acc = load_bias(aux, from_aux.AUX_BIAS + L * 256, 8)
        if L == 5:
            seg_enc(acc)
        if L == 8:
            for tq in range(32):
                for kk in range(4):
                    w = aux[from_aux.AUX_W_ALPHA + (tq * 2 + h) * 4 + kk]
                    alpha_part = alpha_part + w * inp[(4 * tq + kk) >> 4][:, (4 * tq + kk) & 15]
        seg_main(inp, acc)
        inp = np.maximum(acc, 0) if L < 8 else acc.copy()
    av = load_bias(aux, from_aux.AUX_BIAS_V, 4)
    for s in range(18):
        if s < 16:
            b = [inp[(8 * s + i) >> 4][:, (8 * s + i) & 15] for i in range(8)]
        else:
            b = [ed[8 * (s - 16) + i] for i in range(8)]
        step4(st.next_step(), b, av)
    assert st.pos == 290
    part = np.zeros((4, 64), np.float32)
    part[3] = alpha_part
    for c in range(3):
        for mo in range(4):
            for rq in range(4):
                for ri in range(4):
                    w = aux[from_aux.AUX_W_RGB + c * 128 + ((mo * 4 + rq) * 2 + h) * 4 + ri]
                    part[c] = part[c] + w * np.maximum(av[mo][:, rq * 4 + ri], 0)
    raw = np.zeros((32, 4), np.float32)
    for c in range(4):
        bias = aux[from_aux.AUX_B_RGB + c] if c < 3 else aux[from_aux.AUX_B_ALPHA]
        raw[:, c] = part[c][:32] + part[c][32:] + bias
    return raw
