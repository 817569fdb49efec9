"""Lane-level emulation (numpy, CPU) of the register-resident one-launch layer kernel (csrc/layer_rr.hip): the MFMA fragment
layouts of gfx950's v_mfma_f32_32x32x16_{f16,bf16}, the k-slot maps of the prepared weight fragments, the incidence
product that sums the activated edge rows per target, the per-row scales and their cross-lane moves.  One random tile is
pushed through the same sequence of wave-level operations the kernel issues and compared with the plain formula of the layer.
Run:  python scripts/emulate_layer_rr.py   (no GPU).  The index formulas here are the ones the kernel and its prepare kernel use.

Fragment layouts (guide: cdna_hip_programming.md 3, and csrc/layer_fused.hip which runs on them):
  A operand  lane l holds A[i = l & 31][k = 8 (l >> 5) + s], s = 0..7
  B operand  lane l holds B[k = 8 (l >> 5) + s][j = l & 31]
  C / D      lane l, register r holds C[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31]
"""
import numpy as np

LANES = np.arange(64)
LI, LH = LANES & 31, LANES >> 5


def crow(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def mfma(A, B, C):
    """A, B: [64, 8] per-lane operand slots; C: [64, 16] accumulator registers -> new C"""
    Am = np.zeros((32, 16)); Bm = np.zeros((16, 32))
    for l in range(64):
        for s in range(8):
            Am[l & 31, 8 * (l >> 5) + s] = A[l, s]
            Bm[8 * (l >> 5) + s, l & 31] = B[l, s]
    D = Am @ Bm
    out = C.copy()
    for l in range(64):
        for r in range(16):
            out[l, r] += D[crow(r, l >> 5), l & 31]
    return out


def acc_to_operand(acc, cc):
    """registers 8 cc .. 8 cc + 7 of an accumulator tile as one operand fragment (slot s <- register 8 cc + s): the C layout read as an
    A or B operand whose k-slot (cc, h, s) is the C row 16 cc + 8 (s >> 2) + 4 h + (s & 3)"""
    return acc[:, 8 * cc:8 * cc + 8].copy()


def kslot_feature(c, h, s):
    """feature index of k-slot s (lane half h) of chunk c when an operand is made from accumulator tiles of 32 features each"""
    return 32 * (c >> 1) + 16 * (c & 1) + 8 * (s >> 2) + 4 * h + (s & 3)


def bpermute(src_lane, data):
    return data[src_lane]


def main(seed=0):
    rng = np.random.default_rng(seed)
    d_x, d_id, d_ef, W = 28, 12, 4, 128
    KE = 2 * d_x + d_id + d_ef           # 72 -> 5 chunks
    NKE, NKS, NKX = 5, W // 16, 2
    nn = 29
    deg = rng.integers(0, 5, nn)
    deg[3] = 0
    while deg.sum() > 64:
        deg[rng.integers(0, nn)] = 1
    p = np.concatenate([[0], np.cumsum(deg)])       # seg_ptr relative to the tile's first edge
    ne = int(p[-1])
    n_all = 40
    x = rng.standard_normal((n_all, d_x))
    m0 = 5
    tgt = np.repeat(np.arange(nn), deg) + m0
    src = rng.integers(0, n_all, ne)
    ids = rng.standard_normal((ne, d_id)); ef = rng.standard_normal((ne, d_ef))
    We = rng.standard_normal((W, KE)); ce = rng.standard_normal(W)
    W0 = rng.standard_normal((W, d_x + W + 4)); c0 = rng.standard_normal(W)
    W1 = rng.standard_normal((W, W)); c1 = rng.standard_normal(W)

    # ---- reference
    Z = np.concatenate([x[tgt], x[src], ids, ef], 1)
    Y = np.maximum(Z @ We.T + ce, 0)
    S = np.zeros((nn, W))
    for e in range(ne):
        S[tgt[e] - m0] += Y[e]
    IN = np.concatenate([x[m0:m0 + nn], S, deg[:, None].astype(float), np.zeros((nn, 3))], 1)
    H = np.maximum(IN @ W0.T + c0, 0)
    OUT = np.maximum(H @ W1.T + c1, 0)

    # ---- prepared fragments (what the prepare kernel writes): frag[blk][chunk][lane][slot]
    def frag_rows(Wm, kmap, nblk, nchunk):
        f = np.zeros((nblk, nchunk, 64, 8))
        for b in range(nblk):
            for c in range(nchunk):
                for l in range(64):
                    for s in range(8):
                        k = kmap(c, l >> 5, s)
                        f[b, c, l, s] = Wm[32 * b + (l & 31), k] if k >= 0 else 0.0
        return f
    we_f = frag_rows(We, lambda c, h, s: (16 * c + 8 * h + s) if 16 * c + 8 * h + s < KE else -1, W // 32, NKE)

    def k0map(c, h, s):
        if c < NKS:
            return d_x + kslot_feature(c, h, s)                    # S columns of [x | S | deg 0 0 0]
        j = 16 * (c - NKS) + 8 * h + s                             # [x | deg 0 0 0] part
        if j < d_x:
            return j
        if j < d_x + 4:
            return d_x + W + (j - d_x)
        return -1
    w0_f = frag_rows(W0, k0map, W // 32, NKS + NKX)
    w1_f = frag_rows(W1, lambda c, h, s: kslot_feature(c, h, s), W // 32, NKS)

    # ---- edge stage: per 32-row edge block, A = gathered rows (lane (e, h): columns 16 c + 8 h + s of the concatenated row)
    sacc = [np.zeros((64, 16)) for _ in range(W // 32)]            # S^T tiles: C[row = feature in block][col = target]
    zpad = np.zeros((64, 16 * NKE)); zpad[:ne, :KE] = Z
    escale = np.exp2(rng.integers(-3, 4, 64)).astype(float)        # a per-edge power-of-two row scale (the inexact path), undone in the epilogue
    pt = np.zeros(64, dtype=int); pt1 = np.zeros(64, dtype=int)
    for l in range(64):
        t = l & 31
        if t < nn:
            pt[l], pt1[l] = p[t], p[t + 1]
    for eb in range(2):
        A = np.zeros((NKE, 64, 8))
        for c in range(NKE):
            for l in range(64):
                e = 32 * eb + (l & 31)
                A[c, l] = zpad[e, 16 * c + 8 * (l >> 5):16 * c + 8 * (l >> 5) + 8] * escale[e]
        # the row scale of edge e lives in lanes (e, 0) and (e, 1); the epilogue needs it per REGISTER: row crow(r, h) of the block
        inv_lane = 1.0 / escale[32 * eb + LI]
        inv_reg = np.stack([bpermute(crow(r, LH), inv_lane) for r in range(16)], 1)
        # incidence operand: lane (t, h), slot s of chunk cc <-> edge row 16 cc + 8 (s >> 2) + 4 h + (s & 3) of this block; 2.0 where it is an in-edge of t
        M = np.zeros((2, 64, 8))
        for cc in range(2):
            for l in range(64):
                for s in range(8):
                    e_abs = 32 * eb + 16 * cc + 8 * (s >> 2) + 4 * (l >> 5) + (s & 3)
                    M[cc, l, s] = 2.0 if pt[l] <= e_abs < pt1[l] else 0.0
        for fb in range(W // 32):
            acc = np.zeros((64, 16))
            for c in range(NKE):
                acc = mfma(A[c], we_f[fb, c], acc)                  # C[row = edge][col = feature 32 fb + li]
            y = np.maximum(acc * inv_reg + ce[32 * fb + LI][:, None], 0)
            for cc in range(2):
                sacc[fb] = mfma(acc_to_operand(y, cc), M[cc], sacc[fb])   # A = Y^T (i = feature, k = edge), B = M (k = edge, j = target)
    # ---- node stage 0 (transposed): A = W0 fragments (i = hidden feature), B = IN^T (k-slots from the S^T tiles and the x rows, j = target)
    rowscale = np.exp2(rng.integers(-2, 3, 64)).astype(float)[LI]   # per target (both halves equal)
    Bn = np.zeros((NKS + NKX, 64, 8))
    for fb in range(W // 32):
        for cc in range(2):
            Bn[2 * fb + cc] = acc_to_operand(sacc[fb] * 0.5, cc) * rowscale[:, None]
    for c in range(NKX):
        for l in range(64):
            t = l & 31
            for s in range(8):
                j = 16 * c + 8 * (l >> 5) + s
                v = 0.0
                if t < nn:
                    v = x[m0 + t, j] if j < d_x else (float(deg[t]) if j == d_x else 0.0)
                Bn[NKS + c, l, s] = v * rowscale[l]
    hacc = []
    for fbo in range(W // 32):
        # bias: C row = hidden feature (per register), col = target: init = c0[feature] * rowscale[target]
        acc = np.stack([c0[32 * fbo + crow(r, LH)] * rowscale for r in range(16)], 1)
        for c in range(NKS + NKX):
            acc = mfma(w0_f[fbo, c], Bn[c], acc)
        hacc.append(np.maximum(acc / rowscale[:, None], 0))
    # ---- node stage 1: A = H (i = target, k-slots from the H^T tiles), B = W1 fragments (j = output feature)
    hscale = np.exp2(rng.integers(-2, 3, 64)).astype(float)[LI]
    inv_reg = np.stack([bpermute(crow(r, LH), 1.0 / hscale) for r in range(16)], 1)
    out = np.zeros((32, W))
    for fb in range(W // 32):
        acc = np.zeros((64, 16))
        for fbo in range(W // 32):
            for cc in range(2):
                acc = mfma(acc_to_operand(hacc[fbo], cc) * hscale[:, None], w1_f[fb, 2 * fbo + cc], acc)
        o = np.maximum(acc * inv_reg + c1[32 * fb + LI][:, None], 0)
        for l in range(64):
            for r in range(16):
                out[crow(r, l >> 5), 32 * fb + (l & 31)] = o[l, r]
    err = np.abs(out[:nn] - OUT).max() / np.abs(OUT).max()
    print("tile nn=%d ne=%d  max rel err vs plain formula: %.2e" % (nn, ne, err))
    assert err < 1e-12
    return err


if __name__ == "__main__":
    for sd in range(3):
        main(sd)
    print("ok")
