"""Design check (CPU): lane-level numpy emulation of the MFMA data flow used by the HIP kernels.

There is no GPU in the build container, so the index algebra of attention.hip / linear.hip (operand
k-slot convention, swapped QK^T / PV, LDS-DMA piece placement and the K-tile XOR swizzle, C/D register
-> (row, col) maps) is mirrored here expression by expression on 64 simulated lanes and compared with
plain matrix algebra.  v_mfma_f32_32x32x2_f32 semantics (cdna_hip_programming.md section 3):
    lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
    D[i][j] lives in lane (j + 32*((i>>2)&1)), register r with i = (r&3) + 8*(r>>2) + 4*(l>>5).
"""
import numpy as np

LANES = np.arange(64)
L31 = LANES & 31
H = LANES >> 5


def mfma_32x32x2(a, b, acc):
    """a, b: [64] per-lane operands; acc: [64,16] per-lane accumulators (updated copy returned)."""
    A = np.zeros((32, 2))
    B = np.zeros((2, 32))
    A[L31, H] = a
    B[H, L31] = b
    D = A @ B
    out = acc.copy()
    for r in range(16):
        rows = (r & 3) + 8 * (r >> 2) + 4 * H
        out[:, r] += D[rows, L31]
    return out


def test_mfma_emulation_matches_definition():
    rs = np.random.RandomState(0)
    A, B = rs.randn(32, 2), rs.randn(2, 32)
    acc = mfma_32x32x2(A[L31, H], B[H, L31], np.zeros((64, 16)))
    D = A @ B
    for lane in range(64):
        for r in range(16):
            i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
            assert np.isclose(acc[lane, r], D[i, lane & 31])


def test_linear_kernel_mapping():
    """linear.hip: wave tile 32 rows x 32 cols, K=128; A=X rows, B=W rows, k-slot (q,e,h) <-> channel 8q+4h+e."""
    rs = np.random.RandomState(1)
    K = 128
    X, W = rs.randn(32, K), rs.randn(32, K)
    acc = np.zeros((64, 16))
    for q in range(K // 8):
        av = np.stack([X[L31, 8 * q + 4 * H + e] for e in range(4)], 1)      # ds_read_b128 at xa + 8q (+4h)
        bv = np.stack([W[L31, 8 * q + 4 * H + e] for e in range(4)], 1)
        for e in range(4):
            acc = mfma_32x32x2(av[:, e], bv[:, e], acc)
    Y = X @ W.T
    for r in range(16):
        i = (r & 3) + 8 * (r >> 2) + 4 * H
        np.testing.assert_allclose(acc[:, r], Y[i, L31], rtol=1e-12, atol=1e-12)


def _lds_tiles_via_dma(K_tile, V_tile):
    """issue_tile_loads(): 16 pieces of 1 KiB (=2 rows); lane ℓ of piece i writes LDS chunk (i*64+ℓ) with
    K source chunk (ℓ&31) ^ (row&15) of row 2i+(ℓ>>5); V unswizzled."""
    Ks = np.zeros(32 * 128)
    Vs = np.zeros(32 * 128)
    for i in range(16):
        for lane in range(64):
            row = 2 * i + (lane >> 5)
            cph = lane & 31
            dst = i * 256 + lane * 4                     # floats: wave-uniform base + lane*16 B
            ksrc = (cph ^ (row & 15)) << 2
            Ks[dst:dst + 4] = K_tile[row, ksrc:ksrc + 4]
            Vs[dst:dst + 4] = V_tile[row, (cph << 2):(cph << 2) + 4]
    return Ks, Vs


def test_attention_wave_dataflow():
    """attention.hip for one wave (32 queries) over 3 key tiles incl. a ragged tail, vs dense softmax."""
    rs = np.random.RandomState(2)
    N = 80                                                  # 2 full tiles + 16-key tail
    C = 128
    Q = rs.randn(32, C) * 0.3                               # already scaled by log2(e)/sqrt(C)
    Kf = rs.randn(N, C)
    Vf = rs.randn(N, C)
    compat = rs.rand(32, 96)
    compat[:, N:] = np.nan                                  # padding must never leak (select, not multiply)

    qf = [np.stack([Q[L31, 8 * q + 4 * H + e] for e in range(4)], 1) for q in range(16)]
    o = [np.zeros((64, 16)) for _ in range(4)]
    m_run = np.full(64, -1.0e30)
    l_run = np.zeros(64)
    for kt in range(3):
        rows = np.minimum(kt * 32 + np.arange(32), N - 1)    # clamped tail rows
        Ks, Vs = _lds_tiles_via_dma(Kf[rows], Vf[rows])
        s = np.zeros((64, 16))
        for q in range(16):
            off = L31 * C + (((2 * q + H) ^ (L31 & 15)) << 2)
            ka = np.stack([Ks[off + e] for e in range(4)], 1)
            for e in range(4):
                s = mfma_32x32x2(ka[:, e], qf[q][:, e], s)
        x = np.zeros((64, 16))
        for r in range(16):
            g, e = r >> 2, r & 3
            cc = compat[L31, kt * 32 + 8 * g + 4 * H + e]      # crow + kt*32 + 8g (+4h), element e
            key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * H
            x[:, r] = np.where(key < N, cc * s[:, r], -np.inf)
        mloc = x.max(1)
        mloc = np.maximum(mloc, mloc[LANES ^ 32])
        m_new = np.maximum(m_run, mloc)
        alpha = np.exp2(m_run - m_new)
        l_run = l_run * alpha
        for c in range(4):
            o[c] = o[c] * alpha[:, None]
        m_run = m_new
        p = np.exp2(x - m_run[:, None])
        l_run = l_run + p.sum(1)
        for r in range(16):
            key = (r & 3) + 8 * (r >> 2) + 4 * H
            va = np.stack([Vs[key * C + 4 * L31 + c] for c in range(4)], 1)
            for c in range(4):
                o[c] = mfma_32x32x2(va[:, c], p[:, r], o[c])
    l_tot = l_run + l_run[LANES ^ 32]
    out = np.zeros((32, C))
    for lane in range(64):
        for r in range(16):
            i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
            for c in range(4):
                out[lane & 31, 4 * i + c] = o[c][lane, r] / l_tot[lane]
    # both halves of a query's lane pair write disjoint channel sets: every channel written exactly once
    S = (Q @ Kf.T) * compat[:, :N]
    P = np.exp2(S - S.max(1, keepdims=True))
    ref = (P / P.sum(1, keepdims=True)) @ Vf
    np.testing.assert_allclose(out, ref, rtol=1e-9, atol=1e-9)


def test_k_swizzle_is_bank_conflict_free():
    """ds_read_b128 lane groups (MI355X_MICROARCH.md LDS table) must touch 16 distinct 16-B bank slots."""
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
              list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
    groups += [[l + 32 for l in g] for g in groups]
    for q in range(16):
        for g in groups:
            lanes = np.array(g)
            off_bytes = ((lanes & 31) * 128 + ((((2 * q + (lanes >> 5)) ^ (lanes & 15))) << 2)) * 4
            slots = (off_bytes // 16) % 16
            assert len(set(slots.tolist())) == 16
