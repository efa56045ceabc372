"""Lane-level model of the MFMA operand enumeration of csrc/single_query_wave.hip (the wave-local single-query key passes):
v_mfma_f32_16x16x4_f32 semantics  D[row = 4 (lane >> 4) + r][col = lane & 15] += sum_{k = lane >> 4} A[m = lane & 15][k] B[k][n = lane & 15]
executed in numpy with the kernel's index expressions, against the plain matrix formulas.  Checks the transposed projection, the
score / dp products with register B operands, the dX product (+ the p . dxbar k-step) and the key-contracting weight gradient."""
import numpy as np

LD, XLD, E, H, HD = 68, 80, 60, 4, 15


def mfma(a, b, c):
    """a, b: (64,) per-lane scalars; c: (64, 4) accumulators -> new c"""
    A = np.zeros((16, 4)); Bm = np.zeros((4, 16))
    for lane in range(64):
        A[lane & 15, lane >> 4] = a[lane]
        Bm[lane >> 4, lane & 15] = b[lane]
    D = A @ Bm
    out = c.copy()
    for lane in range(64):
        for r in range(4):
            out[lane, r] += D[4 * (lane >> 4) + r, lane & 15]
    return out


def test_wave_tile_products():
    rs = np.random.RandomState(0)
    W = np.zeros((64, LD)); W[:E, :E] = rs.randn(E, E)
    WT = np.zeros((64, LD)); WT[:64, :64] = W[:64, :64].T
    X = np.zeros((16, 64)); X[:, :E] = rs.randn(16, E)            # the wave's 16 key rows
    Qm = np.zeros((16, LD)); Dm = np.zeros((16, LD))
    q = rs.randn(H, HD)
    for h in range(H):
        Qm[h, h * HD:(h + 1) * HD] = q[h]
    Dm[:H, :E] = rs.randn(H, E)
    lanes = np.arange(64); li, g = lanes & 15, lanes >> 4
    xk = [[X[li, ct * 16 + g * 4 + e] for e in range(4)] for ct in range(4)]       # SqwKey.x[ct].{x,y,z,w}
    # ---- projection (sqw_project_rope without bias / rotation)
    acc = [np.zeros((64, 4)) for _ in range(4)]
    for jt in range(4):
        for e in range(4):
            for ct in range(4):
                a = W[ct * 16 + li, jt * 16 + g * 4 + e]
                acc[ct] = mfma(a, xk[jt][e], acc[ct])
    T = X @ W[:64, :64].T                                           # T[key][c]
    for ct in range(4):
        for r in range(4):
            np.testing.assert_allclose(acc[ct][:, r], T[li, ct * 16 + g * 4 + r], rtol=1e-12, atol=1e-12)
    # ---- scores / dp (sqw_heads_dot): rows = heads in the g == 0 lanes
    def heads_dot(M, v):
        s = np.zeros((64, 4))
        for ct in range(4):
            for e in range(4):
                s = mfma(M[li, ct * 16 + g * 4 + e], v[ct][:, e], s)
        return s
    sc = heads_dot(Qm, acc)
    ref = T @ Qm[:4, :64].T                                         # [key][h]
    for h in range(4):
        np.testing.assert_allclose(sc[:16, h], ref[:, h], rtol=1e-12, atol=1e-12)      # lanes 0..15 = g == 0
    xv = [np.stack(xk[ct], axis=1) for ct in range(4)]
    dp = heads_dot(Dm, xv)
    refd = X @ Dm[:4, :64].T
    for h in range(4):
        np.testing.assert_allclose(dp[:16, h], refd[:, h], rtol=1e-12, atol=1e-12)
    # ---- dX^T = W^T G (+ p . dxbar as one more k-step)
    G = [rs.randn(64, 4) for _ in range(4)]                          # G[key li][ct * 16 + g * 4 + r] per lane
    Gm = np.zeros((16, 64))
    for ct in range(4):
        for r in range(4):
            Gm[li, ct * 16 + g * 4 + r] = G[ct][:, r]
    p = rs.rand(16, 4)                                               # p[key][h]
    pg = p[li, g]
    dxa = [mfma(Dm[g, mt * 16 + li], pg, np.zeros((64, 4))) for mt in range(4)]
    for ct in range(4):
        for e in range(4):
            for mt in range(4):
                dxa[mt] = mfma(WT[mt * 16 + li, ct * 16 + g * 4 + e], G[ct][:, e], dxa[mt])
    refx = Gm @ W[:64, :64] + p @ Dm[:4, :64]                        # dX[key][cin]
    for mt in range(4):
        for r in range(4):
            np.testing.assert_allclose(dxa[mt][:, r], refx[li, mt * 16 + g * 4 + r], rtol=1e-12, atol=1e-12)
    # ---- dW[c][cin] = sum_key G[key][c] X[key][cin] through the wave's LDS tiles
    gw = np.zeros((16, XLD)); xw = np.zeros((16, XLD))
    for ct in range(4):
        for e in range(4):
            gw[li, ct * 16 + g * 4 + e] = G[ct][:, e]
            xw[li, ct * 16 + g * 4 + e] = xk[ct][e]
    wacc = [[np.zeros((64, 4)) for _ in range(4)] for _ in range(4)]
    for s in range(4):
        for ct in range(4):
            for kt in range(4):
                wacc[ct][kt] = mfma(gw[4 * s + g, ct * 16 + li], xw[4 * s + g, kt * 16 + li], wacc[ct][kt])
    refw = Gm.T @ X
    for ct in range(4):
        for kt in range(4):
            for r in range(4):
                np.testing.assert_allclose(wacc[ct][kt][:, r], refw[ct * 16 + g * 4 + r, kt * 16 + li], rtol=1e-12, atol=1e-12)
