"""Lane-level numpy emulation of the matrix-core plumbing of gps-gaussian_amd/csrc/gsr_composite_mfma.hip (no GPU needed).

Mirrors, index for index, what the kernels do with v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 / v_permlane32_swap and the
transposed LDS staging, using the operand layouts documented in /opt/skills/guides/cdna_hip_programming.md section 3, and checks
the results against direct evaluation.  It validates the index arithmetic of the design; gsr_selftest() validates the same
device functions on the hardware.
"""
import numpy as np

L = np.arange(64)


def mfma_32x32x2(a, b, acc):
    """a, b: [64] per-lane operands; acc: [64,16].  A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; reg v of lane l = D[(v//4)*8 + (l//32)*4 + v%4][l%32]."""
    A = np.zeros((32, 2), np.float64); B = np.zeros((2, 32), np.float64)
    A[L & 31, L >> 5] = a; B[L >> 5, L & 31] = b
    D = A @ B
    out = acc.copy()
    for v in range(16):
        out[:, v] += D[(v // 4) * 8 + (L // 32) * 4 + v % 4, L % 32]
    return out


def mfma_16x16x4(a, b, acc):
    """A[l&15][k=l>>4], B[k=l>>4][l&15]; reg r of lane l = D[4*(l>>4) + r][l&15]."""
    A = np.zeros((16, 4), np.float64); B = np.zeros((4, 16), np.float64)
    A[L & 15, L >> 4] = a; B[L >> 4, L & 15] = b
    D = A @ B
    out = acc.copy()
    for r in range(4):
        out[:, r] += D[4 * (L >> 4) + r, L & 15]
    return out


def swap32(a, b):
    """v_permlane32_swap: -> ([a_lo | b_lo], [a_hi | b_hi])"""
    return np.concatenate([a[:32], b[:32]]), np.concatenate([a[32:], b[32:]])


def mono_at(m, p):
    u = (p & 7) - 3.5; v = (p >> 3) - 3.5
    return [np.ones_like(u), u, v, u * u, u * v, v * v][m]


def test_power_tiles(rng):
    cx, cy = 515.5, 259.5
    x = cx + (rng.random(64) - .5) * 30; y = cy + (rng.random(64) - .5) * 30
    A = .05 + 3 * rng.random(64); C = .05 + 3 * rng.random(64); B = (rng.random(64) - .5) * 1.8 * np.sqrt(A * C)
    X, Y = x - cx, y - cy
    cu, cv = A * X + B * Y, B * X + C * Y
    coef = [-.5 * (X * cu + Y * cv), cu, cv, -.5 * A, -B, -.5 * C]       # lane l holds splat l's coefficients
    hi = [np.rint(c * 2048) / 2048 for c in coef]; lo = [c - h for c, h in zip(coef, hi)]
    opA = np.zeros((2, 6, 64))
    for t in range(3):
        opA[0, t], opA[1, t] = swap32(hi[2 * t], hi[2 * t + 1])
        opA[0, 3 + t], opA[1, 3 + t] = swap32(lo[2 * t], lo[2 * t + 1])
    monoB = np.zeros((2, 3, 64))
    for h in range(2):
        for t in range(3):
            monoB[h, t] = np.where(L >> 5 == 0, mono_at(2 * t, 32 * h + (L & 31)), mono_at(2 * t + 1, 32 * h + (L & 31)))
    tiles = {}
    for S in range(2):
        for h in range(2):
            acc = np.zeros((64, 16))
            for t in range(6):
                acc = mfma_32x32x2(opA[S, t], monoB[h, t % 3], acc)
            tiles[S, h] = acc
    P = np.zeros((64, 64))  # [staged splat j][pixel = lane]
    for gq in range(8):
        S, q = gq >> 2, gq & 3
        for e in range(4):
            lo_pair, hi_pair = swap32(tiles[S, 0][:, 4 * q + e], tiles[S, 1][:, 4 * q + e])
            P[8 * gq + e] = lo_pair; P[8 * gq + e + 4] = hi_pair
    px = cx - 3.5 + (L & 7); py = cy - 3.5 + (L >> 3)
    ref = np.zeros((64, 64))
    for j in range(64):
        dx, dy = x[j] - px, y[j] - py
        ref[j] = -.5 * (A[j] * dx * dx + C[j] * dy * dy) - B[j] * dx * dy
    err = np.abs(P - ref).max()
    assert err < 1e-9, err
    return err


XT_K, XT_C = 320, 20


def test_reduction(rng):
    d = rng.standard_normal((3, 64))
    w = rng.random((8, 64)); s = rng.standard_normal((8, 64))      # [splat e][pixel]
    sX = np.full(4 * XT_K, np.nan)
    xw = (L >> 4) * XT_K + (L & 15)
    for e in range(8):
        sX[xw + XT_C * e] = w[e]; sX[xw + XT_C * (8 + e)] = s[e]
    i, k = L & 15, L >> 4
    RA = np.zeros((16, 64))
    for t in range(16):
        pix = 16 * k + t
        m = np.zeros(64)
        for row in range(3, 9):
            m = np.where(i == row, mono_at(row - 3, pix), m)
        dv = d[np.minimum(i, 2), pix]
        RA[t] = np.where(i < 3, dv, m)
    base = (L >> 4) * XT_K + (L & 15) * XT_C
    bv = np.stack([sX[base + t] for t in range(16)])
    bv = np.nan_to_num(bv)
    acc = np.zeros((64, 4))
    for t in range(16):
        acc = mfma_16x16x4(RA[t], bv[t], acc)
    accF = np.zeros((8, 12))
    for l in range(64):
        c = l & 7
        if l < 8: accF[c, 0:3] = acc[l, 0:3]
        if 8 <= l < 16: accF[c, 3] = acc[l, 3]
        if 24 <= l < 32: accF[c, 4:8] = acc[l]
        if 40 <= l < 48: accF[c, 8] = acc[l, 0]
    u = (L & 7) - 3.5; v = (L >> 3) - 3.5
    for e in range(8):
        ref = [(w[e] * d[0]).sum(), (w[e] * d[1]).sum(), (w[e] * d[2]).sum(), s[e].sum(), (s[e] * u).sum(), (s[e] * v).sum(),
               (s[e] * u * u).sum(), (s[e] * u * v).sum(), (s[e] * v * v).sum()]
        assert np.allclose(accF[e, :9], ref, atol=1e-9), (e, accF[e, :9], ref)
    # the flush: moments about the bin centre -> about the splat centre
    X, Y = 5.3, -2.1
    m0, mu, mv, muu, muv, mvv = accF[0, 3:9]
    Sx, Sy = X * m0 - mu, Y * m0 - mv
    Sxx, Sxy, Syy = X * (Sx - mu) + muu, X * Sy - Y * mu + muv, Y * (Sy - mv) + mvv
    dx, dy = X - u, Y - v
    assert np.allclose([Sx, Sy, Sxx, Sxy, Syy], [(s[0] * dx).sum(), (s[0] * dy).sum(), (s[0] * dx * dx).sum(), (s[0] * dx * dy).sum(), (s[0] * dy * dy).sum()])
    # LDS bank check of the operand fetch: ds_read_b128 is serviced in these 16-lane groups; 64 banks of 4 bytes
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
    groups += [[l + 32 for l in g] for g in groups]
    for q in range(4):
        for g in groups:
            banks = set()
            for l in g:
                for wd in range(4):
                    banks.add((base[l] + 4 * q + wd) % 64)
            assert len(banks) == 64, "bank conflict in the operand fetch"
    return True


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    print("power tiles: max |err| = %.2e" % test_power_tiles(rng))
    print("reduction + park + flush + bank check:", test_reduction(rng))
