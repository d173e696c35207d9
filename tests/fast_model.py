"""Lane-level numpy model of the specialised long-block kernel (lw_kernels_long.hip), n = 2048.

One "wave" = 64 lanes; every per-lane register is a numpy array of shape [64]; LDS is a flat float32
array addressed in dwords.  The model follows the kernel's exact data movement (register layouts B/C/D/E,
the bpermute exchange, the XOR/transposing LDS swizzles, the re-ordered twiddle images) and exact f32
operation order, so (a) it documents the design, (b) it pins the index algebra on the CPU: its output must
be bit-identical to the oracle's sequential transform, and (c) it consumes the product's own LDS table image
(lw_debug_fast_image) so the host-side image builder is tested without a GPU.
"""
import numpy as np

F = np.float32
N = 2048
N2, N4, N8 = N // 2, N // 4, N // 8
LANES = np.arange(64)


def rev_bits(x, nbits):
    r = np.zeros_like(x)
    for i in range(nbits):
        r |= ((x >> i) & 1) << (nbits - 1 - i)
    return r


# ---- LDS slot functions (in units of pairs = 8 bytes) ------------------------------------------
def slot_t2(p):
    """T2: written in layout B (lane = p[5:0], reg = p[8:6]), read in layout C."""
    return p ^ (((p >> 6) & 3) << 3)


def slot_t3(p):
    """T3: written in layout C, read in layout D (lane nu = p[8:3], reg z = p[2:0])."""
    nu, z = p >> 3, p & 7
    return 8 * nu + (z ^ ((nu >> 2) & 7))


def slot_t4(p):
    """T4: written in layout D, read by the bit-reverse gather of layout E."""
    return (p & ~127) | ((p & 3) << 5) | ((p >> 2) & 31)


class Image:
    """Views into the product's LDS image (see lw_fast.hpp for the layout)."""

    def __init__(self, blob, offs):
        f = np.frombuffer(blob, np.float32)
        o = {k: v // 4 for k, v in offs.items()}
        self.apair = f[o["apair"]: o["apair"] + 1024].reshape(512, 2)
        self.tw_s2 = f[o["tw_s2"]: o["tw_s2"] + 512].reshape(4, 64, 2)
        self.tw_l0 = f[o["tw_l0"]: o["tw_l0"] + 256].reshape(2, 64, 2)
        self.tw_l1 = f[o["tw_l1"]: o["tw_l1"] + 128].reshape(64, 2)
        self.tw_l2 = f[o["tw_l2"]: o["tw_l2"] + 64].reshape(4, 8, 2)
        self.tw_l3 = f[o["tw_l3"]: o["tw_l3"] + 32].reshape(2, 8, 2)
        self.tw_l4 = f[o["tw_l4"]: o["tw_l4"] + 16].reshape(8, 2)
        self.a2 = f[o["a2"]]
        self.c4 = f[o["c4"]: o["c4"] + 512].reshape(2, 64, 4)
        self.b_lo = f[o["b_lo"]: o["b_lo"] + 512].reshape(2, 64, 4)
        self.b_hi = f[o["b_hi"]: o["b_hi"] + 512].reshape(2, 64, 4)
        self.win = f[o["win"]: o["win"] + 1024].reshape(2, 64, 8)
        self.inv_db = f[o["inv_db"]: o["inv_db"] + 256]


def bfly(H, L, t0, t1):
    """imdct.rs:36-41 on pairs (e0, e1): H, L are [2][64] arrays (index 0 = e0 = u[hi-1], 1 = e1 = u[hi])."""
    k00 = H[1] - L[1]
    k01 = H[0] - L[0]
    H1 = H[1] + L[1]
    H0 = H[0] + L[0]
    L1 = k00 * t0 - k01 * t1
    L0 = k01 * t0 + k00 * t1
    return np.stack([H0, H1]), np.stack([L0, L1])


def imdct_wave(X, img):
    """X: spectrum [1024] f32 of one channel.  Returns (pa, pb) per lane: dict q -> value arrays, and the
    full time-domain block assembled from them (for comparison with the oracle)."""
    X = np.asarray(X, F)
    lam = LANES
    # ---- load layout: lane holds float4 groups m = 64x + lane, x = 0..3
    P = [None] * 8  # layout B registers: P[x] = pair p = 64x + lane, as [2][64] (e0, e1)
    up = [None] * 4
    for x in range(4):
        m = 64 * x + lam
        X0, X1, X2, X3 = X[4 * m], X[4 * m + 1], X[4 * m + 2], X[4 * m + 3]
        au = img.apair[m]          # (A[2m], A[2m+1])
        al = img.apair[511 - m]    # (A[1022-2m], A[1023-2m])
        # upper pair 511 - m (imdct.rs:356-357)
        u1 = X0 * au[:, 0] - X2 * au[:, 1]
        u0 = X0 * au[:, 1] + X2 * au[:, 0]
        up[x] = np.stack([u0, u1])
        # lower pair m (imdct.rs:365-366)
        l1 = (-X3) * al[:, 0] - (-X1) * al[:, 1]
        l0 = (-X3) * al[:, 1] + (-X1) * al[:, 0]
        P[x] = np.stack([l0, l1])
    # bpermute: lane receives from lane 63 - lane; register x' = 7 - x_src
    for xs in range(4):
        P[7 - xs] = up[xs][:, 63 - lam]
    # ---- step 2 (imdct.rs:385-430): pairs x and x + 4
    for x in range(4):
        t = img.tw_s2[x]
        P[x + 4], P[x] = bfly(P[x + 4], P[x], t[:, 0], t[:, 1])
    # ---- stage l = 0 (pair bit 7): hi x in {2,3,6,7}, lo = x - 2
    for x in (2, 3, 6, 7):
        t = img.tw_l0[x & 1]
        P[x], P[x - 2] = bfly(P[x], P[x - 2], t[:, 0], t[:, 1])
    # ---- stage l = 1 (pair bit 6): hi x odd
    for x in (1, 3, 5, 7):
        t = img.tw_l1
        P[x], P[x - 1] = bfly(P[x], P[x - 1], t[:, 0], t[:, 1])
    # ---- T2: layout B -> C through LDS
    lds = np.zeros((576, 2), F)
    for x in range(8):
        lds[slot_t2(64 * x + lam)] = P[x].T
    Xc, lo3 = lam >> 3, lam & 7
    Q = [lds[slot_t2(64 * Xc + 8 * y + lo3)].T.copy() for y in range(8)]  # regs y = p[5:3]
    # ---- stages l = 2,3,4 (pair bits 5,4,3)
    for y in (4, 5, 6, 7):
        t = img.tw_l2[y & 3][lo3]
        Q[y], Q[y - 4] = bfly(Q[y], Q[y - 4], t[:, 0], t[:, 1])
    for y in (2, 3, 6, 7):
        t = img.tw_l3[y & 1][lo3]
        Q[y], Q[y - 2] = bfly(Q[y], Q[y - 2], t[:, 0], t[:, 1])
    for y in (1, 3, 5, 7):
        t = img.tw_l4[lo3]
        Q[y], Q[y - 1] = bfly(Q[y], Q[y - 1], t[:, 0], t[:, 1])
    # ---- T3: layout C -> D
    for y in range(8):
        lds[slot_t3(64 * Xc + 8 * y + lo3)] = Q[y].T
    z = np.zeros((16, 64), F)  # z[k] = u[16 nu + k]
    for zz in range(8):
        pr = lds[slot_t3(8 * lam + zz)]
        z[2 * zz], z[2 * zz + 1] = pr[:, 0], pr[:, 1]
    # ---- fused last three stages (imdct.rs:234-288), lane local
    a2 = img.a2
    k00 = z[15] - z[7]; k11 = z[14] - z[6]; z[15] = z[15] + z[7]; z[14] = z[14] + z[6]; z[7] = k00; z[6] = k11
    k00 = z[13] - z[5]; k11 = z[12] - z[4]; z[13] = z[13] + z[5]; z[12] = z[12] + z[4]
    z[5] = (k00 + k11) * a2; z[4] = (k11 - k00) * a2
    k00 = z[3] - z[11]; k11 = z[10] - z[2]; z[11] = z[11] + z[3]; z[10] = z[10] + z[2]; z[3] = k11; z[2] = k00
    k00 = z[1] - z[9]; k11 = z[8] - z[0]; z[9] = z[9] + z[1]; z[8] = z[8] + z[0]
    z[1] = (k00 + k11) * a2; z[0] = (k00 - k11) * a2
    for b in (8, 0):
        w = z[b:b + 8]
        k00 = w[7] - w[3]; y0 = w[7] + w[3]; y2 = w[5] + w[1]; k22 = w[5] - w[1]
        k33 = w[4] - w[0]; k11 = w[6] - w[2]; y1 = w[6] + w[2]; y3 = w[4] + w[0]
        w7, w5, w3, w1 = y0 + y2, y0 - y2, k00 + k33, k00 - k33
        w6, w4, w2, w0 = y1 + y3, y1 - y3, k11 - k22, k11 + k22
        z[b:b + 8] = np.stack([w0, w1, w2, w3, w4, w5, w6, w7])
    # ---- T4: layout D -> bit-reversed gather
    for zz in range(8):
        lds[slot_t4(8 * lam + zz)] = np.stack([z[2 * zz], z[2 * zz + 1]]).T
    pa, pb = {}, {}
    for c in range(2):
        mp = 2 * lam + c                       # m' handled by this lane
        v = rev_bits(mp, 7)
        q2 = 2 * v
        pq = lds[slot_t4(q2)]                  # pair 2v      = u[4v], u[4v+1]
        pq256 = lds[slot_t4(q2 + 256)]         # pair 2v+256
        p255 = lds[slot_t4(255 - q2)]          # pair 255-2v  = u[4(127-v)+2], +3
        p511 = lds[slot_t4(511 - q2)]          # pair 511-2v
        D = [p511[:, 1], p511[:, 0], p255[:, 1], p255[:, 0]]
        E = [pq256[:, 1], pq256[:, 0], pq[:, 1], pq[:, 0]]
        C = img.c4[c]
        # step 7 (imdct.rs:547-579)
        a02 = D[0] - E[2]; a11 = D[1] + E[3]
        b0 = C[:, 1] * a02 + C[:, 0] * a11; b1 = C[:, 1] * a11 - C[:, 0] * a02
        b2 = D[0] + E[2]; b3 = D[1] - E[3]
        D0, D1, E2, E3 = b2 + b0, b3 + b1, b2 - b0, b1 - b3
        a02 = D[2] - E[0]; a11 = D[3] + E[1]
        b0 = C[:, 3] * a02 + C[:, 2] * a11; b1 = C[:, 3] * a11 - C[:, 2] * a02
        b2 = D[2] + E[0]; b3 = D[3] - E[1]
        D2, D3, E0, E1 = b2 + b0, b3 + b1, b2 - b0, b1 - b3
        # step 8 (imdct.rs:618-657): w pairs 2m', 2m'+1 (from D) and 510-2m', 511-2m' (from E)
        Bl, Bh = img.b_lo[c], img.b_hi[c]
        for (w0, w1, bc, bs, qq) in ((D0, D1, Bl[:, 0], Bl[:, 1], 511 - 2 * mp), (D2, D3, Bl[:, 2], Bl[:, 3], 510 - 2 * mp),
                                     (E0, E1, Bh[:, 0], Bh[:, 1], 1 + 2 * mp), (E2, E3, Bh[:, 2], Bh[:, 3], 2 * mp)):
            a = w0 * bs - w1 * bc
            b = (-w0) * bc - w1 * bs
            for l in range(64):
                pa[int(qq[l])] = a[l]
                pb[int(qq[l])] = b[l]
    out = np.zeros(N, F)
    for q in range(N4):
        out[q] = pa[q]
        out[N2 - 1 - q] = -pa[q]
        out[N2 + q] = pb[q]
        out[N - 1 - q] = pb[q]
    return out


def floor_lane_model(rec, xs, img_inv_db):
    """Per-bin floor through the segment table the kernel builds (k_long's floor_table / floor_bin): entry {dy, c0, 1/adx, w},
    y(k) = ((bits(fma(fma(k, dy, c0), 1/adx, w)) & 0x7fc) >> 2) - 1 with w = 2^21 + 1 + y_base."""
    Fp = len(xs)
    act = np.array([(rec[i] & 0x8000) != 0 for i in range(Fp)])
    y = np.array([int(rec[i] & 0xFF) for i in range(Fp)])
    k = np.arange(N2)
    sid = np.searchsorted(np.array(xs), k, side="right") - 1
    out = np.zeros(N2, F)
    for s in range(Fp):
        lo = max(i for i in range(s + 1) if act[i])
        his = [i for i in range(s + 1, Fp) if act[i]]
        if his:
            hi = his[0]
            x0, x1, y0, y1, adx = xs[lo], xs[hi], y[lo], y[hi], xs[hi] - xs[lo]
        else:
            x0, x1, y0, y1, adx = xs[lo], xs[lo], y[lo], y[lo], 1
        dy = y1 - y0
        sel = k[sid == s]
        if len(sel) == 0:
            continue
        rinv = F(1.0) / F(adx)
        if dy < 0:
            c0 = F(F(F(0.875) * F(adx)) - F(0.5)) - F(F(x1) * F(dy))
            w = F(y1) + F(2097153.0)
        else:
            c0 = F(F(0.5) - F(F(0.125) * F(adx))) - F(F(x0) * F(dy))
            w = F(y0) + F(2097153.0)
        z = sel.astype(np.float64) * float(dy) + float(c0)          # fma: one rounding; the sum is exact in f32 anyway
        assert np.array_equal(z.astype(F).astype(np.float64), z)
        t = (z * float(rinv) + float(w)).astype(F)                   # fma (the f64 product is exact; see test_fast_model)
        yk = ((t.view(np.uint32) & 0x7FC) >> 2).astype(np.int64) - 1
        out[sel] = img_inv_db[yk]
    return out


# =================================================================================================
# Packed-op formulation (v2 of the kernel): every arithmetic step as v_pk_mul_f32 / v_pk_add_f32 with
# op_sel / op_sel_hi / neg_lo / neg_hi modifiers.  A "register pair" is an array [2][64] (lo, hi).
# Semantics (VOP3P): res.lo = f(s0[op_sel[0]], s1[op_sel[1]]) with neg_lo applied per source,
#                    res.hi = f(s0[op_sel_hi[0]], s1[op_sel_hi[1]]) with neg_hi applied per source.
# =================================================================================================
def _src(a, sel, neg):
    v = a[sel]
    return -v if neg else v


def pk(op, a, b, sel=(0, 0), selhi=(1, 1), nlo=(0, 0), nhi=(0, 0)):
    lo0, lo1 = _src(a, sel[0], nlo[0]), _src(b, sel[1], nlo[1])
    hi0, hi1 = _src(a, selhi[0], nhi[0]), _src(b, selhi[1], nhi[1])
    if op == "mul":
        return np.stack([(lo0 * lo1).astype(F), (hi0 * hi1).astype(F)])
    return np.stack([(lo0 + lo1).astype(F), (hi0 + hi1).astype(F)])


def bfly_pk(H, L, t):
    """t = (t0, t1) as a pair.  5 packed ops."""
    S = pk("add", H, L)
    K = pk("add", H, L, nlo=(0, 1), nhi=(0, 1))                    # (k01, k00)
    M1 = pk("mul", K, t, sel=(0, 0), selhi=(1, 0))                 # (k01 t0, k00 t0)
    M2 = pk("mul", K, t, sel=(1, 1), selhi=(0, 1), nhi=(0, 1))     # (k00 t1, -k01 t1)
    return S, pk("add", M1, M2)


def last3_pk(Z, a2):
    """Z: list of 8 pairs (z[2j], z[2j+1]); a2 as pair (a2, a2).  imdct.rs:234-288 in 28 packed ops."""
    Z = list(Z)
    Z7n = pk("add", Z[7], Z[3]); Z3n = pk("add", Z[7], Z[3], nlo=(0, 1), nhi=(0, 1))
    Z[7], Z[3] = Z7n, Z3n
    Z6n = pk("add", Z[6], Z[2]); D = pk("add", Z[6], Z[2], nlo=(0, 1), nhi=(0, 1))      # (k11, k00)
    E = pk("add", D, D, sel=(0, 1), selhi=(1, 0), nlo=(0, 1))                            # (k11-k00, k00+k11)
    Z[6], Z[2] = Z6n, pk("mul", E, a2)
    Z5n = pk("add", Z[5], Z[1])
    Z1n = pk("add", Z[1], Z[5], sel=(1, 1), selhi=(0, 0), nlo=(0, 1), nhi=(1, 0))        # (z3-z11, z10-z2)
    Z[5], Z[1] = Z5n, Z1n
    Z4n = pk("add", Z[4], Z[0])
    Kq = pk("add", Z[0], Z[4], sel=(0, 0), selhi=(1, 1), nlo=(1, 0), nhi=(0, 1))         # (k11, k00)
    E = pk("add", Kq, Kq, sel=(1, 0), selhi=(1, 0), nlo=(0, 1))                          # (k00-k11, k00+k11)
    Z[4], Z[0] = Z4n, pk("mul", E, a2)
    for b in (4, 0):
        W0, W1, W2, W3 = Z[b], Z[b + 1], Z[b + 2], Z[b + 3]
        A = pk("add", W3, W1)                                        # (y1, y0)
        Bm = pk("add", W3, W1, nlo=(0, 1), nhi=(0, 1))               # (k11, k00)
        Cc = pk("add", W2, W0)                                       # (y3, y2)
        Dm = pk("add", W2, W0, nlo=(0, 1), nhi=(0, 1))               # (k33, k22)
        Z[b + 3] = pk("add", A, Cc)
        Z[b + 2] = pk("add", A, Cc, nlo=(0, 1), nhi=(0, 1))
        Z[b + 1] = pk("add", Bm, Dm, sel=(0, 1), selhi=(1, 0), nlo=(0, 1))   # (k11-k22, k00+k33)
        Z[b] = pk("add", Bm, Dm, sel=(0, 1), selhi=(1, 0), nhi=(0, 1))       # (k11+k22, k00-k33)
    return Z


def step7_block_pk(P, Q, C2):
    """P = (D1, D0), Q = (E3, E2), C2 = (C0, C1) -> (D1', D0'), (E3', E2')  (imdct.rs:548-560). 7 packed ops."""
    Aa = pk("add", P, Q, nhi=(0, 1))                                  # (a11, a02)
    Bb = pk("add", P, Q, nlo=(0, 1))                                  # (b3, b2)
    M1 = pk("mul", Aa, C2, sel=(0, 1), selhi=(1, 1))                  # (C1 a11, C1 a02)
    M2 = pk("mul", Aa, C2, sel=(1, 0), selhi=(0, 0), nlo=(0, 1))      # (-C0 a02, C0 a11)
    Bv = pk("add", M1, M2)                                            # (b1, b0)
    Dn = pk("add", Bb, Bv)                                            # (b3+b1, b2+b0)
    En = pk("add", Bv, Bb, nlo=(0, 1), nhi=(1, 0))                    # (b1-b3, b2-b0)
    return Dn, En


def step8_pk(Wv, Bq):
    """Wv = (w1, w0), Bq = (Bc, Bs) -> (pa, pb)  (imdct.rs:619-620). 3 packed ops."""
    N1 = pk("mul", Wv, Bq, sel=(1, 1), selhi=(1, 0), nhi=(1, 0))      # (w0 Bs, -w0 Bc)
    N2 = pk("mul", Wv, Bq, sel=(0, 0), selhi=(0, 1), nlo=(0, 1), nhi=(0, 1))  # (-w1 Bc, -w1 Bs)
    return pk("add", N1, N2)


def ola_pk(R, PP, ppsel, S2):
    """R = (pa, pb), PP pair holding pp at index ppsel, S2 = (s[q], s[1023-q]) -> (out[q], out[1023-q]). 3 packed ops."""
    O1 = pk("mul", R, S2, sel=(0, 0), selhi=(0, 1), nhi=(1, 0))       # (pa sq, -pa sr)
    O2 = pk("mul", PP, S2, sel=(ppsel, 1), selhi=(ppsel, 0))          # (pp sr, pp sq)
    return pk("add", O1, O2)


def imdct_wave_pk(X, img, prev_pb=None, window=None):
    """Packed-op version of imdct_wave.  Returns the n-sample block; if prev_pb (512 values) and the image window are
    given also returns the 1024 overlap-added output samples (audio.rs:1116-1118)."""
    X = np.asarray(X, F)
    lam = LANES
    P = [None] * 8
    up = [None] * 4
    for x in range(4):
        m = 64 * x + lam
        Xa = np.stack([X[4 * m], X[4 * m + 1]])
        Xb = np.stack([X[4 * m + 2], X[4 * m + 3]])
        au = img.apair[m].T.copy()
        al = img.apair[511 - m].T.copy()
        T1 = pk("mul", Xa, au, sel=(0, 1), selhi=(0, 0))                       # (X0 a1, X0 a0)
        T2 = pk("mul", Xb, au, sel=(0, 0), selhi=(0, 1), nhi=(0, 1))           # (X2 a0, -X2 a1)
        up[x] = pk("add", T1, T2)
        T3 = pk("mul", Xb, al, sel=(1, 1), selhi=(1, 0), nlo=(1, 0), nhi=(1, 0))  # (-X3 b1, -X3 b0)
        T4 = pk("mul", Xa, al, sel=(1, 0), selhi=(1, 1), nlo=(1, 0))           # (-X1 b0, X1 b1)
        P[x] = pk("add", T3, T4)
    for xs in range(4):
        P[7 - xs] = up[xs][:, 63 - lam]
    for x in range(4):
        P[x + 4], P[x] = bfly_pk(P[x + 4], P[x], img.tw_s2[x].T)
    for x in (2, 3, 6, 7):
        P[x], P[x - 2] = bfly_pk(P[x], P[x - 2], img.tw_l0[x & 1].T)
    for x in (1, 3, 5, 7):
        P[x], P[x - 1] = bfly_pk(P[x], P[x - 1], img.tw_l1.T)
    lds = np.zeros((576, 2), F)
    for x in range(8):
        lds[slot_t2(64 * x + lam)] = P[x].T
    Xc, lo3 = lam >> 3, lam & 7
    Q = [lds[slot_t2(64 * Xc + 8 * y + lo3)].T.copy() for y in range(8)]
    for y in (4, 5, 6, 7):
        Q[y], Q[y - 4] = bfly_pk(Q[y], Q[y - 4], img.tw_l2[y & 3][lo3].T)
    for y in (2, 3, 6, 7):
        Q[y], Q[y - 2] = bfly_pk(Q[y], Q[y - 2], img.tw_l3[y & 1][lo3].T)
    for y in (1, 3, 5, 7):
        Q[y], Q[y - 1] = bfly_pk(Q[y], Q[y - 1], img.tw_l4[lo3].T)
    for y in range(8):
        lds[slot_t3(64 * Xc + 8 * y + lo3)] = Q[y].T
    Z = [lds[slot_t3(8 * lam + zz)].T.copy() for zz in range(8)]
    a2 = np.stack([np.full(64, img.a2, F)] * 2)
    Z = last3_pk(Z, a2)
    for zz in range(8):
        lds[slot_t4(8 * lam + zz)] = Z[zz].T
    pa, pb, outq = {}, {}, {}
    for c in range(2):
        mp = 2 * lam + c
        q2 = 2 * rev_bits(mp, 7)
        pq, pq256 = lds[slot_t4(q2)].T.copy(), lds[slot_t4(q2 + 256)].T.copy()
        p255, p511 = lds[slot_t4(255 - q2)].T.copy(), lds[slot_t4(511 - q2)].T.copy()
        C = img.c4[c]
        Dn1, En1 = step7_block_pk(p511, pq, np.stack([C[:, 0], C[:, 1]]))      # (D1',D0'), (E3',E2')
        Dn2, En2 = step7_block_pk(p255, pq256, np.stack([C[:, 2], C[:, 3]]))   # (D3',D2'), (E1',E0')
        Bl, Bh = img.b_lo[c], img.b_hi[c]
        R = [step8_pk(Dn1, np.stack([Bl[:, 0], Bl[:, 1]])), step8_pk(Dn2, np.stack([Bl[:, 2], Bl[:, 3]])),
             step8_pk(En2, np.stack([Bh[:, 0], Bh[:, 1]])), step8_pk(En1, np.stack([Bh[:, 2], Bh[:, 3]]))]
        qs = [511 - 2 * mp, 510 - 2 * mp, 1 + 2 * mp, 2 * mp]
        for k in range(4):
            for l in range(64):
                pa[int(qs[k][l])] = R[k][0][l]
                pb[int(qs[k][l])] = R[k][1][l]
            if prev_pb is not None:
                PP = np.stack([np.asarray(prev_pb, F)[qs[k]], np.zeros(64, F)])
                S2 = np.stack([img.win[c][:, 2 * k], img.win[c][:, 2 * k + 1]])
                O = ola_pk(R[k], PP, 0, S2)
                for l in range(64):
                    outq[int(qs[k][l])] = O[0][l]
                    outq[1023 - int(qs[k][l])] = O[1][l]
    out = np.zeros(N, F)
    for q in range(N4):
        out[q] = pa[q]
        out[N2 - 1 - q] = -pa[q]
        out[N2 + q] = pb[q]
        out[N - 1 - q] = pb[q]
    if prev_pb is not None:
        return out, np.array([outq[i] for i in range(1024)], F)
    return out
