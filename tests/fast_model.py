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
    """Per-bin floor through the segment table the kernel builds (trunc((t*dy +- 0.5) * (1/adx)) form)."""
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
            x0, y0, dy, adx = xs[lo], y[lo], y[hi] - y[lo], xs[hi] - xs[lo]
        else:
            x0, y0, dy, adx = xs[lo], y[lo], 0, 1
        sel = k[sid == s]
        if len(sel) == 0:
            continue
        rinv = F(1.0) / F(adx)
        tf = (sel - x0).astype(F)
        zf = (tf * F(dy) + F(np.copysign(0.5, dy if dy != 0 else 1.0))).astype(F)  # exact: |t*dy| < 2^18
        q = np.trunc((zf * rinv).astype(F)).astype(np.int64)
        out[sel] = img_inv_db[y0 + q]
    return out
