"""Thread-level numpy model of the big-block kernel k_big<BS> (lw_kernels_big.hip), n = 2^BS = 4096 / 8192.

One workgroup of T = n / 32 threads transforms one (block, channel) at a time; every per-thread register is a numpy array of
shape [T].  The n/4 complex pairs of the transform (imdct.rs:291-659; pair q = floats (2q, 2q + 1) of the reference's work
array) live in LDS between the passes, addressed by q' = n/4 - 1 - q and padded by one pair per eight (pad()); a thread holds
eight pairs per pass and runs up to three butterfly stages on them in registers:
  P0  step 1 (imdct.rs:337-371) of j = t + T i (i < 4): the pairs q' = j and n/8 + j, then the stage of distance n/8 (step 2,
      :385-430) and, n = 8192, the stage of distance 512
  P1  distances 256, 128, 64      q' = 512 (t / 64) + t % 64 + 64 i
  P2  distances 32, 16, 8         q' = 64 (t / 8) + t % 8 + 8 i
  P3  the fused last three stages (:234-288) on q' = 8 t + i
  E   m = t + T e (e < 2): bit-reverse gather (:490-528), step 7 (:533-580), step 8 (:589-658) -> (pa, pb) at
      p = 2m, 2m + 1, n/4 - 2 - 2m, n/4 - 1 - 2m; window / overlap-add (audio.rs:1116-1118) of the samples n/4 - 1 - p and
      n/4 + p with the predecessor's pb(p), which the same thread holds
Every stage of distance D is the reference's butterfly (hi = q', lo = q' + D, twiddle A[r n / (2 D)], r = q' mod 2D < D): step 2
is the stage of distance n/8.  All arithmetic is single f32 operations in the reference's order (no contraction); the model
must reproduce the oracle bit for bit (tests/test_big_model.py)."""
import numpy as np

F = np.float32


def pad(q):
    return q + (q >> 3)


def bfly(V, hi, lo, t0, t1):
    """imdct.rs:445-477 on pairs (x = even float, y = odd float): hi <- hi + lo, lo <- (hi - lo) * twiddle"""
    k00 = V[hi][1] - V[lo][1]
    k01 = V[hi][0] - V[lo][0]
    h = (V[hi][0] + V[lo][0], V[hi][1] + V[lo][1])
    l = (k01 * t0 + k00 * t1, k00 * t0 - k01 * t1)
    V[hi], V[lo] = h, l


def radix8(V, q0, A, n, D, stages=3):
    """V: list of 8 (x, y) register pairs of the pairs q' = q0 + (D / 4) i ... for three stages D, D/2, D/4 (or the first `stages`)"""
    step = D // 4
    off = q0 % step if stages == 3 else None
    for s in range(stages):
        d = D >> s                      # distance of this stage
        k1 = n // (2 * d)
        span = d // step                # register distance
        for i in range(8):
            if (i // span) % 2 == 0:
                r = (q0 + step * i) % (2 * d)
                assert np.all(r < d)
                bfly(V, i, i + span, A[r * k1], A[r * k1 + 1])


def iter54(w):
    k00 = w[7] - w[3]
    y0 = w[7] + w[3]
    y2 = w[5] + w[1]
    k22 = w[5] - w[1]
    k33 = w[4] - w[0]
    k11 = w[6] - w[2]
    y1 = w[6] + w[2]
    y3 = w[4] + w[0]
    return [k11 + k22, k00 - k33, k11 - k22, k00 + k33, y1 - y3, y0 - y2, y1 + y3, y0 + y2]


def last3(z, a2):
    """imdct.rs:234-288 on z[0..16) (z[15 - k] is the reference's z![-k])"""
    z = list(z)
    k00 = z[15] - z[7]
    k11 = z[14] - z[6]
    z[15] = z[15] + z[7]
    z[14] = z[14] + z[6]
    z[7] = k00
    z[6] = k11
    k00 = z[13] - z[5]
    k11 = z[12] - z[4]
    z[13] = z[13] + z[5]
    z[12] = z[12] + z[4]
    z[5] = (k00 + k11) * a2
    z[4] = (k11 - k00) * a2
    k00 = z[3] - z[11]
    k11 = z[10] - z[2]
    z[11] = z[11] + z[3]
    z[10] = z[10] + z[2]
    z[3] = k11
    z[2] = k00
    k00 = z[1] - z[9]
    k11 = z[8] - z[0]
    z[9] = z[9] + z[1]
    z[8] = z[8] + z[0]
    z[1] = (k00 + k11) * a2
    z[0] = (k00 - k11) * a2
    z[8:16] = iter54(z[8:16])
    z[0:8] = iter54(z[0:8])
    return z


def block(spec, bs, tabs, prev_pb=None):
    """spec: n/2 f32 (floor x residue of one channel).  Returns (time-domain block n f32, overlap-added samples n/2 f32 or None,
    pb[n/4] as the threads hold it, indexed by p)."""
    A, B, Cc, W, br = tabs
    n = 1 << bs
    n2, n4, n8, T = n // 2, n // 4, n // 8, n // 32
    t = np.arange(T)
    U = np.asarray(spec, F)
    lds = np.zeros((pad(n4 - 1) + 1, 2), F)

    def put(q, v):
        lds[pad(q), 0], lds[pad(q), 1] = v[0], v[1]

    def get(q):
        return (lds[pad(q), 0].copy(), lds[pad(q), 1].copy())

    # ---- P0
    V = [None] * 8
    for i in range(4):
        j = t + T * i
        x0, x2 = U[4 * j], U[4 * j + 2]
        V[i] = (x0 * A[2 * j + 1] + x2 * A[2 * j], x0 * A[2 * j] - x2 * A[2 * j + 1])
        e, a = n2 - 3 - 4 * j, n4 + 2 * j
        me2, me0 = -U[e + 2], -U[e]
        V[4 + i] = (me2 * A[a + 1] + me0 * A[a], me2 * A[a] - me0 * A[a + 1])
    for i in range(4):                      # distance n/8 (step 2): r = j, k1 = 4
        j = t + T * i
        bfly(V, i, 4 + i, A[4 * j], A[4 * j + 1])
    if bs == 13:                            # distance 512: (j, j + 512) in both halves, r = j (< 512), k1 = 8
        for i in (0, 1, 4, 5):
            j = t + T * (i & 1)
            bfly(V, i, i + 2, A[8 * j], A[8 * j + 1])
    for i in range(4):
        put(t + T * i, V[i])
        put(n8 + t + T * i, V[4 + i])
    # ---- P1, P2
    for D in (256, 32):
        step = D // 4
        q0 = (t // step) * (2 * D) + t % step
        V = [get(q0 + step * i) for i in range(8)]
        radix8(V, q0, A, n, D)
        for i in range(8):
            put(q0 + step * i, V[i])
    # ---- P3: z[2k], z[2k+1] = pair q' = 8t + 7 - k
    z = [None] * 16
    for k in range(8):
        z[2 * k], z[2 * k + 1] = get(8 * t + 7 - k)
    z = last3(z, A[n8])
    for k in range(8):
        put(8 * t + 7 - k, (z[2 * k], z[2 * k + 1]))
    # ---- E
    td = np.zeros(n, F)
    ola = None if prev_pb is None else np.zeros(n2, F)
    pb_out = np.zeros(n4, F)

    def pair_of_float(k):                   # the pair holding floats (k, k + 1) of the reference's array, k even
        return get(n4 - 1 - k // 2)
    for e_ in range(2):
        m = t + T * e_
        t0 = n // 16 - 1 - m
        k1, k1p, k0, k0p = br[2 * m], br[2 * m + 1], br[2 * t0], br[2 * t0 + 1]
        Pa, Pb, Pc, Pd = pair_of_float(k1), pair_of_float(k1p), pair_of_float(k0 + 2), pair_of_float(k0p + 2)
        d = 4 * m
        ve = [Pb[1], Pb[0], Pa[1], Pa[0]]   # v[e .. e+3], e = n/2 - 4 - 4m
        vd = [Pd[1], Pd[0], Pc[1], Pc[0]]   # v[d .. d+3]
        C0, C1, C2, C3 = Cc[d], Cc[d + 1], Cc[d + 2], Cc[d + 3]
        a02 = vd[0] - ve[2]
        a11 = vd[1] + ve[3]
        b0 = C1 * a02 + C0 * a11
        b1 = C1 * a11 - C0 * a02
        b2 = vd[0] + ve[2]
        b3 = vd[1] - ve[3]
        vd[0], vd[1], ve[2], ve[3] = b2 + b0, b3 + b1, b2 - b0, b1 - b3
        a02 = vd[2] - ve[0]
        a11 = vd[3] + ve[1]
        b0 = C3 * a02 + C2 * a11
        b1 = C3 * a11 - C2 * a02
        b2 = vd[2] + ve[0]
        b3 = vd[3] - ve[1]
        vd[2], vd[3], ve[0], ve[1] = b2 + b0, b3 + b1, b2 - b0, b1 - b3
        for p, w0, w1 in ((2 * m, vd[0], vd[1]), (2 * m + 1, vd[2], vd[3]), (n4 - 2 - 2 * m, ve[0], ve[1]), (n4 - 1 - 2 * m, ve[2], ve[3])):
            pa = w0 * B[2 * p + 1] - w1 * B[2 * p]
            pb = (-w0) * B[2 * p] - w1 * B[2 * p + 1]
            q = n4 - 1 - p
            td[q], td[n2 - 1 - q], td[n2 + q], td[n - 1 - q] = pa, -pa, pb, pb
            pb_out[p] = pb
            if prev_pb is not None:
                pp = prev_pb[p]
                ola[q] = (pa * W[q]) + (pp * W[n2 - 1 - q])
                ola[n2 - 1 - q] = ((-pa) * W[n2 - 1 - q]) + (pp * W[q])
    return td, ola, pb_out


def floor_entries(xs, ys, active):
    """k_big's floor segment table (floor_entry in lw_kernels_big.hip): one entry {dy, c0, 1/adx, w} per STATIC interval (behind post
    s of the floor configuration, ascending x), describing the ACTIVE segment that covers it."""
    Fp = len(xs)
    ent = np.zeros((Fp, 4), F)
    for s in range(Fp):
        lo = max(i for i in range(s + 1) if active[i])
        above = [i for i in range(s + 1, Fp) if active[i]]
        hi = above[0] if above else lo
        xlo, xhi, ylo, yhi = F(xs[lo]), F(xs[hi]), int(ys[lo]), int(ys[hi])
        dy = F(yhi - ylo)
        adx = xhi - xlo if above else F(1.0)
        down = yhi < ylo
        ent[s, 0] = dy
        ent[s, 1] = (F(0.875) * adx - F(0.5)) - xhi * dy if down else (F(0.5) - F(0.125) * adx) - xlo * dy
        ent[s, 2] = F(1.0) / adx if above else F(1.0)
        ent[s, 3] = F(yhi if down else ylo) + F(2097153.0)
    return ent


def floor_bins(xs, ent, n2):
    """y of every bin: the static interval of the bin (largest s with xs[s] <= k), its entry, two fused multiply-adds and a mask
    (floor_bin).  The f64 evaluation rounds far below the quarter the f32 result is rounded to (see tests/test_big_model.py)."""
    k = np.arange(n2)
    sid = np.searchsorted(np.asarray(xs), k, side="right") - 1
    e = ent[sid].astype(np.float64)
    inner = (k * e[:, 0] + e[:, 1])
    assert np.array_equal(inner.astype(np.float32).astype(np.float64), inner)
    t = (inner * e[:, 2] + e[:, 3]).astype(np.float32)
    return ((t.view(np.uint32) & 0x7FC) >> 2).astype(np.int64) - 1


def block_wave(spec, bs=12):
    """NOT built yet -- the index maps of the next design: ONE wave per channel of a 4096-point block, 16 pairs per
    lane, no barrier.  Lane l holds j = l + 64 i (i < 8): step 1 gives it the pairs q' = j and 512 + j, and the stages of distance
    512, 256, 128 and 64 all pair registers of the SAME lane (i with the mirrored set, i with i + 4, i + 2, i + 1): four stages
    before the first LDS round trip.  Then two units of the workgroup kernel's P2 / P3 / E per lane (virtual threads l and l + 64
    of block() with T = 128): three LDS round trips per channel instead of five, ordered by the wave's own LDS ordering.
    Returns the time-domain block (compared with the oracle in tests/test_big_model.py)."""
    from common import po
    assert bs == 12
    A, B, Cc, W, br = po.tables(bs)
    n = 1 << bs
    n2, n4, n8 = n // 2, n // 4, n // 8
    l = np.arange(64)
    U = np.asarray(spec, F)
    lds = np.zeros((pad(n4 - 1) + 1, 2), F)

    def put(q, v):
        lds[pad(q), 0], lds[pad(q), 1] = v[0], v[1]

    def get(q):
        return (lds[pad(q), 0].copy(), lds[pad(q), 1].copy())

    # ---- P0: step 1 + distances 512, 256, 128, 64 in registers; H[i] = pair j_i, L[i] = pair 512 + j_i
    H, L = [None] * 8, [None] * 8
    for i in range(8):
        j = l + 64 * i
        x0, x2 = U[4 * j], U[4 * j + 2]
        H[i] = (x0 * A[2 * j + 1] + x2 * A[2 * j], x0 * A[2 * j] - x2 * A[2 * j + 1])
        e, a = n2 - 3 - 4 * j, n4 + 2 * j
        me2, me0 = -U[e + 2], -U[e]
        L[i] = (me2 * A[a + 1] + me0 * A[a], me2 * A[a] - me0 * A[a + 1])
    R = H + L                                   # registers 0..7 = q' = j_i, 8..15 = 512 + j_i
    for i in range(8):                          # distance 512 (step 2): r = j
        j = l + 64 * i
        bfly(R, i, 8 + i, A[4 * j], A[4 * j + 1])
    for D, span in ((256, 4), (128, 2), (64, 1)):
        k1 = n // (2 * D)
        for half in (0, 8):
            for i in range(8):
                if (i // span) % 2 == 0:
                    r = (l + 64 * i) % (2 * D)
                    assert np.all(r < D)
                    bfly(R, half + i, half + i + span, A[r * k1], A[r * k1 + 1])
    for i in range(8):
        put(l + 64 * i, R[i])
        put(n8 + l + 64 * i, R[8 + i])
    # ---- the rest: block()'s P2, P3, E with two virtual threads per lane
    T = 128
    t = np.arange(T)
    step = 8
    q0 = (t // step) * 64 + t % step
    V = [get(q0 + step * i) for i in range(8)]
    radix8(V, q0, A, n, 32)
    for i in range(8):
        put(q0 + step * i, V[i])
    z = [None] * 16
    for k in range(8):
        z[2 * k], z[2 * k + 1] = get(8 * t + 7 - k)
    z = last3(z, A[n8])
    for k in range(8):
        put(8 * t + 7 - k, (z[2 * k], z[2 * k + 1]))
    td = np.zeros(n, F)
    for e_ in range(2):
        m = t + T * e_
        t0 = n // 16 - 1 - m
        k1, k1p, k0, k0p = br[2 * m], br[2 * m + 1], br[2 * t0], br[2 * t0 + 1]
        Pa, Pb = get(n4 - 1 - k1 // 2), get(n4 - 1 - k1p // 2)
        Pc, Pd = get(n4 - 2 - k0 // 2), get(n4 - 2 - k0p // 2)
        d = 4 * m
        ve = [Pb[1], Pb[0], Pa[1], Pa[0]]
        vd = [Pd[1], Pd[0], Pc[1], Pc[0]]
        C0, C1, C2, C3 = Cc[d], Cc[d + 1], Cc[d + 2], Cc[d + 3]
        a02, a11 = vd[0] - ve[2], vd[1] + ve[3]
        b0, b1, b2, b3 = C1 * a02 + C0 * a11, C1 * a11 - C0 * a02, vd[0] + ve[2], vd[1] - ve[3]
        vd[0], vd[1], ve[2], ve[3] = b2 + b0, b3 + b1, b2 - b0, b1 - b3
        a02, a11 = vd[2] - ve[0], vd[3] + ve[1]
        b0, b1, b2, b3 = C3 * a02 + C2 * a11, C3 * a11 - C2 * a02, vd[2] + ve[0], vd[3] - ve[1]
        vd[2], vd[3], ve[0], ve[1] = b2 + b0, b3 + b1, b2 - b0, b1 - b3
        for p, w0, w1 in ((2 * m, vd[0], vd[1]), (2 * m + 1, vd[2], vd[3]), (n4 - 2 - 2 * m, ve[0], ve[1]), (n4 - 1 - 2 * m, ve[2], ve[3])):
            pa = w0 * B[2 * p + 1] - w1 * B[2 * p]
            pb = (-w0) * B[2 * p] - w1 * B[2 * p + 1]
            q = n4 - 1 - p
            td[q], td[n2 - 1 - q], td[n2 + q], td[n - 1 - q] = pa, -pa, pb, pb
    return td
