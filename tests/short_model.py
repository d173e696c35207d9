"""Lane-level numpy model of the block kernel k_short<L> (lw_kernels_long.hip), n = 32 L = 256 / 512 / 1024.

One wave = 64 lanes = 64 / L blocks ("slots") x L lanes; every per-lane register is a numpy array of shape [64].  The model
follows the kernel's data movement (layouts B' / C' / D' / E' of lw_fast.hpp, the mirror exchange of step 1, the register <->
lane exchanges, the LDS bit-reverse gather) and its exact packed f32 operations (fast_model.pk), consumes the
product's own LDS image (lw_debug_short_image) and must reproduce the oracle bit for bit (tests/test_short_model.py)."""
import numpy as np

from fast_model import F, bfly_pk, last3_pk, ola_pk, pk, rev_bits, step7_block_pk, step8_pk
LANES = np.arange(64)


def layout(L):
    """byte offsets of the image sections (LwBlkLayout<L> in lw_fast.hpp)"""
    o = {}
    o["apair"] = 0
    o["tw_s2"] = o["apair"] + 64 * L
    o["tw_l0"] = o["tw_s2"] + 32 * L
    o["tw_l1"] = o["tw_l0"] + 16 * L
    o["tw_c2"] = o["tw_l1"] + 8 * L
    o["tw_c3"] = o["tw_c2"] + 128
    o["a2"] = o["tw_c3"] + 64
    o["c4"] = o["a2"] + 16
    o["b_lo"] = o["c4"] + 32 * L
    o["b_hi"] = o["b_lo"] + 32 * L
    o["win"] = o["b_hi"] + 32 * L
    o["inv_db"] = o["win"] + 64 * L
    o["xsf"] = o["inv_db"] + 1024
    o["sid16"] = o["xsf"] + 512
    o["end"] = o["sid16"] + 64 * L
    o["total"] = (o["end"] + 1023) & ~1023
    return o


class Image:
    def __init__(self, blob, L=8):
        self.L = L
        f = np.frombuffer(blob, np.float32)
        o = {k: v // 4 for k, v in layout(L).items()}
        self.apair = f[o["apair"]: o["apair"] + 16 * L].reshape(8 * L, 2)
        self.tw_s2 = f[o["tw_s2"]: o["tw_s2"] + 8 * L].reshape(4, L, 2)
        self.tw_l0 = f[o["tw_l0"]: o["tw_l0"] + 4 * L].reshape(2, L, 2)
        self.tw_l1 = f[o["tw_l1"]: o["tw_l1"] + 2 * L].reshape(L, 2)
        self.tw_c2 = f[o["tw_c2"]: o["tw_c2"] + 32].reshape(2, 8, 2)
        self.tw_c3 = f[o["tw_c3"]: o["tw_c3"] + 16].reshape(8, 2)
        self.a2 = f[o["a2"]]
        self.c4 = f[o["c4"]: o["c4"] + 8 * L].reshape(2, L, 4)
        self.b_lo = f[o["b_lo"]: o["b_lo"] + 8 * L].reshape(2, L, 4)
        self.b_hi = f[o["b_hi"]: o["b_hi"] + 8 * L].reshape(2, L, 4)
        self.win = f[o["win"]: o["win"] + 16 * L].reshape(2, L, 8)
        self.inv_db = f[o["inv_db"]: o["inv_db"] + 256]
        self.xsf = f[o["xsf"]: o["xsf"] + 128].reshape(2, 64)
        so = layout(L)["sid16"] // 2
        self.sid16 = np.frombuffer(blob, np.uint16)[so: so + 32 * L].reshape(2, 4, L, 4)


def xchg(Q, regbit, lanebit):
    """exchange of a register-index bit with a lane bit for the eight pairs of every lane (permlane / DPP swaps)"""
    out = [None] * 8
    lb = (LANES >> lanebit) & 1
    for x in range(8):
        rb = (x >> regbit) & 1
        src = np.where(lb == rb, LANES, LANES ^ (1 << lanebit))
        other = Q[x ^ (1 << regbit)]
        out[x] = np.where((lb == rb)[None, :], Q[x], other[:, src])
    return out


def blk_slot(L, g, p):
    """8-byte slot of pair p of block g in a wave's bit-reverse gather area (blk_slot<L> in lw_kernels_long.hip): a bijection of
    (g, p) onto 0 .. 511, linear over GF(2), free of bank conflicts for the kernel's writes and reads (gather_bank_cycles)"""
    pb = lambda k: (p >> k) & 1
    gb = lambda k: (g >> k) & 1
    if L == 8:
        return pb(3) | gb(0) << 1 | (pb(4) ^ pb(1)) << 2 | (pb(5) ^ pb(2)) << 3 | gb(1) << 4 | pb(1) << 5 | pb(2) << 6 | pb(0) << 7 | gb(2) << 8
    if L == 16:
        return pb(3) | pb(4) << 1 | (pb(5) ^ pb(1)) << 2 | (pb(6) ^ pb(2)) << 3 | gb(0) << 4 | pb(1) << 5 | pb(2) << 6 | pb(0) << 7 | gb(1) << 8
    return pb(3) | pb(4) << 1 | pb(5) << 2 | (pb(1) ^ pb(7)) << 3 | pb(2) << 4 | pb(0) << 5 | pb(6) << 6 | pb(7) << 7 | gb(0) << 8


def lane_hi(L, l):
    """pair bits above the register's three held by lane l of a block in layout D'"""
    b = lambda k: (l >> k) & 1
    if L == 8:
        return l
    if L == 16:
        return b(2) << 3 | b(1) << 2 | b(3) << 1 | b(0)
    return b(2) << 4 | b(4) << 3 | b(3) << 2 | b(1) << 1 | b(0)


def gather_bank_cycles(L, slot=None):
    """LDS-array cycles of the gather's 8 ds_write_b64 and 8 ds_read_b64 of one wave under MI355X's banking rules
    (MI355X_MICROARCH.md: ds_write_b64 is served in four groups of 16 contiguous lanes on 32 banks of 4 bytes, ds_read_b64 in two
    groups of 32 lanes on 64 banks; a group costs as many cycles as its busiest bank has distinct dwords).  Conflict-free: (32, 16)."""
    slot = slot or (lambda g, p: blk_slot(L, g, p))
    P, vb = 8 * L, {8: 4, 16: 5, 32: 6}[L]

    def cycles(addrs, group, banks):
        tot = 0
        for g0 in range(0, 64, group):
            busy = {}
            for lane in range(g0, g0 + group):
                for d in range(2):
                    a = addrs[lane] + 4 * d
                    busy.setdefault((a // 4) % banks, set()).add(a // 4)
            tot += max(len(v) for v in busy.values())
        return tot

    w = sum(cycles([8 * slot(lane // L, 8 * lane_hi(L, lane % L) + zz) for lane in range(64)], 16, 32) for zz in range(8))
    r = 0
    for c in range(2):
        for kind in range(4):
            addrs = []
            for lane in range(64):
                q = 2 * int(rev_bits(np.array([2 * (lane % L) + c]), vb)[0])
                addrs.append(8 * slot(lane // L, [q, q + P // 2, P // 2 - 1 - q, P - 1 - q][kind]))
            r += cycles(addrs, 32, 64)
    return w, r


def imdct_wave(X, img, prev_pb=None):
    """X: [64 / L][16 L] spectra (one channel of the wave's blocks).  Returns the [64 / L][32 L] time-domain blocks; with prev_pb
    ([64 / L][8 L]: right part pb(q) of each block's predecessor) also the overlap-added samples (audio.rs:1116-1118) and pb."""
    L = img.L
    P = 8 * L
    N = 4 * P
    N2, N4 = N // 2, N // 4
    S = 64 // L
    G, Ln = LANES // L, LANES % L
    X = np.asarray(X, F)
    # ---- load layout: lane (g, l) holds float4 groups m = L x + l, x = 0..3, of block g
    Pq = [None] * 8   # layout B': Pq[x] = pair p = L x + l
    up = [None] * 4
    for x in range(4):
        m = L * x + Ln
        Xa = np.stack([X[G, 4 * m], X[G, 4 * m + 1]])
        Xb = np.stack([X[G, 4 * m + 2], X[G, 4 * m + 3]])
        au = img.apair[m].T.copy()
        al = img.apair[P - 1 - m].T.copy()
        T1 = pk("mul", Xa, au, sel=(0, 1), selhi=(0, 0))
        T2 = pk("mul", Xb, au, sel=(0, 0), selhi=(0, 1), nhi=(0, 1))
        up[x] = pk("add", T1, T2)                                           # pair P-1-m, still on the mirror lane
        T3 = pk("mul", Xb, al, sel=(1, 1), selhi=(1, 0), nlo=(1, 0), nhi=(1, 0))
        T4 = pk("mul", Xa, al, sel=(1, 0), selhi=(1, 1), nlo=(1, 0))
        Pq[x] = pk("add", T3, T4)                                           # pair m
    mirror = L * G + (L - 1 - Ln)                                           # DPP row_half_mirror / row_mirror / ds_bpermute
    for xs in range(4):
        Pq[7 - xs] = up[xs][:, mirror]
    # ---- step 2 and stages l = 0, 1: register-local (the three highest pair bits)
    for x in range(4):
        Pq[x + 4], Pq[x] = bfly_pk(Pq[x + 4], Pq[x], img.tw_s2[x][Ln].T)
    for x in (2, 3, 6, 7):
        Pq[x], Pq[x - 2] = bfly_pk(Pq[x], Pq[x - 2], img.tw_l0[x & 1][Ln].T)
    for x in (1, 3, 5, 7):
        Pq[x], Pq[x - 1] = bfly_pk(Pq[x], Pq[x - 1], img.tw_l1[Ln].T)
    lo3 = Ln & 7
    if L == 32:       # register bits 1, 0 <-> lane bits 4, 3; stages l = 2, 3
        Pq = xchg(xchg(Pq, 1, 4), 0, 3)
        for x in (2, 3, 6, 7):
            Pq[x], Pq[x - 2] = bfly_pk(Pq[x], Pq[x - 2], img.tw_c2[x & 1][lo3].T)
        for x in (1, 3, 5, 7):
            Pq[x], Pq[x - 1] = bfly_pk(Pq[x], Pq[x - 1], img.tw_c3[lo3].T)
    elif L == 16:     # register bit 0 <-> lane bit 3; stage l = 2
        Pq = xchg(Pq, 0, 3)
        for x in (1, 3, 5, 7):
            Pq[x], Pq[x - 1] = bfly_pk(Pq[x], Pq[x - 1], img.tw_c2[0][lo3].T)
    # ---- 8 x 8 transpose register index <-> lane bits 2..0 (t3_inreg): register z = p[2:0]
    Z = xchg(xchg(xchg(Pq, 2, 2), 1, 1), 0, 0)
    a2 = np.stack([np.full(64, img.a2, F)] * 2)
    Z = last3_pk(Z, a2)
    # pair bits above the register's three, from the lane bits
    b = lambda k: (Ln >> k) & 1
    if L == 8:
        hi = Ln
    elif L == 16:
        hi = b(2) << 3 | b(1) << 2 | b(3) << 1 | b(0)
    else:
        hi = b(2) << 4 | b(4) << 3 | b(3) << 2 | b(1) << 1 | b(0)
    # ---- bit-reverse gather through LDS (the wave's 512 pairs in blk_slot order; the reads below go through the same map)
    assert hi.tolist() == [lane_hi(L, int(l)) for l in Ln]
    slot_of = np.array([[blk_slot(L, g, p) for p in range(P)] for g in range(S)])
    assert sorted(slot_of.reshape(-1).tolist()) == list(range(S * P))
    lds_raw = np.zeros((S * P, 2), F)
    for zz in range(8):
        lds_raw[slot_of[G, 8 * hi + zz]] = Z[zz].T

    class _View:          # lds[P * g + p] = pair p of block g
        def __getitem__(self, idx):
            idx = np.asarray(idx)
            return lds_raw[slot_of[idx // P, idx % P]]
    lds = _View()
    pa = np.zeros((S, N4), F)
    pb = np.zeros((S, N4), F)
    out_ola = np.zeros((S, N2), F)
    vb = {8: 4, 16: 5, 32: 6}[L]
    for c in range(2):
        mp = 2 * Ln + c
        q2 = 2 * rev_bits(mp, vb)
        pq, pqh = lds[P * G + q2].T.copy(), lds[P * G + q2 + P // 2].T.copy()
        ph, pf = lds[P * G + P // 2 - 1 - q2].T.copy(), lds[P * G + P - 1 - q2].T.copy()
        C = img.c4[c][Ln]
        Dn1, En1 = step7_block_pk(pf, pq, np.stack([C[:, 0], C[:, 1]]))
        Dn2, En2 = step7_block_pk(ph, pqh, np.stack([C[:, 2], C[:, 3]]))
        Bl, Bh = img.b_lo[c][Ln], img.b_hi[c][Ln]
        R = [step8_pk(Dn1, np.stack([Bl[:, 0], Bl[:, 1]])), step8_pk(Dn2, np.stack([Bl[:, 2], Bl[:, 3]])),
             step8_pk(En2, np.stack([Bh[:, 0], Bh[:, 1]])), step8_pk(En1, np.stack([Bh[:, 2], Bh[:, 3]]))]
        qs = [P - 1 - 2 * mp, P - 2 - 2 * mp, 1 + 2 * mp, 2 * mp]
        for k in range(4):
            pa[G, qs[k]] = R[k][0]
            pb[G, qs[k]] = R[k][1]
            if prev_pb is not None:
                PP = np.stack([np.asarray(prev_pb, F)[G, qs[k]], np.zeros(64, F)])
                S2 = np.stack([img.win[c][Ln][:, 2 * k], img.win[c][Ln][:, 2 * k + 1]])
                O = ola_pk(R[k], PP, 0, S2)
                out_ola[G, qs[k]] = O[0]
                out_ola[G, N2 - 1 - qs[k]] = O[1]
    out = np.zeros((S, N), F)
    q = np.arange(N4)
    out[:, q] = pa
    out[:, N2 - 1 - q] = -pa
    out[:, N2 + q] = pb
    out[:, N - 1 - q] = pb
    if prev_pb is not None:
        return out, out_ola, pb
    return out


def floor_group_model(rec, xs, inv_db, sid16_slot, L=8):
    """Per-bin floor of one block-channel the way k_short builds it: active-post mask, one interval entry per post
    {dy, 0.5 sgn(dy) - x0 dy, 1/adx, y0}, then y(k) = y0 + trunc((k dy + c0) (1/adx)) with the static interval index of
    the image (sid16_slot: [4][L][4] u16 = 16 * interval of bin 4(L x + l) + j)."""
    Fp = len(xs)
    n2 = 16 * L
    if rec[0] == 0xFFFF:
        return np.zeros(n2, F)
    act = [(int(rec[i]) & 0x8000) != 0 for i in range(Fp)]
    y = [int(rec[i]) & 0xFF for i in range(Fp)]
    ent = []
    for s in range(Fp):
        lo = max(i for i in range(s + 1) if act[i])
        his = [i for i in range(s + 1, Fp) if act[i]]
        if his:
            hi = his[0]
            dy = F(y[hi] - y[lo])
            rinv = F(1.0) / F(xs[hi] - xs[lo])
        else:
            dy = F(0.0)
            rinv = F(1.0)
        c0 = F(np.copysign(F(0.5), dy)) - F(xs[lo]) * dy
        ent.append((dy, F(c0), rinv, y[lo]))
    out = np.zeros(n2, F)
    for x in range(4):
        for l in range(L):
            for j in range(4):
                k = 4 * (L * x + l) + j
                dy, c0, rinv, y0 = ent[int(sid16_slot[x][l][j]) // 16]
                z = F(F(k) * dy + c0)          # one fma in the kernel; exact either way (|k dy| < 2^18)
                q = int(np.trunc(F(z * rinv)))
                out[k] = inv_db[y0 + q]
    return out
