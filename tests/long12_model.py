"""Lane-level numpy model of k_long12 (lw_long12.inc): k_long's design for blocksize_1 = 12 (n = 4096), one wave per channel,
16 complex pairs per lane.

After step 1 (on the coalesced load layout, exchange with the mirror lane) and step 2 (pair bit 9, register-local) the 1024 pairs
of a channel are two independent 512-pair problems H0 (p9 = 0) and H1 (p9 = 1): each goes through k_long's register layouts
  B: lane = p[5:0], reg = p[8:6]                      stages l = 0, 1, 2   (pair bits 8, 7, 6)
  C: lane = (p[8:6], p[2:0]), reg = p[5:3]            stages l = 3, 4, 5   (pair bits 5, 4, 3)
  D: lane = p[8:3], reg = p[2:0]                      fused last three stages (imdct.rs:234-288)
with the register <-> lane exchanges of k_long (t2_inreg, t3_inreg), and they meet again in the bit-reverse gather through 8 KB of
LDS: pair p sits in slot (p >> 3 & 31) | (p & 7) << 5 | (p >> 8) << 8 (conflict-free on both sides, gather_bank_cycles).
Layout E: lane l finishes m' = 128 h + 2 l + c2 for k4 = 2 h + c2 = 0 .. 3: bit-reverse gather (imdct.rs:490-528), step 7, step 8,
window / overlap-add.  The model consumes the product's LDS image (lw_debug_short_image of a blocksize-12 stream: LwL12Layout in
lw_fast.hpp) and the A table, uses the kernel's packed f32 operations (fast_model.pk ...) and must reproduce the oracle bit for
bit (tests/test_long12_model.py)."""
import numpy as np

from fast_model import F, bfly_pk, last3_pk, ola_pk, pk, rev_bits, step7_block_pk, step8_pk
from short_model import xchg

LANES = np.arange(64)
N = 4096
N2, N4, N8 = N // 2, N // 4, N // 8
P = 1024   # complex pairs per channel

LAYOUT = {"tw_s2": 0, "tw_l0": 4096, "tw_l1": 6144, "tw_l2": 7168, "tw_l3": 7680, "tw_l4": 7936, "tw_l5": 8064, "a2": 8128,
          "c4": 8144, "b_lo": 12240, "b_hi": 16336, "win": 20432, "inv_db": 28624, "xsf": 29648, "end": 30160, "total": 30208}


class Image:
    def __init__(self, blob):
        assert len(blob) == LAYOUT["total"]
        f = np.frombuffer(blob, np.float32)
        o = {k: v // 4 for k, v in LAYOUT.items()}
        self.tw_s2 = f[o["tw_s2"]: o["tw_s2"] + 1024].reshape(8, 64, 2)
        self.tw_l0 = f[o["tw_l0"]: o["tw_l0"] + 512].reshape(4, 64, 2)
        self.tw_l1 = f[o["tw_l1"]: o["tw_l1"] + 256].reshape(2, 64, 2)
        self.tw_l2 = f[o["tw_l2"]: o["tw_l2"] + 128].reshape(64, 2)
        self.tw_l3 = f[o["tw_l3"]: o["tw_l3"] + 64].reshape(4, 8, 2)
        self.tw_l4 = f[o["tw_l4"]: o["tw_l4"] + 32].reshape(2, 8, 2)
        self.tw_l5 = f[o["tw_l5"]: o["tw_l5"] + 16].reshape(8, 2)
        self.a2 = f[o["a2"]]
        self.c4 = f[o["c4"]: o["c4"] + 1024].reshape(4, 64, 4)
        self.b_lo = f[o["b_lo"]: o["b_lo"] + 1024].reshape(4, 64, 4)
        self.b_hi = f[o["b_hi"]: o["b_hi"] + 1024].reshape(4, 64, 4)
        self.win = f[o["win"]: o["win"] + 2048].reshape(4, 64, 8)
        self.inv_db = f[o["inv_db"]: o["inv_db"] + 256]
        self.xsf = f[o["xsf"]: o["xsf"] + 128].reshape(2, 64)


def gather_slot(p):
    """8-byte slot of pair p in the wave's 8 KB gather area"""
    return ((p >> 3) & 31) | (p & 7) << 5 | (p >> 8) << 8


def half_transform(Q, img):
    """one 512-pair half (Q[x] = pair 64 x + lane of the half, layout B) -> layout D registers Z[zz] (pair 8 lane + zz of the half)"""
    lam = LANES
    Q = list(Q)
    for x in range(4):                                   # l = 0 (pair bit 8)
        Q[x + 4], Q[x] = bfly_pk(Q[x + 4], Q[x], img.tw_l0[x].T)
    for x in (2, 3, 6, 7):                               # l = 1 (bit 7)
        Q[x], Q[x - 2] = bfly_pk(Q[x], Q[x - 2], img.tw_l1[x & 1].T)
    for x in (1, 3, 5, 7):                               # l = 2 (bit 6)
        Q[x], Q[x - 1] = bfly_pk(Q[x], Q[x - 1], img.tw_l2.T)
    Q = xchg(xchg(xchg(Q, 2, 5), 1, 4), 0, 3)            # t2_inreg: register bits (2, 1, 0) <-> lane bits (5, 4, 3): layout C
    lo3 = lam & 7
    for y in (4, 5, 6, 7):                               # l = 3 (bit 5)
        Q[y], Q[y - 4] = bfly_pk(Q[y], Q[y - 4], img.tw_l3[y & 3][lo3].T)
    for y in (2, 3, 6, 7):                               # l = 4 (bit 4)
        Q[y], Q[y - 2] = bfly_pk(Q[y], Q[y - 2], img.tw_l4[y & 1][lo3].T)
    for y in (1, 3, 5, 7):                               # l = 5 (bit 3)
        Q[y], Q[y - 1] = bfly_pk(Q[y], Q[y - 1], img.tw_l5[lo3].T)
    Z = xchg(xchg(xchg(Q, 2, 2), 1, 1), 0, 0)            # t3_inreg: register bits <-> lane bits (2, 1, 0): layout D
    a2 = np.stack([np.full(64, img.a2, F)] * 2)
    return last3_pk(Z, a2)


def imdct_wave(X, img, A, prev_pb=None):
    """X: [2048] spectrum of one channel, A: the block size's A table (step 1 reads it from HBM).  Returns the 4096-sample block;
    with prev_pb ([1024]: pb(q) of the predecessor) also the 2048 overlap-added samples (audio.rs:1116-1118) and pb."""
    X = np.asarray(X, F)
    lam = LANES
    apair = np.asarray(A, F).reshape(P, 2)
    Q = [None] * 16          # Q[x] = pair 64 x + lane
    up = [None] * 8
    for x in range(8):       # step 1 (imdct.rs:337-371) on the load layout: float4 m = 64 x + lane
        m = 64 * x + lam
        Xa = np.stack([X[4 * m], X[4 * m + 1]])
        Xb = np.stack([X[4 * m + 2], X[4 * m + 3]])
        au = apair[m].T.copy()
        al = apair[P - 1 - m].T.copy()
        T1 = pk("mul", Xa, au, sel=(0, 1), selhi=(0, 0))
        T2 = pk("mul", Xb, au, sel=(0, 0), selhi=(0, 1), nhi=(0, 1))
        up[x] = pk("add", T1, T2)                                            # pair 1023 - m, still on the mirror lane
        T3 = pk("mul", Xb, al, sel=(1, 1), selhi=(1, 0), nlo=(1, 0), nhi=(1, 0))
        T4 = pk("mul", Xa, al, sel=(1, 0), selhi=(1, 1), nlo=(1, 0))
        Q[x] = pk("add", T3, T4)                                             # pair m
    for xs in range(8):
        Q[15 - xs] = up[xs][:, 63 - lam]
    for x in range(8):       # step 2 (imdct.rs:385-430): pairs p and p + 512
        Q[x + 8], Q[x] = bfly_pk(Q[x + 8], Q[x], img.tw_s2[x].T)
    Z = [half_transform(Q[0:8], img), half_transform(Q[8:16], img)]
    # layout D: lane = p[8:3] of the half, register zz = p[2:0]; the gather area holds all 1024 pairs
    lds = np.zeros((P, 2), F)
    for h in range(2):
        for zz in range(8):
            lds[gather_slot(512 * h + 8 * lam + zz)] = Z[h][zz].T
    pa, pb = np.zeros(N4, F), np.zeros(N4, F)
    out_ola = np.zeros(N2, F)
    for k4 in range(4):
        mp = 128 * (k4 >> 1) + 2 * lam + (k4 & 1)
        q2 = 2 * rev_bits(mp, 8)
        pq, pqh = lds[gather_slot(q2)].T.copy(), lds[gather_slot(q2 + 512)].T.copy()
        ph, pf = lds[gather_slot(511 - q2)].T.copy(), lds[gather_slot(1023 - q2)].T.copy()
        C = img.c4[k4]
        Dn1, En1 = step7_block_pk(pf, pq, np.stack([C[:, 0], C[:, 1]]))
        Dn2, En2 = step7_block_pk(ph, pqh, np.stack([C[:, 2], C[:, 3]]))
        Bl, Bh = img.b_lo[k4], img.b_hi[k4]
        R = [step8_pk(Dn1, np.stack([Bl[:, 0], Bl[:, 1]])), step8_pk(Dn2, np.stack([Bl[:, 2], Bl[:, 3]])),
             step8_pk(En2, np.stack([Bh[:, 0], Bh[:, 1]])), step8_pk(En1, np.stack([Bh[:, 2], Bh[:, 3]]))]
        qs = [P - 1 - 2 * mp, P - 2 - 2 * mp, 1 + 2 * mp, 2 * mp]
        for k in range(4):
            pa[qs[k]] = R[k][0]
            pb[qs[k]] = R[k][1]
            if prev_pb is not None:
                PP = np.stack([np.asarray(prev_pb, F)[qs[k]], np.zeros(64, F)])
                S2 = np.stack([img.win[k4][:, 2 * k], img.win[k4][:, 2 * k + 1]])
                O = ola_pk(R[k], PP, 0, S2)
                out_ola[qs[k]] = O[0]
                out_ola[N2 - 1 - qs[k]] = O[1]
    out = np.zeros(N, F)
    q = np.arange(N4)
    out[q] = pa
    out[N2 - 1 - q] = -pa
    out[N2 + q] = pb
    out[N - 1 - q] = pb
    if prev_pb is not None:
        return out, out_ola, pb
    return out


def edge_form(pa, pb, E, edge_l, edge_r):
    """k_long12<EDGE>'s stores of a block with a short slope (lw_long12.inc, finish phase), lane by lane: returns
    (samples {index relative to the window start: value}, raw left edge [E] | None, raw right edge [E] | None, short state [2 E] | None).
    A lane owns, per output half h, the raw quads lo = q in [256 h + 4 lane, +4) and hi = q in [1020 - 256 h - 4 lane, +4)."""
    e_ls = 1024 - E if edge_l else 0
    out, left, right, state = {}, None, None, None
    if edge_l:
        left = np.zeros(E, F)
    if edge_r:
        right, state = np.zeros(E, F), np.zeros(2 * E, F)
    for lane in range(64):
        inner = lane >= E // 4
        for h in range(2):
            qlo, qhi = 256 * h + 4 * lane, 1020 - 256 * h - 4 * lane
            if edge_l:       # cur[2047 - q] = -pa(q): the quad reversed and negated
                for j in range(4):
                    out[2044 - 256 * h - 4 * lane - e_ls + j] = -pa[qlo + 3 - j]
                    if h == 1 or inner:
                        out[1024 + 256 * h + 4 * lane - e_ls + j] = -pa[qhi + 3 - j]
            if edge_r:       # cur[2048 + q] = pb(q)
                for j in range(4):
                    out[2048 + 256 * h + 4 * lane - e_ls + j] = pb[qlo + j]
                    if h == 1 or inner:
                        out[3068 - 256 * h - 4 * lane - e_ls + j] = pb[qhi + j]
        if not inner:
            qhi = 1020 - 4 * lane
            for j in range(4):
                if edge_l:
                    left[E - 4 - 4 * lane + j] = pa[qhi + j]
                if edge_r:
                    right[E - 4 - 4 * lane + j] = pb[qhi + j]
                    state[E - 4 - 4 * lane + j] = pb[qhi + j]
                    state[E + 4 * lane + j] = pb[qhi + 3 - j]
    return out, left, right, state


def gather_bank_cycles():
    """LDS-array cycles of the gather's 16 ds_write_b64 and 16 ds_read_b64 of one wave (short_model.gather_bank_cycles: banking rules
    of MI355X_MICROARCH.md).  Conflict-free: (64, 32)."""
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
    w = sum(cycles([8 * gather_slot(512 * h + 8 * lane + zz) for lane in range(64)], 16, 32) for h in range(2) for zz in range(8))
    r = 0
    for k4 in range(4):
        for kind in range(4):
            addrs = []
            for lane in range(64):
                q = 2 * int(rev_bits(np.array([128 * (k4 >> 1) + 2 * lane + (k4 & 1)]), 8)[0])
                addrs.append(8 * gather_slot([q, q + 512, 511 - q, 1023 - q][kind]))
            r += cycles(addrs, 32, 64)
    return w, r
