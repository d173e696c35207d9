"""Lane-level numpy model of the short-block kernel k_short (lw_kernels_long.hip), n = 256.

One wave = 64 lanes = 8 blocks ("slots") x 8 lanes; every per-lane register is a numpy array of shape [64].  The model
follows the kernel's data movement (layouts B' / D' / E' of lw_fast.hpp, the half-mirror exchange of step 1, the 8 x 8
register <-> lane transpose, the LDS bit-reverse gather) and its exact packed f32 operations (fast_model.pk), consumes the
product's own LDS image (lw_debug_short_image) and must reproduce the oracle bit for bit (tests/test_short_model.py)."""
import numpy as np

from fast_model import F, bfly_pk, last3_pk, ola_pk, pk, rev_bits, step7_block_pk, step8_pk

N = 256
N2, N4, N8, P = N // 2, N // 4, N // 8, N // 4   # P = complex pairs per block
LANES = np.arange(64)
G, L = LANES >> 3, LANES & 7               # slot of a lane, lane inside the slot

# byte offsets of the image sections (LWS_* in lw_fast.hpp)
LWS = dict(apair=0, tw_s2=512, tw_l0=768, tw_l1=896, a2=960, c4=976, b_lo=1232, b_hi=1488, win=1744, inv_db=2256, xsf=3280,
           sid16=3792, total=5120)


class Image:
    def __init__(self, blob):
        f = np.frombuffer(blob, np.float32)
        o = {k: v // 4 for k, v in LWS.items()}
        self.apair = f[o["apair"]: o["apair"] + 128].reshape(64, 2)
        self.tw_s2 = f[o["tw_s2"]: o["tw_s2"] + 64].reshape(4, 8, 2)
        self.tw_l0 = f[o["tw_l0"]: o["tw_l0"] + 32].reshape(2, 8, 2)
        self.tw_l1 = f[o["tw_l1"]: o["tw_l1"] + 16].reshape(8, 2)
        self.a2 = f[o["a2"]]
        self.c4 = f[o["c4"]: o["c4"] + 64].reshape(2, 8, 4)
        self.b_lo = f[o["b_lo"]: o["b_lo"] + 64].reshape(2, 8, 4)
        self.b_hi = f[o["b_hi"]: o["b_hi"] + 64].reshape(2, 8, 4)
        self.win = f[o["win"]: o["win"] + 128].reshape(2, 8, 8)
        self.inv_db = f[o["inv_db"]: o["inv_db"] + 256]
        self.xsf = f[o["xsf"]: o["xsf"] + 128].reshape(2, 64)
        self.sid16 = np.frombuffer(blob, np.uint16)[LWS["sid16"] // 2: LWS["sid16"] // 2 + 256].reshape(2, 4, 8, 4)


def slot_t4(g, p):
    """LDS slot (in pairs) of pair p of block g for the bit-reverse gather."""
    return 64 * g + p


def imdct_wave(X, img, prev_pb=None):
    """X: [8][128] spectra (one channel of eight blocks).  Returns [8][256] time-domain blocks; with prev_pb ([8][64]: right
    part pb(0..63) of each block's predecessor) also the [8][128] overlap-added samples (audio.rs:1116-1118)."""
    X = np.asarray(X, F)
    # ---- load layout: lane (g, l) holds float4 groups m = 8x + l, x = 0..3, of block g
    Pq = [None] * 8   # layout B': Pq[x] = pair p = 8x + l
    up = [None] * 4
    for x in range(4):
        m = 8 * x + L
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
    mirror = 8 * G + (7 - L)                                                # DPP row_half_mirror
    for xs in range(4):
        Pq[7 - xs] = up[xs][:, mirror]
    # ---- step 2 and stages l = 0, 1: register-local (pair bits 5, 4, 3)
    for x in range(4):
        Pq[x + 4], Pq[x] = bfly_pk(Pq[x + 4], Pq[x], img.tw_s2[x][L].T)
    for x in (2, 3, 6, 7):
        Pq[x], Pq[x - 2] = bfly_pk(Pq[x], Pq[x - 2], img.tw_l0[x & 1][L].T)
    for x in (1, 3, 5, 7):
        Pq[x], Pq[x - 1] = bfly_pk(Pq[x], Pq[x - 1], img.tw_l1[L].T)
    # ---- 8 x 8 transpose register index <-> lane inside the block (t3_inreg): lane nu = p[5:3], reg z = p[2:0]
    Z = []
    for zz in range(8):
        src_lane = 8 * G + zz
        sel = np.stack([Pq[x][:, src_lane] for x in range(8)])              # [x][2][64]
        Z.append(sel[L, :, LANES].T.copy())                                  # value of reg x = nu(this lane) on lane zz
    a2 = np.stack([np.full(64, img.a2, F)] * 2)
    Z = last3_pk(Z, a2)
    # ---- bit-reverse gather through LDS
    lds = np.zeros((8 * 64, 2), F)
    for zz in range(8):
        lds[slot_t4(G, 8 * L + zz)] = Z[zz].T
    pa = np.zeros((8, N4), F)
    pb = np.zeros((8, N4), F)
    out_ola = np.zeros((8, N2), F)
    for c in range(2):
        mp = 2 * L + c
        q2 = 2 * rev_bits(mp, 4)
        pq, pq32 = lds[slot_t4(G, q2)].T.copy(), lds[slot_t4(G, q2 + P // 2)].T.copy()
        p31, p63 = lds[slot_t4(G, P // 2 - 1 - q2)].T.copy(), lds[slot_t4(G, P - 1 - q2)].T.copy()
        C = img.c4[c][L]
        Dn1, En1 = step7_block_pk(p63, pq, np.stack([C[:, 0], C[:, 1]]))
        Dn2, En2 = step7_block_pk(p31, pq32, np.stack([C[:, 2], C[:, 3]]))
        Bl, Bh = img.b_lo[c][L], img.b_hi[c][L]
        R = [step8_pk(Dn1, np.stack([Bl[:, 0], Bl[:, 1]])), step8_pk(Dn2, np.stack([Bl[:, 2], Bl[:, 3]])),
             step8_pk(En2, np.stack([Bh[:, 0], Bh[:, 1]])), step8_pk(En1, np.stack([Bh[:, 2], Bh[:, 3]]))]
        qs = [P - 1 - 2 * mp, P - 2 - 2 * mp, 1 + 2 * mp, 2 * mp]
        for k in range(4):
            pa[G, qs[k]] = R[k][0]
            pb[G, qs[k]] = R[k][1]
            if prev_pb is not None:
                PP = np.stack([np.asarray(prev_pb, F)[G, qs[k]], np.zeros(64, F)])
                S2 = np.stack([img.win[c][L][:, 2 * k], img.win[c][L][:, 2 * k + 1]])
                O = ola_pk(R[k], PP, 0, S2)
                out_ola[G, qs[k]] = O[0]
                out_ola[G, N2 - 1 - qs[k]] = O[1]
    out = np.zeros((8, N), F)
    q = np.arange(N4)
    out[:, q] = pa
    out[:, N2 - 1 - q] = -pa
    out[:, N2 + q] = pb
    out[:, N - 1 - q] = pb
    if prev_pb is not None:
        return out, out_ola, pb
    return out


def floor_group_model(rec, xs, inv_db, sid16_slot):
    """Per-bin floor of one block-channel the way k_short builds it: active-post mask, one interval entry per post
    {dy, 0.5 sgn(dy) - x0 dy, 1/adx, y0}, then y(k) = y0 + trunc((k dy + c0) (1/adx)) with the static interval index of
    the image (sid16_slot: [4][8][4] u16 = 16 * interval of bin 4(8x + l) + j)."""
    Fp = len(xs)
    if rec[0] == 0xFFFF:
        return np.zeros(N2, F)
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
    out = np.zeros(N2, F)
    for x in range(4):
        for l in range(8):
            for j in range(4):
                k = 4 * (8 * x + l) + j
                dy, c0, rinv, y0 = ent[int(sid16_slot[x][l][j]) // 16]
                z = F(F(k) * dy + c0)          # one fma in the kernel; exact either way (|k dy| < 2^18)
                q = int(np.trunc(F(z * rinv)))
                out[k] = inv_db[y0 + q]
    return out
