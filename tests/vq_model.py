"""numpy model of k_residue_vq (lewton_amd/csrc/lw_kernels.hip): residue vectors from codeword symbols, with the kernel's
data flow -- accumulation in the bitstream's coordinates, pass after pass, then the mapping to [channel][bin].
Used by the CPU tests to pin the symbol format and the algorithm without a GPU."""
import ctypes as C

import numpy as np

from lewton_amd import _native as N


def symbols(ident, setup, packet, cap=1 << 16):
    ch = ident.audio_channels
    stride = N.lw_setup_floor_stride(setup._h)
    floor = np.zeros((ch, stride), np.uint16)
    ops = np.zeros(cap, np.uint64)
    n = C.c_size_t(0)
    pass_off = (C.c_uint32 * 9)()
    bs, mode, flags = C.c_uint8(0), C.c_uint8(0), C.c_uint8(0)
    curve = np.zeros(ch * (1 << ident.blocksize_1) // 2, np.float32)
    pkt = bytes(packet)
    rc = N.lw_entropy_symbols_host(ident._h, setup._h, pkt, len(pkt), floor.ctypes.data_as(N.u16p),
                                   ops.ctypes.data_as(C.POINTER(C.c_uint64)), cap, C.byref(n), pass_off, C.byref(bs),
                                   C.byref(mode), C.byref(flags), curve.ctypes.data_as(N.f32p))
    return rc, dict(ops=ops[: n.value].copy(), pass_off=list(pass_off), bs=bs.value, mode=mode.value, floor=floor)


def codebook_vq(setup, book):
    dims, entries = C.c_uint32(0), C.c_uint32(0)
    rc = N.lw_setup_codebook_vq(setup._h, book, None, 0, C.byref(dims), C.byref(entries))
    if rc:
        return None
    t = np.zeros(dims.value * entries.value, np.float32)
    assert N.lw_setup_codebook_vq(setup._h, book, t.ctypes.data_as(N.f32p), t.size, None, None) == 0
    return t.reshape(entries.value, dims.value)


def submaps(setup, mode):
    out = []
    for sm in range(16):
        rt, ps, n = C.c_uint8(0), C.c_uint32(0), C.c_size_t(0)
        chans = (C.c_uint8 * 256)()
        if N.lw_setup_submap_info(setup._h, mode, sm, C.byref(rt), C.byref(ps), chans, 256, C.byref(n)):
            break
        out.append((rt.value, ps.value, [chans[i] for i in range(n.value)]))
    return out


def residue_from_symbols(ident, setup, sym):
    """[ch][n/2] f32, the way the kernel builds it."""
    ch = ident.audio_channels
    half = (1 << sym["bs"]) // 2
    sms = submaps(setup, sym["mode"])
    vbase, before = [], 0
    for _t, _p, chans in sms:
        vbase.append(before * half)
        before += len(chans)
    acc = np.zeros(ch * half, np.float32)
    tables = {}
    ops, po = sym["ops"], sym["pass_off"]
    assert po[0] == 0 and po[8] == len(ops) and all(po[i] <= po[i + 1] for i in range(8))
    for p in range(8):
        touched = set()
        for o in ops[po[p]:po[p + 1]]:
            o = int(o)
            coord, book, entry, sm, ps = o & 0xFFFFFF, (o >> 24) & 0xFF, (o >> 32) & 0xFFFFFF, (o >> 56) & 0xF, (o >> 60) & 7
            assert ps == p                                            # sorted by pass
            if book not in tables:
                tables[book] = codebook_vq(setup, book)
            row = tables[book][entry]
            rtype, psize, _chans = sms[sm]
            step = psize // len(row) if rtype == 0 else 1
            for j, e in enumerate(row):
                i = vbase[sm] + coord + j * step
                assert i not in touched                               # one element per position and pass: order-free inside a pass
                touched.add(i)
                acc[i] = np.float32(acc[i] + e)
    out = np.zeros((ch, half), np.float32)
    for sm, (rtype, _psize, chans) in enumerate(sms):
        for pos, c in enumerate(chans):
            if rtype == 2:
                out[c] = acc[vbase[sm] + pos: vbase[sm] + len(chans) * half: len(chans)]
            else:
                out[c] = acc[vbase[sm] + pos * half: vbase[sm] + (pos + 1) * half]
    return out
