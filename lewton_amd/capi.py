"""ctypes view of lewton's C API (include/lewton.h = src/capi.rs:13-147) as exported by liblewton_amd.so --
what an FFmpeg-style C caller sees.  Thin on purpose: tests and examples drive the raw symbols."""
import ctypes as C

import numpy as np

from . import _native as N

_L = N.lib
CAPI_SYMBOLS = {
    "lewton_context_from_extradata": (C.c_void_p, [C.c_char_p, C.c_size_t]),
    "lewton_context_reset": (None, [C.c_void_p]),
    "lewton_decode_packet": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "lewton_samples_count": (C.c_size_t, [C.c_void_p]),
    "lewton_samples_f32": (C.POINTER(C.c_float), [C.c_void_p, C.c_size_t]),
    "lewton_samples_drop": (None, [C.c_void_p]),
    "lewton_context_drop": (None, [C.c_void_p]),
}
for _n, (_r, _a) in CAPI_SYMBOLS.items():
    _f = getattr(_L, _n)
    _f.restype, _f.argtypes = _r, _a
    globals()[_n] = _f


def xiph_lace(n):
    return b"\xff" * (n // 255) + bytes([n % 255])


def make_extradata(ident, comment, setup):
    """Matroska CodecPrivate for Vorbis: 0x02, xiph-laced lengths of the first two headers, then the three headers."""
    return b"\x02" + xiph_lace(len(ident)) + xiph_lace(len(comment)) + bytes(ident) + bytes(comment) + bytes(setup)


def decode_packet(ctx, pkt, max_channels=256):
    """(rc, [channel arrays] | None) through lewton_decode_packet / lewton_samples_*."""
    out = C.c_void_p()
    pkt = bytes(pkt)
    rc = lewton_decode_packet(ctx, pkt, len(pkt), C.byref(out))  # noqa: F821
    if rc:
        return rc, None
    n = lewton_samples_count(out)  # noqa: F821
    chans = []
    for c in range(max_channels):
        p = lewton_samples_f32(out, c)  # noqa: F821
        if not p:
            break
        chans.append(np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.float32))
    lewton_samples_drop(out)  # noqa: F821
    return 0, chans
