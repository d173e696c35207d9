"""ctypes binding of the CPU oracle (oracle/lewton_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under lewton_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblewton_oracle.so")

OK = 0
AUDIO_END_OF_PACKET, AUDIO_BAD_FORMAT, AUDIO_IS_HEADER, AUDIO_BUFFER_NOT_ADDRESSABLE = 1, 2, 3, 4
HDR_END_OF_PACKET, HDR_NOT_VORBIS, HDR_UNSUPPORTED_VERSION, HDR_BAD_FORMAT = 16, 17, 18, 19
HDR_BAD_TYPE, HDR_IS_AUDIO, HDR_UTF8, HDR_BUFFER_NOT_ADDRESSABLE = 20, 21, 22, 23


def build(force=False):
    src = os.path.join(_HERE, "lewton_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


class Taps(C.Structure):
    _fields_ = [("residue_pre_inverse", C.POINTER(C.c_float)), ("residue_post_inverse", C.POINTER(C.c_float)),
                ("pre_mdct", C.POINTER(C.c_float)), ("post_mdct", C.POINTER(C.c_float)), ("n", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    u8p, f32p, u32p, i16p = C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_int16)
    szp = C.POINTER(C.c_size_t)
    L.lwo_read_header_ident.restype = C.c_void_p
    L.lwo_read_header_ident.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]
    L.lwo_ident_free.argtypes = [C.c_void_p]
    L.lwo_ident_field.restype = C.c_int64
    L.lwo_ident_field.argtypes = [C.c_void_p, C.c_int]
    L.lwo_read_header_setup.restype = C.c_void_p
    L.lwo_read_header_setup.argtypes = [C.c_char_p, C.c_size_t, C.c_uint8, C.c_uint8, C.c_uint8, C.POINTER(C.c_int)]
    L.lwo_setup_free.argtypes = [C.c_void_p]
    L.lwo_setup_count.argtypes = [C.c_void_p, C.c_int]
    L.lwo_pwr_new.restype = C.c_void_p
    L.lwo_pwr_clone.restype = C.c_void_p
    L.lwo_pwr_clone.argtypes = [C.c_void_p]
    L.lwo_pwr_is_empty.argtypes = [C.c_void_p]
    L.lwo_pwr_reset.argtypes = [C.c_void_p]
    L.lwo_pwr_free.argtypes = [C.c_void_p]
    L.lwo_pwr_len.restype = C.c_size_t
    L.lwo_pwr_len.argtypes = [C.c_void_p]
    L.lwo_pwr_copy.argtypes = [C.c_void_p, f32p]
    L.lwo_get_decoded_sample_count.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t, szp]
    L.lwo_read_audio_packet_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, f32p,
                                            C.c_size_t, szp, C.POINTER(Taps)]
    L.lwo_read_audio_packet_i16.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, i16p,
                                            C.c_size_t, szp]
    L.lwo_read_audio_packet_i16_itl.argtypes = L.lwo_read_audio_packet_i16.argtypes
    L.lwo_decode_stream_i16.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64), u32p, C.c_size_t,
                                        C.c_void_p, i16p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    L.lwo_tables.argtypes = [C.c_uint8, f32p, f32p, f32p, f32p, u32p]
    L.lwo_inverse_mdct.argtypes = [C.c_uint8, f32p]
    L.lwo_inverse_mdct_slow.argtypes = [f32p, C.c_size_t]
    L.lwo_render_point.restype = C.c_uint32
    L.lwo_render_point.argtypes = [C.c_uint32] * 5
    L.lwo_low_neighbor.argtypes = [u32p, C.c_size_t, szp, u32p]
    L.lwo_high_neighbor.argtypes = [u32p, C.c_size_t, szp, u32p]
    L.lwo_render_line.restype = C.c_size_t
    L.lwo_render_line.argtypes = [C.c_uint32] * 4 + [u32p]
    L.lwo_floor1_curve.argtypes = [C.c_void_p, C.c_int, u32p, C.c_uint32, f32p, u32p, u8p]
    L.lwo_ilog.restype = C.c_uint8
    L.lwo_ilog.argtypes = [C.c_uint64]
    L.lwo_lookup1_values.restype = C.c_uint32
    L.lwo_lookup1_values.argtypes = [C.c_uint32, C.c_uint16]
    L.lwo_float32_unpack.restype = C.c_float
    L.lwo_float32_unpack.argtypes = [C.c_uint32]
    L.lwo_inverse_couple.argtypes = [C.c_float, C.c_float, f32p, f32p]
    L.lwo_sample_i16.restype = C.c_int16
    L.lwo_sample_i16.argtypes = [C.c_float]
    L.lwo_inverse_db_table.restype = f32p
    L.lwo_bitread_seq.restype = C.c_size_t
    L.lwo_bitread_seq.argtypes = [C.c_char_p, C.c_size_t, u8p, C.c_size_t, C.POINTER(C.c_uint64)]
    L.lwo_huffman_check.argtypes = [u8p, C.c_size_t, C.c_char_p, C.c_size_t, u32p, C.c_size_t, szp]
    _lib = L
    return L


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


REF_PANIC = 64   # LWO_REF_PANIC: the reference panics on this input (no defined result)


class OracleError(Exception):
    def __init__(self, code):
        super().__init__("oracle error %d" % code)
        self.code = code


class Ident:
    def __init__(self, packet):
        err = C.c_int(0)
        self.h = lib().lwo_read_header_ident(bytes(packet), len(packet), C.byref(err))
        if not self.h:
            raise OracleError(err.value)
        f = lambda i: int(lib().lwo_ident_field(self.h, i))
        self.audio_channels, self.audio_sample_rate = f(0), f(1)
        self.bitrate_maximum, self.bitrate_nominal, self.bitrate_minimum = f(2), f(3), f(4)
        self.blocksize_0, self.blocksize_1 = f(5), f(6)

    def __del__(self):
        if getattr(self, "h", None):
            lib().lwo_ident_free(self.h)
            self.h = None


class Setup:
    def __init__(self, packet, ident):
        err = C.c_int(0)
        self.h = lib().lwo_read_header_setup(bytes(packet), len(packet), ident.audio_channels, ident.blocksize_0,
                                             ident.blocksize_1, C.byref(err))
        if not self.h:
            raise OracleError(err.value)

    def count(self, which):
        return lib().lwo_setup_count(self.h, which)

    def __del__(self):
        if getattr(self, "h", None):
            lib().lwo_setup_free(self.h)
            self.h = None


class Pwr:
    """PreviousWindowRight (src/audio.rs:847-861)."""

    def __init__(self, h=None):
        self.h = h if h is not None else lib().lwo_pwr_new()

    def is_empty(self):
        return bool(lib().lwo_pwr_is_empty(self.h))

    def clone(self):
        return Pwr(lib().lwo_pwr_clone(self.h))

    def reset(self):
        lib().lwo_pwr_reset(self.h)

    def data(self, ch):
        n = lib().lwo_pwr_len(self.h)
        if self.is_empty():
            return None
        out = np.zeros((ch, n), np.float32)
        lib().lwo_pwr_copy(self.h, _f32p(out))
        return out

    def __del__(self):
        if getattr(self, "h", None):
            lib().lwo_pwr_free(self.h)
            self.h = None


def get_decoded_sample_count(ident, setup, packet):
    n = C.c_size_t(0)
    rc = lib().lwo_get_decoded_sample_count(ident.h, setup.h, bytes(packet), len(packet), C.byref(n))
    if rc:
        raise OracleError(rc)
    return n.value


def read_audio_packet(ident, setup, packet, pwr, fmt="i16", taps=False):
    """Returns planar [ch][m] (i16 / f32) or interleaved [m*ch] (fmt='i16_itl'); raises OracleError."""
    ch = ident.audio_channels
    cap = 1 << ident.blocksize_1
    m = C.c_size_t(0)
    pkt = bytes(packet)
    if fmt == "f32":
        out = np.zeros((ch, cap), np.float32)
        t = None
        tarr = None
        if taps:
            tarr = {k: np.zeros((ch, cap), np.float32) for k in ("residue_pre_inverse", "residue_post_inverse", "pre_mdct", "post_mdct")}
            t = Taps(_f32p(tarr["residue_pre_inverse"]), _f32p(tarr["residue_post_inverse"]), _f32p(tarr["pre_mdct"]),
                     _f32p(tarr["post_mdct"]), 0)
        rc = lib().lwo_read_audio_packet_f32(ident.h, setup.h, pkt, len(pkt), pwr.h, _f32p(out), cap, C.byref(m),
                                             C.byref(t) if t is not None else None)
        if rc:
            raise OracleError(rc)
        res = out.reshape(-1)[: ch * m.value].reshape(ch, m.value).copy()
        if taps:
            n = t.n
            d = {}
            for k, a in tarr.items():
                w = n if k == "post_mdct" else n // 2
                d[k] = a.reshape(-1)[: ch * w].reshape(ch, w).copy()
            d["n"] = n
            return res, d
        return res
    out = np.zeros(ch * cap, np.int16)
    fn = lib().lwo_read_audio_packet_i16 if fmt == "i16" else lib().lwo_read_audio_packet_i16_itl
    rc = fn(ident.h, setup.h, pkt, len(pkt), pwr.h, out.ctypes.data_as(C.POINTER(C.c_int16)), cap, C.byref(m))
    if rc:
        raise OracleError(rc)
    if fmt == "i16":
        return out[: ch * m.value].reshape(ch, m.value).copy()
    return out[: ch * m.value].copy()


def stage_split(ident, setup, packets):
    """(seconds in the bit-serial stage A2-A5, total seconds) of one perf.rs-shaped pass over `packets`."""
    L = lib()
    L.lwo_stage_timing.argtypes = [C.c_int]
    L.lwo_stage_entropy_seconds.restype = C.c_double
    L.lwo_stage_timing(1)
    try:
        _, _, total = decode_stream_i16(ident, setup, packets, keep=False)
        return L.lwo_stage_entropy_seconds(), total
    finally:
        L.lwo_stage_timing(0)


def decode_stream_i16(ident, setup, packets, pwr=None, keep=True):
    """perf.rs-shaped loop over a list of packets; returns (list-concatenated planar-per-packet i16, total, seconds)."""
    pwr = pwr or Pwr()
    data = b"".join(bytes(p) for p in packets)
    lens = np.array([len(p) for p in packets], np.uint32)
    offs = np.zeros(len(packets), np.uint64)
    if len(packets) > 1:
        offs[1:] = np.cumsum(lens[:-1].astype(np.uint64))
    ch = ident.audio_channels
    cap = ch * (1 << ident.blocksize_1) // 2 * len(packets) + 16
    out = np.zeros(cap if keep else 1, np.int16)
    tot = C.c_uint64(0)
    sec = C.c_double(0)
    rc = lib().lwo_decode_stream_i16(ident.h, setup.h, data, offs.ctypes.data_as(C.POINTER(C.c_uint64)),
                                     lens.ctypes.data_as(C.POINTER(C.c_uint32)), len(packets), pwr.h,
                                     out.ctypes.data_as(C.POINTER(C.c_int16)) if keep else None, cap, C.byref(tot),
                                     C.byref(sec))
    if rc:
        raise OracleError(rc)
    return (out[: tot.value * ch] if keep else None), tot.value, sec.value


def tables(bs):
    n = 1 << bs
    A, B = np.zeros(n // 2, np.float32), np.zeros(n // 2, np.float32)
    Ct, W = np.zeros(n // 4, np.float32), np.zeros(n // 2, np.float32)
    br = np.zeros(n // 8, np.uint32)
    lib().lwo_tables(bs, _f32p(A), _f32p(B), _f32p(Ct), _f32p(W), br.ctypes.data_as(C.POINTER(C.c_uint32)))
    return A, B, Ct, W, br


def inverse_mdct(spectrum_half, bs):
    """spectrum_half: n/2 f32; returns n f32 (zero-extended then transformed in place like audio.rs:1044-1052)."""
    n = 1 << bs
    buf = np.zeros(n, np.float32)
    buf[: n // 2] = np.asarray(spectrum_half, np.float32)
    lib().lwo_inverse_mdct(bs, _f32p(buf))
    return buf


def inverse_mdct_slow(spectrum_half):
    n = 2 * len(spectrum_half)
    buf = np.zeros(n, np.float32)
    buf[: n // 2] = np.asarray(spectrum_half, np.float32)
    lib().lwo_inverse_mdct_slow(_f32p(buf), n)
    return buf


def inverse_db_table():
    p = lib().lwo_inverse_db_table()
    return np.ctypeslib.as_array(p, shape=(256,)).copy()
