"""Mirror of lewton's `inside_ogg` module (src/inside_ogg.rs): `read_headers` and `OggStreamReader` with the
reference's method names, argument meaning and error kinds, executed by the C++ host layer (lw_ogg.cpp) and the
HIP decode path.  `read_dec_packets` is the look-ahead queue of INTEGRATION.md section 3 (an extension: many
packets, one batch of kernel launches)."""
import ctypes as C

import numpy as np

from . import _native as N
from . import header as H
from .audio import _FMT, AudioReadError
from .ogg import OggReadError, PacketReader


class VorbisError(Exception):
    """src/lib.rs:120-157: BadAudio(AudioReadError) | BadHeader(HeaderReadError) | OggError(OggReadError)"""

    def __init__(self, code):
        self.code = code
        if 1 <= code <= 4:
            self.kind, self.inner = "BadAudio", AudioReadError(code)
        elif 16 <= code <= 23:
            self.kind, self.inner = "BadHeader", H.HeaderReadError(code)
        elif 49 <= code <= 53:
            self.kind, self.inner = "OggError", OggReadError(code)
        else:
            self.kind, self.inner = "Library", RuntimeError("%d %s" % (code, N.device_error()))
        super().__init__("%s(%s)" % (self.kind, self.inner))


def read_headers(rdr):
    """inside_ogg.rs:30-49: ((ident_hdr, comment_hdr, setup_hdr), stream_serial) from a PacketReader."""
    try:
        pck = rdr.read_packet_expected()
        ident = H.read_header_ident(pck.data)
        serial = pck.stream_serial()
        pck = rdr.read_packet_expected()
        while pck.stream_serial() != serial:
            pck = rdr.read_packet_expected()
        comment = H.read_header_comment(pck.data)
        pck = rdr.read_packet_expected()
        while pck.stream_serial() != serial:
            pck = rdr.read_packet_expected()
        setup = H.read_header_setup(pck.data, ident.audio_channels, (ident.blocksize_0, ident.blocksize_1))
    except OggReadError as e:
        raise VorbisError(e.code)
    except H.HeaderReadError as e:
        raise VorbisError(e.code)
    rdr.delete_unread_packets()
    return (ident, comment, setup), pck.stream_serial()


class _Borrowed:
    """header handles owned by the C++ stream object"""

    def __del__(self):
        self._h = None


class _BorrowedIdent(_Borrowed, H.IdentHeader):
    pass


class _BorrowedSetup(_Borrowed, H.SetupHeader):
    pass


def _comment_of(h):
    n = C.c_size_t(0)
    p = N.lw_comment_vendor(h, C.byref(n))
    vendor = C.string_at(p, n.value).decode("utf-8")
    out = []
    for i in range(N.lw_comment_count(h)):
        k, v, kl, vl = C.c_void_p(), C.c_void_p(), C.c_size_t(), C.c_size_t()
        N.lw_comment_get(h, i, C.byref(k), C.byref(kl), C.byref(v), C.byref(vl))
        out.append((C.string_at(k, kl.value).decode("utf-8"), C.string_at(v, vl.value).decode("utf-8")))
    return H.CommentHeader(vendor, out)


class OggStreamReader:
    """inside_ogg.rs:66-314.  `src`: bytes, a path, a binary file object, or a lewton_amd.ogg.PacketReader
    (`from_ogg_reader`)."""

    def __init__(self, src, device=0):
        rdr = src if isinstance(src, PacketReader) else PacketReader(src)
        self._src_keep = rdr  # keeps the byte source / callbacks alive
        err = C.c_int(0)
        self._h = N.lw_ogg_stream_open(rdr._take(), device, C.byref(err))
        if not self._h:
            raise VorbisError(err.value)
        self._refresh_headers()

    new = classmethod(lambda cls, rdr, device=0: cls(rdr, device))
    from_ogg_reader = classmethod(lambda cls, rdr, device=0: cls(rdr, device))

    def _refresh_headers(self):
        self._link = N.lw_ogg_stream_link_index(self._h)
        self.ident_hdr = _BorrowedIdent(N.lw_ogg_stream_ident(self._h))
        self.setup_hdr = _BorrowedSetup(N.lw_ogg_stream_setup(self._h))
        self.comment_hdr = _comment_of(N.lw_ogg_stream_comment(self._h))

    def _after_call(self):
        if N.lw_ogg_stream_link_index(self._h) != self._link:
            self._refresh_headers()  # chained stream: the context was re-initialised (inside_ogg.rs:120-151)

    def _buffers(self, samples, packets=1):
        fmt = _FMT[samples]
        ch = self.ident_hdr.audio_channels
        cap = 1 << self.ident_hdr.blocksize_1
        return fmt, ch, cap, np.zeros(ch * cap * packets, np.float32 if fmt == N.FMT_F32_PLANAR else np.int16)

    @staticmethod
    def _shape(out, fmt, ch, m):
        if fmt == N.FMT_I16_INTERLEAVED:
            return out[: ch * m].copy()
        return out[: ch * m].reshape(ch, m).copy()

    def read_dec_packet_generic(self, samples="i16"):
        """Ok(None) -> None at the end of the stream."""
        m = C.c_size_t(0)
        for _attempt in range(3):
            fmt, ch, cap, out = self._buffers(samples)
            rc = N.lw_ogg_stream_read_dec_packet(self._h, fmt, out.ctypes.data_as(C.c_void_p), out.size, C.byref(m))
            self._after_call()
            if rc != N.ERR_CAPACITY:  # else: the next link of a chained file needs a larger buffer (at most once per call)
                break
        if rc == N.OGG_EOF:
            return None
        if rc:
            raise VorbisError(rc)
        return self._shape(out, fmt, self.ident_hdr.audio_channels, m.value)

    def read_dec_packet(self):
        return self.read_dec_packet_generic("i16")

    def read_dec_packet_itl(self):
        return self.read_dec_packet_generic("i16_interleaved")

    def set_entropy_on_device(self, on=True):
        """look-ahead batches decode their floors and residues on the GPU (k_entropy) when the stream is eligible"""
        N.lw_ogg_stream_set_entropy_on_device(self._h, 1 if on else 0)

    def set_read_ahead(self, max_packets, n_threads=0):
        """read_dec_packet* hand out the packets of batches of up to max_packets decoded ahead by the look-ahead pipeline
        (lw_ogg_stream_set_read_ahead): the same sequence of results, call for call, at the batched rate.  0 = off (default)."""
        rc = N.lw_ogg_stream_set_read_ahead(self._h, int(max_packets), int(n_threads))
        if rc:
            raise VorbisError(rc)

    def read_dec_packets(self, max_packets, samples="i16", n_threads=0):
        """Look-ahead queue: up to max_packets packets with one batch.  Returns a list of (samples | AudioReadError);
        [] in front of a chain boundary (call read_dec_packet_generic to cross it), None at the end of the stream."""
        fmt, ch, cap, out = self._buffers(samples, max_packets)
        ns = (C.c_uint32 * max_packets)()
        st = (C.c_int32 * max_packets)()
        n = C.c_size_t(0)
        rc = N.lw_ogg_stream_read_dec_packets(self._h, fmt, max_packets, n_threads, out.ctypes.data_as(C.c_void_p),
                                              out.size, ns, st, C.byref(n))
        if rc == N.OGG_EOF:
            return None
        if rc:
            raise VorbisError(rc)
        res, o = [], 0
        for i in range(n.value):
            if st[i]:
                res.append(AudioReadError(st[i]))
                continue
            res.append(self._shape(out[o:], fmt, ch, ns[i]))
            o += ch * ns[i]
        return res

    def skip_samples_linear(self, to_skip, samples="i16"):
        """inside_ogg.rs:244-283: (Some(packet) | None, leftover)."""
        m, left, got = C.c_size_t(0), C.c_size_t(0), C.c_int(0)
        for _attempt in range(3):   # (LW_ERR_CAPACITY: the next link of a chained file needs a larger buffer, at most once per call)
            fmt, ch, cap, out = self._buffers(samples)
            rc = N.lw_ogg_stream_skip_samples_linear(self._h, to_skip, fmt, out.ctypes.data_as(C.c_void_p), out.size,
                                                     C.byref(m), C.byref(left), C.byref(got))
            self._after_call()
            if rc != N.ERR_CAPACITY:
                break
            to_skip = left.value
        if rc:
            raise VorbisError(rc)
        if not got.value:
            return None, left.value
        return self._shape(out, fmt, self.ident_hdr.audio_channels, m.value), left.value

    def stream_serial(self):
        return N.lw_ogg_stream_serial(self._h)

    def get_last_absgp(self):
        v = C.c_uint64(0)
        return v.value if N.lw_ogg_stream_last_absgp(self._h, C.byref(v)) else None

    def seek_absgp_pg(self, absgp):
        rc = N.lw_ogg_stream_seek_absgp_pg(self._h, absgp)
        if rc:
            raise VorbisError(rc)

    def close(self):
        if getattr(self, "_h", None):
            N.lw_ogg_stream_close(self._h)
            self._h = None

    def __del__(self):
        if N is not None and getattr(N, "lw_ogg_stream_close", None) is not None:
            self.close()
