"""Mirror of lewton's `audio` module surface (src/audio.rs): same names, argument meaning and errors.

    read_audio_packet(ident, setup, packet, pwr)            -> i16 [ch][m]      (audio.rs:1170)
    read_audio_packet_generic(ident, setup, packet, pwr, S) -> S                (audio.rs:919)
    get_decoded_sample_count(ident, setup, packet)          -> int              (audio.rs:874)
    PreviousWindowRight(), .is_empty(), .clone()                                (audio.rs:847-861)
    AudioReadError                                                               (audio.rs:26-41)

The bit-serial entropy stage runs on the host (C++), everything after it on the GPU (HIP); the call
is synchronous like the reference's.  For throughput use lewton_amd.batch.
"""
import ctypes as C

import numpy as np

from . import _native as N


class AudioReadError(Exception):
    KINDS = {N.AUDIO_END_OF_PACKET: "EndOfPacket", N.AUDIO_BAD_FORMAT: "AudioBadFormat",
             N.AUDIO_IS_HEADER: "AudioIsHeader", N.AUDIO_BUFFER_NOT_ADDRESSABLE: "BufferNotAddressable"}

    def __init__(self, code):
        self.code = code
        self.kind = self.KINDS.get(code, "Library(%d: %s)" % (code, N.device_error()))
        super().__init__(self.kind)


class Decoder:
    """Device context of one (ident, setup) pair on one GPU (lw_decoder)."""

    def __init__(self, ident, setup, device=0):
        err = C.c_int(0)
        self.ident, self.setup, self.device = ident, setup, device
        self._h = N.lw_decoder_create(ident._h, setup._h, device, C.byref(err))
        if not self._h:
            raise RuntimeError("lw_decoder_create failed (%d): %s" % (err.value, N.device_error()))

    def set_shared_device(self, on=True):
        """other decoders' rings (of this process or another) run on this decoder's GPU as well: the rings made for it afterwards
        hand their PCM copies to the device's copier thread and run their launches' kernels in launch order
        (lw_decoder_set_shared_device)"""
        rc = N.lw_decoder_set_shared_device(self._h, 1 if on else 0)
        if rc:
            raise ValueError("lw_decoder_set_shared_device: %d" % rc)

    def set_cu_share(self, part, parts):
        """several decoders on one GPU: this one launches on CUs [32 part / parts, 32 (part + 1) / parts) of every XCD only (the
        rings made for it afterwards; lw_decoder_set_cu_share).  Returns the compute units its batches are planned for."""
        rc = N.lw_decoder_set_cu_share(self._h, part, parts)
        if rc:
            raise ValueError("lw_decoder_set_cu_share(%d, %d): %d" % (part, parts, rc))
        return N.lw_decoder_cu_count(self._h)

    def close(self):
        if getattr(self, "_h", None):
            N.lw_decoder_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


def decoder_for(ident, setup, device=0):
    """One cached Decoder per (ident, setup, device): what `&IdentHeader, &SetupHeader` are to the reference."""
    cache = setup.__dict__.setdefault("_decoders", {})
    key = (id(ident), device)
    if key not in cache:
        cache[key] = Decoder(ident, setup, device)
    return cache[key]


class PreviousWindowRight:
    """The right part of the previous window (audio.rs:847-861); lives in HBM."""

    def __init__(self, _decoder=None, _handle=None):
        self._dec = _decoder
        self._h = _handle

    def _bind(self, dec):
        if self._h is None:
            self._dec = dec
            self._h = N.lw_pwr_new(dec._h)
            if not self._h:
                raise RuntimeError("lw_pwr_new failed: " + N.device_error())
        elif self._dec is not dec:
            raise AssertionError("PreviousWindowRight does not match the ident header")  # audio.rs:916-917
        return self._h

    def is_empty(self):
        return True if self._h is None else bool(N.lw_pwr_is_empty(self._h))

    def clone(self):
        if self._h is None:
            return PreviousWindowRight()
        return PreviousWindowRight(self._dec, N.lw_pwr_clone(self._h))

    def reset(self):
        if self._h is not None:
            N.lw_pwr_reset(self._h)

    def data(self):
        """Host copy [ch][len] of the stored right part, or None (debug / tests)."""
        if self.is_empty():
            return None
        n = N.lw_pwr_len(self._h)
        out = np.zeros((self._dec.ident.audio_channels, n), np.float32)
        rc = N.lw_pwr_copy_to_host(self._h, out.ctypes.data_as(N.f32p))
        if rc:
            raise RuntimeError("lw_pwr_copy_to_host: %d %s" % (rc, N.device_error()))
        return out

    def __del__(self):
        # N is None while the interpreter shuts down; the library owns nothing that outlives the process
        if N is not None and getattr(self, "_h", None) and self._dec is not None and getattr(self._dec, "_h", None):
            N.lw_pwr_free(self._h)
            self._h = None


def get_decoded_sample_count(ident, setup, packet):
    n = C.c_size_t(0)
    rc = N.lw_get_decoded_sample_count(ident._h, setup._h, bytes(packet), len(packet), C.byref(n))
    if rc:
        raise AudioReadError(rc)
    return n.value


_FMT = {"i16": N.FMT_I16_PLANAR, "i16_interleaved": N.FMT_I16_INTERLEAVED, "f32": N.FMT_F32_PLANAR}


def read_audio_packet_generic(ident, setup, packet, pwr, samples="i16", device=0):
    """`samples`: 'i16' (Vec<Vec<i16>>), 'i16_interleaved' (InterleavedSamples<i16>), 'f32' (Vec<Vec<f32>>)."""
    return read_audio_packet_on(decoder_for(ident, setup, device), packet, pwr, samples)


def read_audio_packet_on(dec, packet, pwr, samples="i16"):
    """The same call on an explicit Decoder (one per GPU / per logical shard of a multi-device process)."""
    ident = dec.ident
    h = pwr._bind(dec)
    ch = ident.audio_channels
    cap = (1 << ident.blocksize_1)
    fmt = _FMT[samples]
    out = np.zeros(ch * cap, np.float32 if fmt == N.FMT_F32_PLANAR else np.int16)
    m = C.c_size_t(0)
    pkt = bytes(packet)
    rc = N.lw_read_audio_packet(dec._h, pkt, len(pkt), h, fmt, out.ctypes.data_as(C.c_void_p), cap, C.byref(m))
    if rc:
        raise AudioReadError(rc)
    if fmt == N.FMT_I16_INTERLEAVED:
        return out[: ch * m.value].copy()
    return out[: ch * m.value].reshape(ch, m.value).copy()


def read_audio_packet(ident, setup, packet, pwr):
    return read_audio_packet_generic(ident, setup, packet, pwr, "i16")


def entropy_decode_host(ident, setup, packet):
    """Host entropy stage only (no GPU): returns dict(floor=[ch][stride] u16, residue=[ch][n/2] f32, floor_curve=[ch][n/2] f32
    (rows of floor-0 channels, record entry 0 == 0xFFFE), bs, mode, flags, bits)."""
    ch = ident.audio_channels
    stride = N.lw_setup_floor_stride(setup._h)
    floor = np.zeros((ch, stride), np.uint16)
    cap = ch * (1 << ident.blocksize_1) // 2
    res = np.zeros(cap, np.float32)
    bs, mode, flags, bits = C.c_uint8(0), C.c_uint8(0), C.c_uint8(0), C.c_uint64(0)
    curve = np.zeros(cap, np.float32)
    pkt = bytes(packet)
    rc = N.lw_entropy_decode_host(ident._h, setup._h, pkt, len(pkt), floor.ctypes.data_as(N.u16p),
                                  res.ctypes.data_as(N.f32p), cap, C.byref(bs), C.byref(mode), C.byref(flags),
                                  C.byref(bits), curve.ctypes.data_as(N.f32p))
    if rc:
        raise AudioReadError(rc)
    half = (1 << bs.value) // 2
    return dict(floor=floor, residue=res[: ch * half].reshape(ch, half).copy(),
                floor_curve=curve[: ch * half].reshape(ch, half).copy(), bs=bs.value, mode=mode.value,
                flags=flags.value, bits=bits.value)
