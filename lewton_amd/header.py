"""Mirror of lewton's `header` module surface that the audio path needs (src/header.rs).

`read_header_ident`, `read_header_comment`, `read_header_setup` have the reference's names, argument
meaning and error kinds; the parsing itself runs in the C++ host library."""
import ctypes as C

from . import _native as N


class HeaderReadError(Exception):
    """src/header.rs:35-63"""
    KINDS = {N.HDR_END_OF_PACKET: "EndOfPacket", N.HDR_NOT_VORBIS: "NotVorbisHeader",
             N.HDR_UNSUPPORTED_VERSION: "UnsupportedVorbisVersion", N.HDR_BAD_FORMAT: "HeaderBadFormat",
             N.HDR_BAD_TYPE: "HeaderBadType", N.HDR_IS_AUDIO: "HeaderIsAudio", N.HDR_UTF8: "Utf8DecodeError",
             N.HDR_BUFFER_NOT_ADDRESSABLE: "BufferNotAddressable"}

    def __init__(self, code):
        self.code = code
        self.kind = self.KINDS.get(code, "Unknown(%d)" % code)
        super().__init__(self.kind)


class IdentHeader:
    """src/header.rs:188-211"""

    def __init__(self, handle):
        self._h = handle
        info = N.IdentInfo()
        N.lw_ident_get_info(handle, C.byref(info))
        self.audio_channels = info.audio_channels
        self.audio_sample_rate = info.audio_sample_rate
        self.bitrate_maximum = info.bitrate_maximum
        self.bitrate_nominal = info.bitrate_nominal
        self.bitrate_minimum = info.bitrate_minimum
        self.blocksize_0 = info.blocksize_0
        self.blocksize_1 = info.blocksize_1

    def __del__(self):
        if getattr(self, "_h", None):
            N.lw_ident_free(self._h)
            self._h = None


class SetupHeader:
    """src/header.rs:471-481 (opaque)"""

    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        if getattr(self, "_h", None):
            N.lw_setup_free(self._h)
            self._h = None


class CommentHeader:
    """src/header.rs:289-300"""

    def __init__(self, vendor, comment_list):
        self.vendor = vendor
        self.comment_list = comment_list


def read_header_ident(packet):
    err = C.c_int(0)
    h = N.lw_read_header_ident(bytes(packet), len(packet), C.byref(err))
    if not h:
        raise HeaderReadError(err.value)
    return IdentHeader(h)


def read_header_setup(packet, audio_channels, blocksizes):
    err = C.c_int(0)
    h = N.lw_read_header_setup(bytes(packet), len(packet), audio_channels, blocksizes[0], blocksizes[1], C.byref(err))
    if not h:
        raise HeaderReadError(err.value)
    return SetupHeader(h)


def read_header_comment(packet):
    err = C.c_int(0)
    h = N.lw_read_header_comment(bytes(packet), len(packet), C.byref(err))
    if not h:
        raise HeaderReadError(err.value)
    try:
        n = C.c_size_t(0)
        p = N.lw_comment_vendor(h, C.byref(n))
        vendor = C.string_at(p, n.value).decode("utf-8")
        out = []
        for i in range(N.lw_comment_count(h)):
            k, v, kl, vl = C.c_void_p(), C.c_void_p(), C.c_size_t(), C.c_size_t()
            N.lw_comment_get(h, i, C.byref(k), C.byref(kl), C.byref(v), C.byref(vl))
            out.append((C.string_at(k, kl.value).decode("utf-8"), C.string_at(v, vl.value).decode("utf-8")))
        return CommentHeader(vendor, out)
    finally:
        N.lw_comment_free(h)
