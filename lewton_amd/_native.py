"""ctypes binding of lewton_amd/_lib/liblewton_amd.so (the C ABI of include/lewton_amd.h).

The library is the product: there is no Python or CPU fallback.  Importing this module fails loudly
when the HIP library has not been built (run `python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "liblewton_amd.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "lewton_amd: %s is missing -- build the HIP extension first (python lewton_amd/build.py); "
        "there is no fallback path" % LIB_PATH)

lib = C.CDLL(LIB_PATH)

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
f32p = C.POINTER(C.c_float)
szp = C.POINTER(C.c_size_t)
intp = C.POINTER(C.c_int)


class IdentInfo(C.Structure):
    _fields_ = [("audio_channels", C.c_uint8), ("audio_sample_rate", C.c_uint32), ("bitrate_maximum", C.c_int32),
                ("bitrate_nominal", C.c_int32), ("bitrate_minimum", C.c_int32), ("blocksize_0", C.c_uint8),
                ("blocksize_1", C.c_uint8)]


class Packet(C.Structure):
    _fields_ = [("data", C.c_void_p), ("len", C.c_size_t), ("pwr", C.c_void_p)]


class PacketResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_samples", C.c_uint32), ("out_offset", C.c_uint64)]


class ShardPacket(C.Structure):
    _fields_ = [("stream", C.c_void_p), ("data", C.c_void_p), ("len", C.c_size_t)]


class PwrState(C.Structure):
    _fields_ = [("present", C.c_uint8), ("parity", C.c_uint8), ("len", C.c_uint32)]


class OggPacket(C.Structure):
    _fields_ = [("data", C.c_void_p), ("len", C.c_size_t), ("stream_serial", C.c_uint32), ("absgp_page", C.c_uint64),
                ("first_in_stream", C.c_uint8), ("last_in_stream", C.c_uint8), ("first_in_page", C.c_uint8),
                ("last_in_page", C.c_uint8)]


OGG_READ_FN = C.CFUNCTYPE(C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t)
OGG_SEEK_FN = C.CFUNCTYPE(C.c_int64, C.c_void_p, C.c_int64, C.c_int)


class OggIo(C.Structure):
    _fields_ = [("read", OGG_READ_FN), ("seek", OGG_SEEK_FN), ("user", C.c_void_p)]


def _sig(name, restype, argtypes):
    fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
    fn.restype = restype
    fn.argtypes = argtypes
    return fn


# every symbol include/lewton_amd.h declares
SYMBOLS = {
    "lw_version": (C.c_char_p, []),
    "lw_default_host_threads": (C.c_int, []),
    "lw_last_device_error": (C.c_char_p, []),
    "lw_read_header_ident": (C.c_void_p, [C.c_char_p, C.c_size_t, intp]),
    "lw_ident_get_info": (C.c_int, [C.c_void_p, C.POINTER(IdentInfo)]),
    "lw_ident_free": (None, [C.c_void_p]),
    "lw_read_header_setup": (C.c_void_p, [C.c_char_p, C.c_size_t, C.c_uint8, C.c_uint8, C.c_uint8, intp]),
    "lw_setup_free": (None, [C.c_void_p]),
    "lw_read_header_comment": (C.c_void_p, [C.c_char_p, C.c_size_t, intp]),
    "lw_comment_vendor": (C.c_void_p, [C.c_void_p, szp]),
    "lw_comment_count": (C.c_size_t, [C.c_void_p]),
    "lw_comment_get": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), szp, C.POINTER(C.c_void_p), szp]),
    "lw_comment_free": (None, [C.c_void_p]),
    "lw_device_count": (C.c_int, []),
    "lw_decoder_create": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_int, intp]),
    "lw_decoder_destroy": (None, [C.c_void_p]),
    "lw_pwr_new": (C.c_void_p, [C.c_void_p]),
    "lw_pwr_is_empty": (C.c_int, [C.c_void_p]),
    "lw_pwr_clone": (C.c_void_p, [C.c_void_p]),
    "lw_pwr_reset": (None, [C.c_void_p]),
    "lw_pwr_free": (None, [C.c_void_p]),
    "lw_pwr_len": (C.c_size_t, [C.c_void_p]),
    "lw_pwr_copy_to_host": (C.c_int, [C.c_void_p, f32p]),
    "lw_get_decoded_sample_count": (C.c_int, [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t, szp]),
    "lw_read_audio_packet": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_size_t, szp]),
    "lw_setup_floor_stride": (C.c_uint32, [C.c_void_p]),
    "lw_entropy_decode_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t, u16p, f32p, C.c_size_t, u8p,
                                         u8p, u8p, C.POINTER(C.c_uint64), f32p]),
    "lw_setup_codebook_vq": (C.c_int, [C.c_void_p, C.c_uint, f32p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "lw_huffman_check": (C.c_int, [u8p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.c_size_t, szp]),
    "lw_debug_imdct": (C.c_int, [C.c_void_p, C.c_int, f32p, f32p]),
    "lw_debug_fast_image": (C.c_size_t, [C.c_void_p, C.c_void_p, u8p, C.c_size_t, C.POINTER(C.c_uint32)]),
    "lw_debug_plan_census": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]),
    "lw_debug_short_image": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_int, u8p, C.c_size_t, u8p, szp, C.POINTER(C.c_uint32)]),
    "lw_batch_create": (C.c_void_p, [C.c_void_p, C.c_size_t, C.c_int, intp]),
    "lw_batch_destroy": (None, [C.c_void_p]),
    "lw_batch_entropy": (C.c_int, [C.c_void_p, C.POINTER(Packet), C.c_size_t, C.c_int]),
    "lw_batch_upload": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lw_batch_synth": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "lw_batch_synth_to_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "lw_batch_size": (C.c_size_t, [C.c_void_p]),
    "lw_batch_out_elems": (C.c_size_t, [C.c_void_p]),
    "lw_batch_results": (C.POINTER(PacketResult), [C.c_void_p]),
    "lw_batch_algorithmic_bytes": (C.c_uint64, [C.c_void_p]),
    "lw_batch_state_bytes": (C.c_uint64, [C.c_void_p]),
    "lw_batch_tap": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, f32p, C.c_size_t]),
    "lw_batch_set_force_generic": (None, [C.c_void_p, C.c_int]),
    "lw_debug_batch_set_rounds": (None, [C.c_void_p, C.c_int]),
    "lw_debug_batch_set_mix": (None, [C.c_void_p, C.c_int]),
    "lw_debug_batch_break_mix": (None, [C.c_void_p, C.c_uint]),
    "lw_debug_break_mix": (None, [C.c_uint]),
    "lw_debug_batch_set_halo": (None, [C.c_void_p, C.c_int]),
    "lw_debug_batch_set_long10": (None, [C.c_void_p, C.c_int]),
    "lw_batch_device_status": (C.c_int, [C.c_void_p]),
    "lw_batch_last_kernels": (C.c_char_p, [C.c_void_p]),
    "lw_decoder_supports_device_entropy": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p)]),
    "lw_batch_set_entropy_on_device": (C.c_int, [C.c_void_p, C.c_int]),
    "lw_batch_device_entropy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lw_ring_create": (C.c_void_p, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, intp]),
    "lw_ring_destroy": (None, [C.c_void_p]),
    "lw_ring_stage": (C.c_int, [C.c_void_p, C.POINTER(Packet), C.c_size_t, C.c_int]),
    "lw_ring_launch": (C.c_int, [C.c_void_p]),
    "lw_ring_submit": (C.c_int, [C.c_void_p, C.POINTER(Packet), C.c_size_t, C.c_int]),
    "lw_ring_collect": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(PacketResult)), szp, C.POINTER(C.c_void_p), szp]),
    "lw_ring_release": (C.c_int, [C.c_void_p]),
    "lw_ring_drain": (C.c_int, [C.c_void_p]),
    "lw_ring_slots": (C.c_size_t, [C.c_void_p]),
    "lw_ring_in_flight": (C.c_size_t, [C.c_void_p]),
    "lw_ring_last_staged_elems": (C.c_size_t, [C.c_void_p]),
    "lw_ring_set_entropy_on_device": (C.c_int, [C.c_void_p, C.c_int]),
    "lw_ring_last_kernels": (C.c_char_p, [C.c_void_p]),
    "lw_sharder_create": (C.c_void_p, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_size_t, C.c_size_t, C.c_int, intp]),
    "lw_sharder_destroy": (None, [C.c_void_p]),
    "lw_sharder_set_entropy_on_device": (C.c_int, [C.c_void_p, C.c_int]),
    "lw_sharder_shards": (C.c_size_t, [C.c_void_p]),
    "lw_sharder_shard_of": (C.c_size_t, [C.c_void_p, C.c_uint64]),
    "lw_sharder_device_of": (C.c_int, [C.c_void_p, C.c_size_t]),
    "lw_sharder_stream_open": (C.c_void_p, [C.c_void_p, C.c_uint64]),
    "lw_sharder_stream_close": (None, [C.c_void_p]),
    "lw_sharder_stream_reset": (None, [C.c_void_p]),
    "lw_sharder_decode": (C.c_int, [C.c_void_p, C.POINTER(ShardPacket), C.c_size_t, C.c_int, C.c_void_p, C.c_size_t,
                                    C.POINTER(PacketResult)]),
    "lw_sharder_submit": (C.c_int, [C.c_void_p, C.POINTER(ShardPacket), C.c_size_t, C.c_int, szp]),
    "lw_sharder_collect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(PacketResult), C.c_size_t]),
    "lw_sharder_in_flight": (C.c_size_t, [C.c_void_p]),
    "lw_sharder_collect_pinned": (C.c_int, [C.c_void_p, C.POINTER(PacketResult), C.c_size_t, C.POINTER(C.c_void_p), szp]),
    "lw_sharder_release": (C.c_int, [C.c_void_p]),
    "lw_pwr_get_state": (None, [C.c_void_p, C.POINTER(PwrState)]),
    "lw_pwr_set_state": (None, [C.c_void_p, C.POINTER(PwrState)]),
    "lw_decoder_device": (C.c_int, [C.c_void_p]),
    "lw_decoder_set_cu_share": (C.c_int, [C.c_void_p, C.c_uint, C.c_uint]),
    "lw_decoder_cu_count": (C.c_int, [C.c_void_p]),
    "lw_decoder_set_shared_device": (C.c_int, [C.c_void_p, C.c_int]),
    "lw_debug_sharder_share_cus": (None, [C.c_int]),
    "lw_debug_ring_policy": (None, [C.c_int]),
    "lw_decoder_device_cu_count": (C.c_int, [C.c_void_p]),
    "lw_sharder_shard_cus": (C.c_int, [C.c_void_p, C.c_size_t]),
    "lw_decoder_max_block_elems": (C.c_size_t, [C.c_void_p]),
    "lw_ogg_reader_open_memory": (C.c_void_p, [C.c_char_p, C.c_size_t, C.c_int]),
    "lw_ogg_reader_open_file": (C.c_void_p, [C.c_char_p, intp]),
    "lw_ogg_reader_open_io": (C.c_void_p, [C.POINTER(OggIo)]),
    "lw_ogg_reader_close": (None, [C.c_void_p]),
    "lw_ogg_read_packet": (C.c_int, [C.c_void_p, C.POINTER(OggPacket)]),
    "lw_ogg_read_packet_expected": (C.c_int, [C.c_void_p, C.POINTER(OggPacket)]),
    "lw_ogg_delete_unread_packets": (None, [C.c_void_p]),
    "lw_ogg_seek_absgp": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_uint64]),
    "lw_ogg_crc32": (C.c_uint32, [C.c_char_p, C.c_size_t, C.c_uint32]),
    "lw_ogg_stream_open": (C.c_void_p, [C.c_void_p, C.c_int, intp]),
    "lw_ogg_stream_close": (None, [C.c_void_p]),
    "lw_ogg_stream_into_inner": (C.c_void_p, [C.c_void_p]),
    "lw_ogg_stream_ident": (C.c_void_p, [C.c_void_p]),
    "lw_ogg_stream_comment": (C.c_void_p, [C.c_void_p]),
    "lw_ogg_stream_setup": (C.c_void_p, [C.c_void_p]),
    "lw_ogg_stream_serial": (C.c_uint32, [C.c_void_p]),
    "lw_ogg_stream_link_index": (C.c_uint32, [C.c_void_p]),
    "lw_ogg_stream_last_absgp": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "lw_ogg_stream_read_dec_packet": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, szp]),
    "lw_ogg_stream_set_entropy_on_device": (C.c_int, [C.c_void_p, C.c_int]),
    "lw_ogg_stream_set_read_ahead": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int]),
    "lw_ogg_stream_read_dec_packets": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t,
                                                 C.POINTER(C.c_uint32), C.POINTER(C.c_int32), szp]),
    "lw_ogg_stream_skip_samples_linear": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, szp, szp,
                                                    intp]),
    "lw_ogg_stream_seek_absgp_pg": (C.c_int, [C.c_void_p, C.c_uint64]),
}

for _n, (_r, _a) in SYMBOLS.items():
    globals()[_n] = _sig(_n, _r, _a)

OK = 0
AUDIO_END_OF_PACKET, AUDIO_BAD_FORMAT, AUDIO_IS_HEADER, AUDIO_BUFFER_NOT_ADDRESSABLE = 1, 2, 3, 4
HDR_END_OF_PACKET, HDR_NOT_VORBIS, HDR_UNSUPPORTED_VERSION, HDR_BAD_FORMAT = 16, 17, 18, 19
HDR_BAD_TYPE, HDR_IS_AUDIO, HDR_UTF8, HDR_BUFFER_NOT_ADDRESSABLE = 20, 21, 22, 23
ERR_NULL_ARG, ERR_DEVICE, ERR_CAPACITY, ERR_STATE_MISMATCH, ERR_UNSUPPORTED = 32, 33, 34, 35, 36
OGG_EOF, OGG_NO_CAPTURE_PATTERN, OGG_INVALID_STREAM_STRUCT_VER, OGG_HASH_MISMATCH = 48, 49, 50, 51
OGG_READ_ERROR, OGG_INVALID_DATA = 52, 53
FMT_I16_PLANAR, FMT_I16_INTERLEAVED, FMT_F32_PLANAR = 0, 1, 2
TAP_RESIDUE_PRE_INVERSE, TAP_RESIDUE_POST_INVERSE, TAP_PRE_MDCT, TAP_POST_MDCT = 0, 1, 2, 3


def device_error():
    return (lw_last_device_error() or b"").decode()  # noqa: F821
