/*
 * lewton_amd.h -- C ABI of the MI355X-native Vorbis audio-packet decode path.
 *
 * This is the drop-in boundary for lewton's `audio` module (src/audio.rs + src/imdct.rs +
 * src/samples.rs, fed by src/header.rs): a Rust/C/Python host binds exactly these symbols
 * (see INTEGRATION.md for the `extern "C"` block a lewton maintainer would add).  Plain pointers
 * and sizes only; no C++/torch types.  All functions are `noexcept` in effect: errors are status
 * codes, nothing unwinds across the boundary.
 *
 * Split of work: the bit-serial entropy stage (packet prologue, floor decode,
 * residue Huffman/VQ decode; audio.rs:921-986) runs on the host inside this library; everything from
 * the residue vectors onward (inverse coupling, floor-1 curve, floor x residue, IMDCT,
 * window/overlap-add, sample conversion; audio.rs:990-1157, imdct.rs:291-659, samples.rs:32-103)
 * runs in hand-written HIP kernels for gfx950.  There is no CPU fallback for the device stage:
 * calls fail with LW_ERR_DEVICE when no GPU is usable.
 *
 * Paths cited below are relative to the reference tree (RustAudio/lewton 0.10.2).
 */
#ifndef LEWTON_AMD_H
#define LEWTON_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------------------- */
enum {
	LW_OK = 0,
	/* AudioReadError, src/audio.rs:26-41 */
	LW_AUDIO_END_OF_PACKET = 1,
	LW_AUDIO_BAD_FORMAT = 2,
	LW_AUDIO_IS_HEADER = 3,
	LW_AUDIO_BUFFER_NOT_ADDRESSABLE = 4,
	/* HeaderReadError, src/header.rs:35-63 */
	LW_HDR_END_OF_PACKET = 16,
	LW_HDR_NOT_VORBIS = 17,
	LW_HDR_UNSUPPORTED_VERSION = 18,
	LW_HDR_BAD_FORMAT = 19,
	LW_HDR_BAD_TYPE = 20,
	LW_HDR_IS_AUDIO = 21,
	LW_HDR_UTF8 = 22,
	LW_HDR_BUFFER_NOT_ADDRESSABLE = 23,
	/* errors of this library (no counterpart in the reference) */
	LW_ERR_NULL_ARG = 32,      /* like capi.rs:106-108 returning 1 */
	LW_ERR_DEVICE = 33,        /* HIP error / no GPU; lw_last_device_error() has the text */
	LW_ERR_CAPACITY = 34,      /* caller-provided buffer or batch too small */
	LW_ERR_STATE_MISMATCH = 35 /* pwr belongs to another decoder (the reference panics, audio.rs:1086) */
};

/* Output sample formats = the `Samples` implementations of src/samples.rs */
enum {
	LW_FMT_I16_PLANAR = 0,      /* Vec<Vec<i16>>: per packet [ch][m]          (samples.rs:20-40, :92-103) */
	LW_FMT_I16_INTERLEAVED = 1, /* InterleavedSamples<i16>: per packet [m][ch] (samples.rs:48-78) */
	LW_FMT_F32_PLANAR = 2       /* Vec<Vec<f32>>: per packet [ch][m]          (samples.rs:86-90; capi.rs:110) */
};

typedef struct lw_ident lw_ident;     /* IdentHeader incl. cached_bs_derived, src/header.rs:188-211 */
typedef struct lw_setup lw_setup;     /* SetupHeader, src/header.rs:471-481 */
typedef struct lw_comment lw_comment; /* CommentHeader, src/header.rs:289-300 */
typedef struct lw_decoder lw_decoder; /* device context: tables of one (ident, setup) pair on one GPU */
typedef struct lw_pwr lw_pwr;         /* PreviousWindowRight, src/audio.rs:847-861 (device resident) */
typedef struct lw_batch lw_batch;     /* pinned staging + device buffers for a batch of packets */

/* ---- headers (host) ----------------------------------------------------------------------- */
typedef struct {
	uint8_t audio_channels;
	uint32_t audio_sample_rate;
	int32_t bitrate_maximum, bitrate_nominal, bitrate_minimum;
	uint8_t blocksize_0, blocksize_1;
} lw_ident_info;

/* read_header_ident, src/header.rs:221-259.  NULL + *err on failure. */
lw_ident *lw_read_header_ident(const uint8_t *packet, size_t len, int *err);
int lw_ident_get_info(const lw_ident *id, lw_ident_info *out);
void lw_ident_free(lw_ident *id);
/* read_header_setup, src/header.rs:1082-1154 */
lw_setup *lw_read_header_setup(const uint8_t *packet, size_t len, uint8_t audio_channels,
		uint8_t blocksize_0, uint8_t blocksize_1, int *err);
void lw_setup_free(lw_setup *s);
/* read_header_comment, src/header.rs:309-355 */
lw_comment *lw_read_header_comment(const uint8_t *packet, size_t len, int *err);
const char *lw_comment_vendor(const lw_comment *c, size_t *len);
size_t lw_comment_count(const lw_comment *c);
int lw_comment_get(const lw_comment *c, size_t i, const char **key, size_t *key_len, const char **val, size_t *val_len);
void lw_comment_free(lw_comment *c);

/* ---- device context ----------------------------------------------------------------------- */
int lw_device_count(void);
/* Uploads the per-blocksize tables (header_cached.rs:34-110, computed on the host with libm exactly
 * like the reference), floor-1 post tables, the inverse-dB table and the mapping/coupling lists. */
lw_decoder *lw_decoder_create(const lw_ident *id, const lw_setup *s, int device, int *err);
void lw_decoder_destroy(lw_decoder *d);
const char *lw_last_device_error(void);

/* ---- PreviousWindowRight (src/audio.rs:847-861) ------------------------------------------- */
lw_pwr *lw_pwr_new(lw_decoder *d);         /* PreviousWindowRight::new */
int lw_pwr_is_empty(const lw_pwr *p);      /* ::is_empty */
lw_pwr *lw_pwr_clone(const lw_pwr *p);     /* #[derive(Clone)] (device copy) */
void lw_pwr_reset(lw_pwr *p);              /* `pwr = PreviousWindowRight::new()`, inside_ogg.rs:307-313 */
void lw_pwr_free(lw_pwr *p);
size_t lw_pwr_len(const lw_pwr *p);        /* per-channel length of the stored right part */
int lw_pwr_copy_to_host(const lw_pwr *p, float *dst /* [ch][len] */);

/* ---- one packet (drop-in for audio.rs) ---------------------------------------------------- */
/* get_decoded_sample_count, src/audio.rs:874-909 (header bits only, host only) */
int lw_get_decoded_sample_count(const lw_ident *id, const lw_setup *s, const uint8_t *packet, size_t len,
		size_t *count);
/* read_audio_packet_generic<S>, src/audio.rs:919-1160 (read_audio_packet :1170 is fmt = LW_FMT_I16_PLANAR).
 * `out` is host memory for ch * cap samples; *n_samples = per-channel sample count (0 for the first
 * packet after a reset, audio.rs:1140-1152).  Synchronous: entropy decode, H2D, kernels, D2H. */
int lw_read_audio_packet(lw_decoder *d, const uint8_t *packet, size_t len, lw_pwr *pwr, int fmt,
		void *out, size_t cap_per_channel, size_t *n_samples);

/* ---- host entropy stage on its own (no GPU needed) ---------------------------------------- */
/* The bit-serial half of read_audio_packet_generic (audio.rs:921-986 + floor-1 amplitude unwrap
 * :391-435): decodes one packet into the GPU-stage record.  floor_out: [ch][lw_setup_floor_stride()] u16
 * (ascending-x order; bits 0-7 = final_y*multiplier, bit 15 = active, entry 0 == 0xFFFF = unused floor);
 * residue_out: [ch][n/2] f32 before inverse coupling.  *blocksize_log2, *mode and *flags (bit 0 long, bit 1
 * prev window flag, bit 2 next window flag) describe the packet; *bits_consumed = position of the bit
 * cursor afterwards. */
uint32_t lw_setup_floor_stride(const lw_setup *s);
int lw_entropy_decode_host(const lw_ident *id, const lw_setup *s, const uint8_t *packet, size_t len,
		uint16_t *floor_out, float *residue_out, size_t residue_cap_floats, uint8_t *blocksize_log2, uint8_t *mode,
		uint8_t *flags, uint64_t *bits_consumed);

/* ---- batches ------------------------------------------------------------------------------ */
typedef struct {
	const uint8_t *data;
	size_t len;
	lw_pwr *pwr; /* packets sharing a pwr must be listed in stream order */
} lw_packet;

typedef struct {
	int32_t status;      /* LW_OK or an AudioReadError code; failed packets produce no samples */
	uint32_t n_samples;  /* per channel */
	uint64_t out_offset; /* element offset of this packet's block in the output buffer */
} lw_packet_result;

/* Tap points = the reference's record_*! macros (src/lib.rs:56-94; audio.rs:988,1004,1041,1054) */
enum { LW_TAP_RESIDUE_PRE_INVERSE = 0, LW_TAP_RESIDUE_POST_INVERSE = 1, LW_TAP_PRE_MDCT = 2, LW_TAP_POST_MDCT = 3 };

lw_batch *lw_batch_create(lw_decoder *d, size_t max_packets, int fmt, int *err);
void lw_batch_destroy(lw_batch *b);
/* Host entropy stage for `n` packets on `n_threads` host threads (0 = hardware concurrency): fills the
 * pinned staging buffers with GPU-stage records and decides sample counts, window geometry and error
 * statuses (all host-decidable).  Advances the host-side bookkeeping of every pwr. */
int lw_batch_entropy(lw_batch *b, const lw_packet *pkts, size_t n, int n_threads);
/* hipMemcpyAsync of the staged records to the device (stream = hipStream_t, NULL = default stream) */
int lw_batch_upload(lw_batch *b, void *hip_stream);
/* Launch the synthesis kernels on `hip_stream`; d_out = DEVICE pointer to lw_batch_out_elems() elements
 * of the batch's format.  Asynchronous.  Re-launching the same uploaded batch is idempotent. */
int lw_batch_synth(lw_batch *b, void *d_out, size_t out_capacity_elems, void *hip_stream);
/* Convenience: synth into an internal device buffer, copy to host memory, synchronise. */
int lw_batch_synth_to_host(lw_batch *b, void *h_out, size_t out_capacity_elems, void *hip_stream);
size_t lw_batch_size(const lw_batch *b);
size_t lw_batch_out_elems(const lw_batch *b);
const lw_packet_result *lw_batch_results(const lw_batch *b);
/* bytes the device stage reads+writes for this batch by the SURVEY 8(d) definition */
uint64_t lw_batch_algorithmic_bytes(const lw_batch *b);
/* Debug taps: copy one intermediate of packet `idx` to host after running the generic kernels.
 * dst: [ch][n/2] floats ([ch][n] for LW_TAP_POST_MDCT). */
int lw_batch_tap(lw_batch *b, size_t idx, int tap, float *dst, size_t cap_floats);
/* Force the generic (any block size / window shape) kernels even where a specialised one applies. */
void lw_batch_set_force_generic(lw_batch *b, int on);
/* names of the kernels the last lw_batch_synth used, comma separated (introspection for tests/bench) */
const char *lw_batch_last_kernels(const lw_batch *b);

/* Test hook for the host Huffman decoder (spec 3.2.1 codeword assignment; src/huffman_tree.rs:183-221):
 * returns 0 valid, 1 overspecified, 2 underpopulated, 3 invalid single entry; when valid and bits != NULL
 * decodes up to max_syms symbols. */
int lw_huffman_check(const uint8_t *lengths, size_t n_entries, const uint8_t *bits, size_t bits_len,
		uint32_t *syms, size_t max_syms, size_t *n_syms);

/* Test hook: run the device IMDCT (src/imdct.rs:291) of block size blocksize_0 (blockflag 0) or blocksize_1
 * (blockflag 1) on `spectrum` (n/2 floats) and return the n time-domain values.  Implemented as a packet
 * record with a unit floor (inverse-dB index 255 = 1.0) through the generic kernels. */
int lw_debug_imdct(lw_decoder *d, int blockflag, const float *spectrum, float *out);

/* Test hook (host only): the LDS table image of the specialised long-block kernel and its 16 section
 * offsets (order of struct LwFastImage in csrc/lw_fast.hpp).  Returns the image size in bytes, 0 if the
 * stream shape is not covered by that kernel; copies min(size, cap) bytes. */
size_t lw_debug_fast_image(const lw_ident *id, const lw_setup *s, uint8_t *dst, size_t cap, uint32_t *offsets16);

/* Library/version introspection */
const char *lw_version(void);

#ifdef __cplusplus
}
#endif
#endif
