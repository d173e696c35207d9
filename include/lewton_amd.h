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
	LW_ERR_NULL_ARG = 32,      /* like capi.rs:106-108 returning 1; a packet given as (NULL, 0) is not an error but an empty packet */
	LW_ERR_DEVICE = 33,        /* HIP error / no GPU; lw_last_device_error() has the text */
	LW_ERR_CAPACITY = 34,      /* caller-provided buffer or batch too small */
	LW_ERR_STATE_MISMATCH = 35, /* pwr belongs to another decoder (the reference panics, audio.rs:1086) */
	LW_ERR_UNSUPPORTED = 36     /* an optional mode is not available for this stream (see the function's comment) */
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
 * (ascending-x order; bits 0-7 = final_y*multiplier, bit 15 = active, entry 0 == 0xFFFF = unused floor, entry 0 ==
 * 0xFFFE = floor 0: the curve of audio.rs:160-212 is in floor_curve_out [ch][n/2] f32, which may be NULL for setups
 * without a floor of type 0); residue_out: [ch][n/2] f32 before inverse coupling.  *blocksize_log2, *mode and *flags (bit 0 long, bit 1
 * prev window flag, bit 2 next window flag) describe the packet; *bits_consumed = position of the bit
 * cursor afterwards. */
uint32_t lw_setup_floor_stride(const lw_setup *s);
int lw_entropy_decode_host(const lw_ident *id, const lw_setup *s, const uint8_t *packet, size_t len,
		uint16_t *floor_out, float *residue_out, size_t residue_cap_floats, uint8_t *blocksize_log2, uint8_t *mode,
		uint8_t *flags, uint64_t *bits_consumed, float *floor_curve_out);

/* Introspection for tests and tools: the dense VQ table of a codebook (header.rs:495-531; entries x dims floats,
 * dst may be NULL to query the sizes; returns LW_ERR_UNSUPPORTED for a book without a lookup table). */
int lw_setup_codebook_vq(const lw_setup *s, unsigned book, float *dst, size_t cap_floats, uint32_t *dims, uint32_t *entries);

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
/* Host entropy stage for `n` packets on `n_threads` host threads (0 = lw_default_host_threads()): fills the
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
/* bytes of PreviousWindowRight that have to cross HBM because of the launch boundary, NOT part of the 8(d) figure: the stored
 * right part of every stream whose first packet of the batch has one (read from the state pool) and the right part every
 * stream's last packet leaves behind (written to it), channels * length * 4 each.  Inside a launch the state passes through LDS. */
uint64_t lw_batch_state_bytes(const lw_batch *b);
/* Debug taps: copy one intermediate of packet `idx` to host after running the generic kernels.
 * dst: [ch][n/2] floats ([ch][n] for LW_TAP_POST_MDCT). */
int lw_batch_tap(lw_batch *b, size_t idx, int tap, float *dst, size_t cap_floats);
/* Force the generic (any block size / window shape) kernels even where a specialised one applies. */
void lw_batch_set_force_generic(lw_batch *b, int on);
/* Test hook: rounds per workgroup of the specialised long-block kernel for this batch's next lw_batch_entropy (1..16;
 * 0 = the planner decides).  Exercises the hand-over of window state across rounds and workgroups on small batches. */
void lw_debug_batch_set_rounds(lw_batch *b, int rounds);
/* Test hook: how the wave-pipeline kernels get the right half of a predecessor that is not the item in front (a chunk of the work
 * list that starts inside a stream): -1 (default) = the predecessor is recomputed inside the launch as an item of its own where
 * that pays, 0 = always by the pre-pass launch (k_long<halo>) */
void lw_debug_batch_set_halo(lw_batch *b, int mode);
/* test hook: 0 = never run a mixed short / long batch as one k_mix launch (two launches: k_long<EDGE>, k_short), -1 = where it applies */
void lw_debug_batch_set_mix(lw_batch *b, int mode);
/* test hook for blocksize_1 = 10 / 12 streams: -1 = k_long10 / k_long12 (long blocks next to short ones in their EDGE form where the
 * short blocks run through k_short<8 / 16>), 1 = k_long10 / k_long12 for the long blocks with two long slopes only (the others
 * through the generic kernels), 0 = the block kernels k_short<32> / k_big<12> instead */
void lw_debug_batch_set_long10(lw_batch *b, int mode);
/* Device-side failures of a batch's launches (audio.rs:27-41: every failure is a status, never wrong samples).  Call once the
 * work lw_batch_synth queued has COMPLETED (after synchronising its stream); lw_batch_synth_to_host and lw_ring_collect call it
 * themselves.  LW_OK, or LW_ERR_DEVICE: a kernel raised the batch's error word (k_mix: a short block's wave never saw the raw
 * edges its long neighbours' waves of the same launch were to write -- the grid was not resident at once, e.g. under a CU mask or
 * next to another process's kernels).  Then every packet's result carries LW_ERR_DEVICE with 0 samples, the PCM written by this
 * batch must not be used, and the window state of the batch's streams is undefined: reset their PreviousWindowRight (or restore
 * a snapshot, lw_pwr_snapshot) before decoding on.  The word and the launch's leftovers are cleared: the batch can run again. */
int lw_batch_device_status(lw_batch *b);
/* test hook: spin != 0 = the long blocks' waves of the next k_mix launches never signal their edges, and the short blocks' waves
 * give up after `spin` polls (lw_batch_device_status then reports LW_ERR_DEVICE); 0 = normal operation */
void lw_debug_batch_break_mix(lw_batch *b, unsigned spin);
/* ... and process-wide, for the batches a ring, a sharder or an Ogg stream reader owns: every batch launched while spin != 0 */
void lw_debug_break_mix(unsigned spin);
/* Entropy stage on the device (csrc/lw_dev_entropy.h, k_entropy): the bit-serial half of read_audio_packet_generic
 * (audio.rs:921-986: floor-1 decode :215-251 + amplitude unwrap :391-435, residue decode :587-760) runs on the GPU, one
 * wave per packet; lw_batch_entropy then only reads the prologues, copies the packets into pinned staging and plans the
 * batch, and the packets themselves (~0.5 KB instead of 8.3 KB of records per stereo long block) cross PCIe.  Records,
 * PCM and statuses are bit-identical to the host stage's.  Eligible streams: floor type 1, residue books with a vector
 * lookup of at most 64 dimensions, at most 16 channels and 16 coupling steps (`why` names the reason otherwise;
 * LW_ERR_UNSUPPORTED from the setters). */
int lw_decoder_supports_device_entropy(const lw_decoder *d, const char **why);
int lw_batch_set_entropy_on_device(lw_batch *b, int on);
/* Runs k_entropy for the records uploaded last (after lw_batch_upload, on the same stream), ahead of lw_batch_synth: the
 * entropy kernel touches no stream state, so a pipeline may queue it before it orders the synthesis kernels behind the
 * previous batch's (the staging ring does).  lw_batch_synth runs it itself when this was not called.  No-op in host mode. */
int lw_batch_device_entropy(lw_batch *b, void *hip_stream);
/* names of the kernels the last lw_batch_synth used, comma separated (introspection for tests/bench) */
const char *lw_batch_last_kernels(const lw_batch *b);

/* ---- staging ring (BASELINE north_star: "pinned hipMemcpyAsync staging ring so entropy decode of packet N+1 overlaps
 * GPU synthesis of packet N") ------------------------------------------------------------------------------------------
 * A ring of `slots` staging slots on the decoder's device; a slot = one batch object (pinned records + device mirror), a device
 * PCM buffer, a pinned host PCM buffer, a HIP stream.  Slots are used first-in first-out:
 *   lw_ring_stage    host entropy stage of `n` packets into the next free slot (what lw_batch_entropy does);
 *                    LW_ERR_CAPACITY when every slot is in flight (collect + release first) or n > max_packets
 *   lw_ring_launch   queues, for the oldest staged slot and without waiting: H2D of its records, the synthesis kernels
 *                    (ordered behind the previous launch's kernels: consecutive batches may carry the same streams'
 *                    window state) and D2H of the PCM into the slot's pinned buffer
 *   lw_ring_submit   = stage + launch: returns while the GPU works, so the next submit's entropy decode overlaps it
 *   lw_ring_collect  waits for the oldest launched slot; results / PCM stay valid until lw_ring_release.  LW_ERR_DEVICE with
 *                    the slot collected all the same (release it as usual): lw_batch_device_status of that batch
 *   lw_ring_release  frees that slot
 *   lw_ring_drain    waits for everything in flight and frees all slots (their results are dropped)
 * stage may run on another thread than launch / collect / release (one thread each).  Every PreviousWindowRight sees its
 * packets in submission order.  The reference has no counterpart (it decodes one packet per call, audio.rs:919); callers
 * that want its call-by-call semantics use lw_read_audio_packet or the Ogg stream layer below, which runs on this ring. */
typedef struct lw_ring lw_ring;
lw_ring *lw_ring_create(lw_decoder *d, size_t slots, size_t max_packets, int fmt, int *err);
void lw_ring_destroy(lw_ring *r);
int lw_ring_stage(lw_ring *r, const lw_packet *pkts, size_t n, int n_threads);
int lw_ring_launch(lw_ring *r);
int lw_ring_submit(lw_ring *r, const lw_packet *pkts, size_t n, int n_threads);
int lw_ring_collect(lw_ring *r, const lw_packet_result **results, size_t *n, const void **pcm, size_t *pcm_elems);
int lw_ring_release(lw_ring *r);
int lw_ring_drain(lw_ring *r);
size_t lw_ring_slots(const lw_ring *r);
size_t lw_ring_in_flight(lw_ring *r); /* slots staged, launched or collected and not yet released */
size_t lw_ring_last_staged_elems(lw_ring *r); /* elements the batch staged last will produce (known after lw_ring_stage) */
int lw_ring_set_entropy_on_device(lw_ring *r, int on); /* lw_batch_set_entropy_on_device for every slot (ring must be idle) */
const char *lw_ring_last_kernels(const lw_ring *r);
/* measurement hook, process-wide, read by lw_ring_create: -1 (default) = by the ring's decoder -- a tenant's ring
 * (lw_decoder_set_cu_share) runs its launches' kernels one launch after the other, and its PCM copies are issued by one copier
 * thread per device once their kernels have finished (a copy queued behind its kernels blocks the copy engine for the other
 * tenants' ready copies); else bit 0 = kernels in launch order per ring, bit 1 = copies by the device's copier */
void lw_debug_ring_policy(int bits);
/* The host half of a PreviousWindowRight (whether a right part is stored, its length, which of the two device buffers
 * holds it).  lw_batch_entropy / lw_ring_stage advance it when they plan a batch; a caller that drops a staged batch (or
 * one launched batch: a launch writes the OTHER device buffer) restores the state it saved before staging. */
typedef struct {
	uint8_t present, parity;
	uint32_t len;
} lw_pwr_state;
void lw_pwr_get_state(const lw_pwr *p, lw_pwr_state *out);
void lw_pwr_set_state(lw_pwr *p, const lw_pwr_state *in);
int lw_decoder_device(const lw_decoder *d);
/* Several decoders on ONE GPU (tenants; nothing in the reference corresponds: lewton decodes one stream on one thread).
 * lw_decoder_set_shared_device(d, 1): other decoders' rings run on d's GPU as well.  The rings created for d AFTERWARDS then hand
 * their PCM copies to one copier thread per device, which issues each copy once its kernels have finished -- a copy queued
 * behind its kernels blocks the copy engine for every other ring's ready copies (measured: two rings on one GPU 8.6-9.0 M
 * packets/s, with the copier 11.1-12.4 M, the rate of one ring) -- and run their launches' kernels in launch order.
 * lw_sharder_create does this by itself when devices[] names a device several times.
 * lw_decoder_set_cu_share: in addition, decoder `part` of `parts` launches on the slice [32 part / parts, 32 (part + 1) / parts)
 * of the 32 CUs of EVERY XCD only (a HIP queue has to keep CUs on all eight; tenants share the L2s, never a CU): the rings
 * created AFTERWARDS launch on CU-masked HIP streams and its batches are planned for that many CUs; parts = 1 gives the device
 * back.  The wave-pipeline kernels take whole compute units, so without a share a 15 us launch next to another tenant's
 * long-running kernel waits for CUs to drain (130-700 us measured; 28 us flat on a half-device share).  Throughput is the
 * same either way (the PCIe link is the bound), so the sharder leaves it off.  Batches launched on a caller's own stream
 * (lw_batch_synth) are planned for the share but run wherever that stream runs.  LW_ERR_UNSUPPORTED: parts > 32.
 * LW_ERR_UNSUPPORTED also: a device that is not the one the mask layout was measured on (gfx950, eight XCDs of equally many CUs).
 * The two kinds of tenant stream never meet in one process.  With the HIP runtime this was measured on (ROCm 7.2) a process that
 * had BOTH copied on the copier's own stream (the ring of a flagged decoder without a share) AND run CU-masked streams was seen
 * not to exit, one run in three.  The library therefore keeps a latch per device and process:
 *   - once a ring has made the copier's own stream on a device, lw_decoder_set_cu_share(parts > 1) on that device and
 *     lw_ring_create for a decoder that already has a share return LW_ERR_UNSUPPORTED;
 *   - once a CU-masked stream exists on a device, the copier of later rings issues their copies on the slots' own streams
 *     instead of a stream of its own (the same results; 9-10 instead of 11-12 M packets/s for two tenants).
 * A deployment picks one of the two modes per process: CU shares for tenants, or none. */
int lw_decoder_set_shared_device(lw_decoder *d, int on);
int lw_decoder_set_cu_share(lw_decoder *d, unsigned part, unsigned parts);
int lw_decoder_cu_count(const lw_decoder *d); /* compute units this decoder's launches are planned for */
int lw_decoder_device_cu_count(const lw_decoder *d); /* compute units of its device */
size_t lw_decoder_max_block_elems(const lw_decoder *d); /* channels * (3 n1 - n0) / 4: the largest block a packet yields */

/* ---- independent streams sharded over the GPUs of a node, one process (BASELINE configs[4]) ----------------
 * Streams never exchange data (audio.rs:919 touches only its own pwr): shard g owns the streams with stream_id mod G == g.
 * A shard = one lw_decoder on devices[g] (tables + the state pool of its streams in that GPU's HBM), one batch with pinned
 * staging, one HIP stream, one worker thread.  devices[] may name a device several times (logical shards).
 * A shard runs on a staging ring of its own (lw_ring_*: pinned records and pinned PCM per slot).
 * lw_sharder_submit takes packets of any streams (those of one stream in stream order), has every shard run the host
 * entropy stage of its packets and queue H2D, kernels and D2H on the shard's own thread and device, all shards at once,
 * and returns when every shard has launched -- while the GPUs work (no collective, nothing crosses xGMI).  The packet
 * bytes are consumed when it returns.  *out_elems = elements the call will produce.  Up to 3 calls may be in flight, so
 * the host stage of call k+1 overlaps the GPU work of call k on every device.
 * lw_sharder_collect waits for the OLDEST call: out = host memory for cap_elems elements (NULL: drop the samples); results[i] (status, n_samples,
 * out_offset = element offset of packet i's block in out) come back in the order of that call's pkts; blocks are laid out
 * shard by shard.  LW_ERR_CAPACITY: more than max_packets_per_shard packets for one shard, three calls already in flight
 * (submit), nothing in flight or out too small (collect).
 * Which returns of lw_sharder_collect consume the call: LW_ERR_CAPACITY and LW_ERR_NULL_ARG come from the argument checks and
 * consume NOTHING (call again); every other return -- LW_OK, or the error of a shard (LW_ERR_DEVICE) -- has taken the call out
 * of the queue and freed its slots; the packets of a shard that failed carry LW_ERR_DEVICE in results[].  A shard whose ring
 * saw a device error is started over (drained): EVERY part that was in flight on that shard at that moment -- of older calls and
 * of newer ones already submitted (up to two) -- reports LW_ERR_DEVICE when it is collected; calls submitted after the drain run
 * normally.  The dropped batches had already advanced the host halves of their streams' window states, so the drain resets the
 * PreviousWindowRight of every stream opened on that shard (lw_sharder_stream_reset): the first packet of each of them after the
 * failure yields 0 samples (audio.rs:1140-1152), never samples overlapped with a stale right part.
 * A submit that fails AFTER some shards have launched (a device error on one shard) still queues the call, so that the
 * slots those shards hold can be freed: collect it (its packets on the failed shard come back with LW_ERR_DEVICE).
 * lw_sharder_decode = submit + collect on an empty pipeline.
 * Logical shards (a device named several times) are tenants of that GPU (lw_decoder_set_shared_device): their PCM copies go
 * through the device's copier thread, ready copies only, one at a time.
 * The process-per-GPU form of the same rule is lewton_amd/shard.py + bench.py under torch.distributed.run. */
typedef struct lw_sharder lw_sharder;
typedef struct lw_shard_stream lw_shard_stream; /* one logical stream: its PreviousWindowRight lives on the owning shard */
typedef struct {
	lw_shard_stream *stream;
	const uint8_t *data;
	size_t len;
} lw_shard_packet;
lw_sharder *lw_sharder_create(const lw_ident *id, const lw_setup *setup, const int *devices, size_t n_shards,
		size_t max_packets_per_shard, int fmt, int *err);
void lw_sharder_destroy(lw_sharder *sh); /* close the streams first */
size_t lw_sharder_shards(const lw_sharder *sh);
int lw_sharder_shard_cus(const lw_sharder *sh, size_t shard); /* compute units shard's launches run on (its lw_decoder_cu_count) */
/* measurement hook, process-wide, read by lw_sharder_create: 1 = logical shards of one device also get a CU share each
 * (lw_decoder_set_cu_share); 0 (default) = they share all its CUs */
void lw_debug_sharder_share_cus(int on);
int lw_sharder_set_entropy_on_device(lw_sharder *sh, int on); /* every shard's entropy stage on its own GPU (k_entropy) */
size_t lw_sharder_shard_of(const lw_sharder *sh, uint64_t stream_id);
int lw_sharder_device_of(const lw_sharder *sh, size_t shard);
lw_shard_stream *lw_sharder_stream_open(lw_sharder *sh, uint64_t stream_id);
void lw_sharder_stream_close(lw_shard_stream *st);
void lw_sharder_stream_reset(lw_shard_stream *st); /* `pwr = PreviousWindowRight::new()` */
int lw_sharder_decode(lw_sharder *sh, const lw_shard_packet *pkts, size_t n, int n_threads_per_shard, void *out,
		size_t cap_elems, lw_packet_result *results);
int lw_sharder_submit(lw_sharder *sh, const lw_shard_packet *pkts, size_t n, int n_threads_per_shard, size_t *out_elems);
int lw_sharder_collect(lw_sharder *sh, void *out, size_t cap_elems, lw_packet_result *results, size_t n_results);
size_t lw_sharder_in_flight(lw_sharder *sh); /* calls submitted and not yet collected */
/* Zero-copy form of collect: the oldest call's PCM stays where the GPUs' copy engines put it, in the shards' pinned ring
 * buffers -- pcm[g] / elems[g] for shard g (arrays of lw_sharder_shards() entries); results[i].out_offset is relative to
 * the block of the shard that owns packet i (lw_sharder_shard_of).  Valid until lw_sharder_release, which frees the slots.
 * Both run on the caller's thread (no hand-over to the shards' workers) and leave the caller's current HIP device as they found
 * it.  After ANY return of lw_sharder_collect_pinned other
 * than LW_ERR_CAPACITY / LW_ERR_NULL_ARG the call is held by the caller: lw_sharder_release takes it out of the queue. */
int lw_sharder_collect_pinned(lw_sharder *sh, lw_packet_result *results, size_t n_results, const void **pcm, size_t *elems);
int lw_sharder_release(lw_sharder *sh);

/* ---- Ogg container either side of the path ---------------------------- */
/* lewton reads Ogg through the external crate `ogg` 0.8.0 (Cargo.lock; `PacketReader`, `Packet`) and wraps it
 * in src/inside_ogg.rs.  The functions below replace both: a page/packet demultiplexer after RFC 3533 with the
 * packet attributes lewton's call sites use, and `OggStreamReader` with the device decode behind it. */
enum {
	LW_OGG_EOF = 48,                     /* Ok(None): clean end of the physical stream */
	/* ogg::OggReadError */
	LW_OGG_NO_CAPTURE_PATTERN = 49,      /* NoCapturePatternFound */
	LW_OGG_INVALID_STREAM_STRUCT_VER = 50, /* InvalidStreamStructVer(u8) */
	LW_OGG_HASH_MISMATCH = 51,           /* HashMismatch(u32, u32) */
	LW_OGG_READ_ERROR = 52,              /* ReadError(io::Error), incl. UnexpectedEof inside a page / read_packet_expected */
	LW_OGG_INVALID_DATA = 53             /* InvalidData */
};

typedef struct lw_ogg_reader lw_ogg_reader; /* ogg::PacketReader<T: Read + Seek> */
typedef struct lw_ogg_stream lw_ogg_stream; /* OggStreamReader<T>, src/inside_ogg.rs:66-77 */

typedef struct {                 /* the `T: Read + Seek` of the reference as C callbacks */
	int64_t (*read)(void *user, uint8_t *dst, size_t n);       /* bytes read (0 = end), < 0 = error */
	int64_t (*seek)(void *user, int64_t offset, int whence);   /* whence 0/1/2 = SEEK_SET/CUR/END; new position or < 0 */
	void *user;
} lw_ogg_io;

typedef struct {                 /* ogg::Packet */
	const uint8_t *data;         /* owned by the reader, valid until its next call */
	size_t len;
	uint32_t stream_serial;      /* Packet::stream_serial() */
	uint64_t absgp_page;         /* Packet::absgp_page(): granule position of the page the packet ended on */
	uint8_t first_in_stream;     /* Packet::first_in_stream(): first packet of a page with the begin-of-stream flag */
	uint8_t last_in_stream;      /* Packet::last_in_stream(): last packet of a page with the end-of-stream flag */
	uint8_t first_in_page;       /* Packet::first_in_page() */
	uint8_t last_in_page;        /* Packet::last_in_page() */
} lw_ogg_packet;

/* PacketReader::new(rdr).  open_memory borrows `data` unless copy != 0. */
lw_ogg_reader *lw_ogg_reader_open_memory(const uint8_t *data, size_t len, int copy);
lw_ogg_reader *lw_ogg_reader_open_file(const char *path, int *err);
lw_ogg_reader *lw_ogg_reader_open_io(const lw_ogg_io *io);
void lw_ogg_reader_close(lw_ogg_reader *r);
/* PacketReader::read_packet: LW_OK + *out, LW_OGG_EOF, or an OggReadError code.  Streams may be multiplexed. */
int lw_ogg_read_packet(lw_ogg_reader *r, lw_ogg_packet *out);
/* PacketReader::read_packet_expected: end of stream is LW_OGG_READ_ERROR (UnexpectedEof) */
int lw_ogg_read_packet_expected(lw_ogg_reader *r, lw_ogg_packet *out);
/* PacketReader::delete_unread_packets (inside_ogg.rs:47) */
void lw_ogg_delete_unread_packets(lw_ogg_reader *r);
/* PacketReader::seek_absgp(stream_serial, absgp) with page granularity (inside_ogg.rs:307-313): reading resumes behind
 * the LAST page of the stream (any stream when has_serial == 0) that completes a packet and whose granule position
 * is <= absgp -- at the start of the physical stream if there is none -- so the position reached is <= absgp.
 * Bisection over the byte range, then a linear scan of the last interval. */
int lw_ogg_seek_absgp(lw_ogg_reader *r, int has_serial, uint32_t serial, uint64_t absgp);
/* CRC of RFC 3533 (polynomial 0x04c11db7, initial 0, no reflection) -- exposed for muxers and tests */
uint32_t lw_ogg_crc32(const uint8_t *data, size_t len, uint32_t crc);

/* OggStreamReader::from_ogg_reader (inside_ogg.rs:100-113) = read_headers (:30-49) + an empty PreviousWindowRight.
 * Takes ownership of `r` (also on failure).  The device context is created on `device` at the first decode, so
 * opening and header access work without a GPU.  *err: HeaderReadError / OggReadError code. */
lw_ogg_stream *lw_ogg_stream_open(lw_ogg_reader *r, int device, int *err);
void lw_ogg_stream_close(lw_ogg_stream *s);
/* pub fields ident_hdr / comment_hdr / setup_hdr (:72-74); owned by the stream, replaced at a chain boundary */
const lw_ident *lw_ogg_stream_ident(const lw_ogg_stream *s);
const lw_comment *lw_ogg_stream_comment(const lw_ogg_stream *s);
const lw_setup *lw_ogg_stream_setup(const lw_ogg_stream *s);
uint32_t lw_ogg_stream_serial(const lw_ogg_stream *s);                  /* stream_serial(), :288-290 */
uint32_t lw_ogg_stream_link_index(const lw_ogg_stream *s);              /* chain boundaries crossed so far (0 = first link) */
int lw_ogg_stream_last_absgp(const lw_ogg_stream *s, uint64_t *absgp); /* get_last_absgp(), :296-298: 1 = Some */
/* read_dec_packet_generic<S> (:195-206; read_dec_packet :167, read_dec_packet_itl :183): decodes the next audio
 * packet of the logical stream (chained streams re-initialise the context and prime it with their first audio
 * packet, :120-151), truncates the last packet of the stream to the final granule position (:219-227) and tracks
 * cur_absgp.  out: host memory for cap_elems elements, filled planar [ch][*n_samples] packed or interleaved.  LW_OK,
 * LW_OGG_EOF (Ok(None)), or a VorbisError: AudioReadError (BadAudio), HeaderReadError (BadHeader), OggReadError
 * (OggError) code.  LW_ERR_CAPACITY: cap_elems < channels << blocksize_1 of the CURRENT logical stream (it may just have
 * changed at a chain boundary: re-read lw_ogg_stream_ident); the packet is kept for the next call. */
int lw_ogg_stream_read_dec_packet(lw_ogg_stream *s, int fmt, void *out, size_t cap_elems, size_t *n_samples);
/* Look-ahead queue (the batched form of the call above): reads up to max_packets audio
 * packets of the current logical stream, decodes them with ONE batch (host entropy threads + one set of kernel
 * launches) and applies the same truncation / granule bookkeeping packet by packet.  out receives the packets'
 * blocks back to back; n_samples[i] / status[i] per packet (a packet that fails to decode has its AudioReadError in
 * status[i] and no samples).  Stops early at the end of the stream and in front of a chain boundary.
 * Returns LW_OK (also with *n_packets == 0 in front of a chain boundary: call lw_ogg_stream_read_dec_packet),
 * LW_OGG_EOF when no packet is left, or an error.
 * THREADING AND READ-AHEAD CONTRACT.  From the first call on the stream runs two library threads (a demultiplexer and an
 * entropy-staging thread) that keep reading the source through the reader's lw_ogg_io callbacks BETWEEN calls and after
 * a call has returned, holding up to about 5 x max_packets packets ahead of what the caller has been handed.  Hence:
 * (1) the io callbacks are invoked on library threads, never concurrently with each other, but concurrently with the
 * caller's own code -- they must not share unsynchronised state with it; (2) the position of the underlying source
 * between calls is unspecified (ahead of the last delivered packet); (3) every other entry point of the stream
 * (read_dec_packet, skip_samples_linear, seek_absgp_pg, into_inner, set_entropy_on_device, close) first stops both
 * threads and rolls the stream back to exactly what the caller has been handed, so the sequence of packets, errors
 * and samples is the one the packet-by-packet calls produce; (4) a device failure on the staging thread is reported
 * by the next call on the caller's thread, with its text republished through lw_last_device_error(). */
int lw_ogg_stream_read_dec_packets(lw_ogg_stream *s, int fmt, size_t max_packets, int n_threads, void *out,
		size_t cap_elems, uint32_t *n_samples, int32_t *status, size_t *n_packets);
/* Read-ahead behind the packet-by-packet call: with max_packets > 0, lw_ogg_stream_read_dec_packet hands out, one per call, the
 * packets of batches of up to max_packets that the look-ahead pipeline above decodes (n_threads host entropy threads; the entropy
 * stage on the device if set) -- the reference's own loop `while let Some(p) = rdr.read_dec_packet()? { .. }` (examples/perf.rs:35-44)
 * at the batched rate instead of one synchronous GPU round trip per packet, with no change to the loop.  What the caller observes is
 * the packet-by-packet sequence, call for call: samples, AudioReadError codes at the packets they belong to, get_last_absgp() as of
 * the packet just handed out, the last packet's truncation, LW_OGG_EOF, chain boundaries (crossed by the call itself).  Every other
 * entry point first returns the packets not yet handed out and re-makes the PreviousWindowRight as of the last one that was (one
 * synchronous decode of that packet: a decoded packet's right half depends on nothing before it, audio.rs:1125-1138).
 * A served batch's samples stay in the staging ring's pinned memory and are copied once, into the caller's buffer, at the call.
 * The threading and read-ahead contract above applies from the first call on.  0 turns it off (the default); more than 65 536
 * packets: LW_ERR_CAPACITY. */
int lw_ogg_stream_set_read_ahead(lw_ogg_stream *s, size_t max_packets, int n_threads);
/* Look-ahead batches with the entropy stage on the device (lw_ring_set_entropy_on_device) whenever the current logical
 * stream is eligible; other streams (and the packet-by-packet call) keep the host stage.  Results are identical. */
int lw_ogg_stream_set_entropy_on_device(lw_ogg_stream *s, int on);
/* skip_samples_linear<S> (:244-283): *got_packet = 0 is (None, left); otherwise the decoded packet that contains the
 * target is in out and *left samples of it remain to be skipped.  LW_ERR_CAPACITY as above: call again with
 * to_skip = *left and a buffer for the current logical stream. */
int lw_ogg_stream_skip_samples_linear(lw_ogg_stream *s, size_t to_skip, int fmt, void *out, size_t cap_elems,
		size_t *n_samples, size_t *left, int *got_packet);
/* seek_absgp_pg (:307-313) */
int lw_ogg_stream_seek_absgp_pg(lw_ogg_stream *s, uint64_t absgp);
/* into_inner (:111-113): gives the reader back and destroys the stream object */
lw_ogg_reader *lw_ogg_stream_into_inner(lw_ogg_stream *s);

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
/* The same for the block kernel k_short<L> (L = lanes per block = block size / 32; section offsets: LwBlkLayout<L> in
 * csrc/lw_fast.hpp) and its unit list (8 bytes per unit: channel a, channel b or -1, coupled, floor slot a / b, post count
 * a / b, 0).  blockflag 0: the short blocks; 1: the long blocks of a stream k_long does not cover.  *n_units: in = room in
 * units8, out = units of the stream.  0 if those blocks are not covered. */
size_t lw_debug_short_image(const lw_ident *id, const lw_setup *s, int blockflag, uint8_t *dst, size_t cap, uint8_t *units8,
		size_t *n_units, uint32_t *lanes);

/* Census hook (host only, no GPU needed): which synthesis kernels the blocks of this stream shape are routed to, and the reason
 * where a faster one does not apply, as one line of text: "long=<kernel> | short=<kernel> | transitions=<how long blocks with a short
 * slope are done> | entropy=<device | host (why)>".  Returns the length of the whole text; copies at most cap - 1 characters. */
size_t lw_debug_plan_census(const lw_ident *id, const lw_setup *s, char *dst, size_t cap);

/* Library/version introspection */
const char *lw_version(void);
/* Host threads the entropy stage uses when a call passes n_threads <= 0: the CPUs this process may run on (hardware
 * threads, cut to the affinity mask and to the container's CFS quota, cgroup cpu.max); LW_HOST_THREADS overrides. */
int lw_default_host_threads(void);

#ifdef __cplusplus
}
#endif
#endif
