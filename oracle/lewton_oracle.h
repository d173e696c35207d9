/*
 * lewton_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the algorithm of RustAudio/lewton 0.10.2 for the
 * Vorbis audio-packet decode path (`audio::read_audio_packet*`) and the header
 * parsing that feeds it.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (lewton_amd/csrc) never
 * links, imports or calls it.
 *
 * Parity pin: the IMDCT, bit-reverse table, render_point, neighbour search,
 * bit reader, Huffman tree, ilog, lookup1_values, float32_unpack and ident
 * header are checked against the reference's own known answers
 * (tests/golden/reference_vectors.json, extracted from /root/reference/src by
 * tests/golden/make_golden.py).  Residue decode, decoupling, render_line,
 * windowing/overlap-add and i16 conversion have NO in-tree vector in the
 * reference: for those rows parity is UNPINNED and rests on
 * this restatement plus definitional cross-checks in tests/.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).
 */
#ifndef LEWTON_ORACLE_H
#define LEWTON_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* AudioReadError, src/audio.rs:26-41 (0 = Ok) */
enum {
	LWO_OK = 0,
	LWO_AUDIO_END_OF_PACKET = 1,
	LWO_AUDIO_BAD_FORMAT = 2,
	LWO_AUDIO_IS_HEADER = 3,
	LWO_AUDIO_BUFFER_NOT_ADDRESSABLE = 4,
	/* HeaderReadError, src/header.rs:35-63 */
	LWO_HDR_END_OF_PACKET = 16,
	LWO_HDR_NOT_VORBIS = 17,
	LWO_HDR_UNSUPPORTED_VERSION = 18,
	LWO_HDR_BAD_FORMAT = 19,
	LWO_HDR_BAD_TYPE = 20,
	LWO_HDR_IS_AUDIO = 21,
	LWO_HDR_UTF8 = 22,
	LWO_HDR_BUFFER_NOT_ADDRESSABLE = 23,
	/* not a lewton result: the reference PANICS on this input (out-of-range index, division by zero).  A decoder behind a C
	 * ABI must report an error instead of unwinding or touching memory; tests accept any non-OK status for it. */
	LWO_REF_PANIC = 64
};

typedef struct lwo_ident lwo_ident;
typedef struct lwo_setup lwo_setup;
typedef struct lwo_pwr lwo_pwr;

/* ---- headers (src/header.rs) ---- */
lwo_ident *lwo_read_header_ident(const uint8_t *pkt, size_t len, int *err);
void lwo_ident_free(lwo_ident *id);
/* fields: 0 channels, 1 sample rate, 2 bitrate max, 3 nominal, 4 min, 5 bs0, 6 bs1 */
int64_t lwo_ident_field(const lwo_ident *id, int which);
lwo_setup *lwo_read_header_setup(const uint8_t *pkt, size_t len, uint8_t channels,
		uint8_t bs0, uint8_t bs1, int *err);
void lwo_setup_free(lwo_setup *s);
/* introspection used by tests: counts 0 codebooks,1 floors,2 residues,3 mappings,4 modes */
int lwo_setup_count(const lwo_setup *s, int which);

/* ---- state (src/audio.rs:847-861) ---- */
lwo_pwr *lwo_pwr_new(void);
lwo_pwr *lwo_pwr_clone(const lwo_pwr *p);
int lwo_pwr_is_empty(const lwo_pwr *p);
void lwo_pwr_reset(lwo_pwr *p);
void lwo_pwr_free(lwo_pwr *p);
/* per-channel length of the stored right half (0 if empty) and a copy-out */
size_t lwo_pwr_len(const lwo_pwr *p);
int lwo_pwr_copy(const lwo_pwr *p, float *dst /* [ch][len] */);

/* ---- packet decode (src/audio.rs:874, :919, :1170) ---- */
int lwo_get_decoded_sample_count(const lwo_ident *id, const lwo_setup *s,
		const uint8_t *pkt, size_t len, size_t *count);

/* Optional taps = the four record_*! points (src/lib.rs:56-94, audio.rs:988,1004,1041,1054).
 * Each, when non-NULL, receives [ch][n/2] (first three) or [ch][n] (post_mdct) floats. */
typedef struct {
	float *residue_pre_inverse;
	float *residue_post_inverse;
	float *pre_mdct;
	float *post_mdct;
	uint32_t n; /* out: block size of the packet */
} lwo_taps;

/* Decode one packet to planar f32 (the `S = Vec<Vec<f32>>` instantiation).
 * out_planar must hold ch * (1 << bs1) / 2 ... the caller passes cap = floats available
 * per channel; *n_samples receives the per-channel count. Returns LWO_OK or an
 * AudioReadError code. */
int lwo_read_audio_packet_f32(const lwo_ident *id, const lwo_setup *s,
		const uint8_t *pkt, size_t len, lwo_pwr *pwr,
		float *out_planar, size_t cap_per_channel, size_t *n_samples, lwo_taps *taps);
/* `S = Vec<Vec<i16>>` (read_audio_packet, audio.rs:1170) */
int lwo_read_audio_packet_i16(const lwo_ident *id, const lwo_setup *s,
		const uint8_t *pkt, size_t len, lwo_pwr *pwr,
		int16_t *out_planar, size_t cap_per_channel, size_t *n_samples);
/* `S = InterleavedSamples<i16>` (samples.rs:55-78) */
int lwo_read_audio_packet_i16_itl(const lwo_ident *id, const lwo_setup *s,
		const uint8_t *pkt, size_t len, lwo_pwr *pwr,
		int16_t *out_itl, size_t cap_per_channel, size_t *n_samples);

/* Decode a whole list of packets of one stream (perf.rs-shaped loop, examples/perf.rs:35-42):
 * returns total per-channel samples; output discarded unless out != NULL (planar appended
 * per packet as [ch][m]).  Used by the CPU baseline. */
int lwo_decode_stream_i16(const lwo_ident *id, const lwo_setup *s,
		const uint8_t *data, const uint64_t *offsets, const uint32_t *lens, size_t n_packets,
		lwo_pwr *pwr, int16_t *out, size_t out_cap, uint64_t *total_samples, double *seconds_synth);

/* test hook: bits consumed by the entropy stage of the most recent packet (not thread safe) */
size_t lwo_debug_bits_consumed(void);
/* bench hook: accumulate the time of the bit-serial stage (audio.rs:921-986) of every packet decoded while on */
void lwo_stage_timing(int on);
double lwo_stage_entropy_seconds(void);

/* ---- unit-level entry points for known-answer tests ---- */
/* header_cached.rs:34-110 -- tables for one blocksize; arrays sized n/2,n/2,n/4,n/2,n/8 */
void lwo_tables(uint8_t bs, float *A, float *B, float *C, float *window, uint32_t *bitrev);
/* imdct.rs:291 -- in place on n = 1<<bs floats */
void lwo_inverse_mdct(uint8_t bs, float *buffer);
/* audio.rs:792-825 definitional transform (O(n^2)) */
void lwo_inverse_mdct_slow(float *buffer, size_t n);
uint32_t lwo_render_point(uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t x); /* audio.rs:354 */
/* audio.rs:285/:290 -- returns 0 and fills idx/val, or -1 where the reference panics */
int lwo_low_neighbor(const uint32_t *v, size_t x, size_t *idx, uint32_t *val);
int lwo_high_neighbor(const uint32_t *v, size_t x, size_t *idx, uint32_t *val);
/* audio.rs:503 -- appends x1-x0 values to out, returns count */
size_t lwo_render_line(uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, uint32_t *out);
/* audio.rs:391 + :526: y list (header order) -> n/2 floor values; needs a floor-1 config */
int lwo_floor1_curve(const lwo_setup *s, int floor_idx, const uint32_t *y, uint32_t n_half,
		float *out, uint32_t *final_y, uint8_t *step2);
uint8_t lwo_ilog(uint64_t v);                                   /* lib.rs:159 */
uint32_t lwo_lookup1_values(uint32_t entries, uint16_t dims);   /* header.rs:616 */
float lwo_float32_unpack(uint32_t v);                           /* bitpacking.rs:304 */
void lwo_inverse_couple(float m, float a, float *nm, float *na);/* audio.rs:763 */
int16_t lwo_sample_i16(float f);                                /* samples.rs:92-103 */
const float *lwo_inverse_db_table(void);                        /* audio.rs:437-501 */
/* bit reader (bitpacking.rs): read `n` bit-fields of widths[] from data; values out; returns
 * number of successful reads (a failed read does not advance, like the reference). */
size_t lwo_bitread_seq(const uint8_t *data, size_t len, const uint8_t *widths, size_t n, uint64_t *vals);
/* Huffman (huffman_tree.rs:183): 0 ok, 1 overspecified, 2 underpopulated, 3 invalid single entry.
 * If ok and bits != NULL decodes up to max_syms symbols from the bitstream. */
int lwo_huffman_check(const uint8_t *lengths, size_t n_entries, const uint8_t *bits, size_t bits_len,
		uint32_t *syms, size_t max_syms, size_t *n_syms);

#ifdef __cplusplus
}
#endif
#endif
