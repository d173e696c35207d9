/*
 * lewton.h -- the C API of lewton (src/capi.rs:13-147; header name and guard from cbindgen.toml:1-7), served by the
 * MI355X-native decode path of liblewton_amd.so.  Same symbols, arguments and return values as the reference's
 * `capi` feature, so an FFmpeg-style caller relinks against this library unchanged.
 *
 * Notes on parity:
 *  - `lewton_samples_f32` is declared `pub unsafe extern fn` WITHOUT `#[no_mangle]` in the reference (capi.rs:132), so the
 *    reference's cdylib does not export it by that name although its own documentation (capi.rs:62-72) tells callers
 *    to fetch channel data; this library exports it.
 *  - Where the reference would panic inside the FFI call (extradata shorter than its own lacing values announce,
 *    capi.rs:46-52) this library returns NULL.
 *  - The device stage runs on GPU `LEWTON_AMD_DEVICE` (environment, default 0); there is no CPU fallback: without a
 *    usable GPU `lewton_decode_packet` returns 2 like any other decode failure.
 */
#ifndef LEWTON_LEWTON_H
#define LEWTON_LEWTON_H

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Main Decoder State (capi.rs:10-19): PreviousWindowRight + ident header + setup header */
typedef struct LewtonContext LewtonContext;
/* A multichannel vector of samples, `Vec<Vec<f32>>` (capi.rs:62-73) */
typedef struct LewtonSamples LewtonSamples;

/* capi.rs:75-91: context from a xiph-laced extradata bundle (Matroska CodecPrivate: 0x02, laced lengths of the ident
 * and comment headers, then ident, comment, setup).  NULL on any failure. */
LewtonContext *lewton_context_from_extradata(const uint8_t *data, size_t len);
/* capi.rs:93-97: reset the decoder to support seeking (`pwr = PreviousWindowRight::new()`) */
void lewton_context_reset(LewtonContext *ctx);
/* capi.rs:99-121: 0 on success (*sample_out = newly allocated samples), 1 on a NULL argument, 2 if the packet
 * cannot be decoded */
int lewton_decode_packet(LewtonContext *ctx, const uint8_t *pkt, size_t len, LewtonSamples **sample_out);
/* capi.rs:123-130: number of samples present in each channel */
size_t lewton_samples_count(const LewtonSamples *samples);
/* capi.rs:132-138: the channel's sample data, NULL if there is no such channel */
const float *lewton_samples_f32(const LewtonSamples *samples, size_t channel);
/* capi.rs:140-143 */
void lewton_samples_drop(LewtonSamples *samples);
/* capi.rs:145-147 */
void lewton_context_drop(LewtonContext *ctx);

#ifdef __cplusplus
}
#endif

#endif /* LEWTON_LEWTON_H */
