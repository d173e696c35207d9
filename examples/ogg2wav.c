/*
 * ogg2wav -- decode an Ogg/Vorbis file to a 16-bit PCM WAV file on the MI355X path (C ABI only, include/lewton_amd.h):
 * the interleaved output of `OggStreamReader::read_dec_packet_itl` (inside_ogg.rs:183-190) written behind a RIFF header.
 * Chained files are written as one WAV when all links share the channel count and sample rate.
 *
 *   ogg2wav in.ogg out.wav [look-ahead packets, default 1024]
 */
#include "lewton_amd.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void put_u32(unsigned char *p, unsigned long v) { p[0] = v & 255, p[1] = (v >> 8) & 255, p[2] = (v >> 16) & 255, p[3] = (v >> 24) & 255; }
static void put_u16(unsigned char *p, unsigned v) { p[0] = v & 255, p[1] = (v >> 8) & 255; }

static int write_header(FILE *f, unsigned channels, unsigned rate, unsigned long data_bytes)
{
	unsigned char h[44];
	memcpy(h, "RIFF", 4);
	put_u32(h + 4, 36 + data_bytes);
	memcpy(h + 8, "WAVEfmt ", 8);
	put_u32(h + 16, 16);
	put_u16(h + 20, 1); /* PCM */
	put_u16(h + 22, channels);
	put_u32(h + 24, rate);
	put_u32(h + 28, (unsigned long)rate * channels * 2);
	put_u16(h + 32, channels * 2);
	put_u16(h + 34, 16);
	memcpy(h + 36, "data", 4);
	put_u32(h + 40, data_bytes);
	return fwrite(h, 1, 44, f) == 44 ? 0 : -1;
}

int main(int argc, char **argv)
{
	if (argc < 3) {
		fprintf(stderr, "usage: ogg2wav in.ogg out.wav [look-ahead packets]\n");
		return 2;
	}
	const size_t K = argc > 3 ? (size_t)strtoul(argv[3], NULL, 10) : 1024;
	int err = 0;
	lw_ogg_reader *rdr = lw_ogg_reader_open_file(argv[1], &err);
	lw_ogg_stream *srr = rdr ? lw_ogg_stream_open(rdr, 0, &err) : NULL;
	if (!srr) {
		fprintf(stderr, "cannot open %s as Ogg/Vorbis (%d)\n", argv[1], err);
		return 1;
	}
	lw_ident_info info;
	lw_ident_get_info(lw_ogg_stream_ident(srr), &info);
	const unsigned channels = info.audio_channels, rate = info.audio_sample_rate;
	FILE *out = fopen(argv[2], "wb");
	if (!out || write_header(out, channels, rate, 0)) {
		fprintf(stderr, "cannot write %s\n", argv[2]);
		return 1;
	}
	size_t cap = ((size_t)channels << info.blocksize_1) * (K ? K : 1);
	int16_t *buf = (int16_t *)malloc(cap * sizeof(int16_t));
	uint32_t *ns = (uint32_t *)malloc((K ? K : 1) * sizeof(uint32_t));
	int32_t *st = (int32_t *)malloc((K ? K : 1) * sizeof(int32_t));
	unsigned long total = 0; /* samples per channel */
	for (;;) {
		size_t got = 0, m = 0;
		int rc = K > 1 ? lw_ogg_stream_read_dec_packets(srr, LW_FMT_I16_INTERLEAVED, K, 0, buf, cap, ns, st, &got)
		               : LW_OK;
		if (rc == LW_OK && got == 0) { /* single-packet mode, or a chain boundary in look-ahead mode */
			rc = lw_ogg_stream_read_dec_packet(srr, LW_FMT_I16_INTERLEAVED, buf, cap, &m);
			if (rc == LW_ERR_CAPACITY) { /* the next link needs a larger buffer */
				lw_ident_get_info(lw_ogg_stream_ident(srr), &info);
				cap = ((size_t)info.audio_channels << info.blocksize_1) * (K ? K : 1);
				buf = (int16_t *)realloc(buf, cap * sizeof(int16_t));
				continue;
			}
			if (rc == LW_OK) {
				lw_ident_get_info(lw_ogg_stream_ident(srr), &info);
				if (info.audio_channels != channels || info.audio_sample_rate != rate) {
					fprintf(stderr, "chained stream changes the format: stopping\n");
					break;
				}
			}
		} else if (rc == LW_OK) {
			for (size_t i = 0; i < got; i++) {
				if (st[i] != LW_OK) {
					fprintf(stderr, "undecodable packet (%d): skipped\n", st[i]);
					continue;
				}
				m += ns[i];
			}
		}
		if (rc == LW_OGG_EOF)
			break;
		if (rc != LW_OK) {
			fprintf(stderr, "decode error %d %s\n", rc, lw_last_device_error());
			return 1;
		}
		if (m && fwrite(buf, sizeof(int16_t) * channels, m, out) != m) {
			fprintf(stderr, "write error\n");
			return 1;
		}
		total += m;
	}
	fseek(out, 0, SEEK_SET);
	write_header(out, channels, rate, total * channels * 2);
	fclose(out);
	lw_ogg_stream_close(srr);
	printf("%s: %u channels, %u Hz, %lu samples per channel (%.3f s)\n", argv[2], channels, rate, total, (double)total / rate);
	free(buf);
	free(ns);
	free(st);
	return 0;
}
