/*
 * perf -- the counterpart of the reference's examples/perf.rs (open an Ogg/Vorbis file, decode every packet, discard
 * the samples, report the time) on the MI355X decode path, written against the C ABI only (include/lewton_amd.h).
 *
 *   perf file.ogg            packet by packet, like `while let Some(pck) = srr.read_dec_packet()?`
 *   perf file.ogg K [T]      look-ahead queue: K packets per batch (one set of kernel launches), T host entropy threads
 *   perf file.ogg K T dev    the same with the entropy stage on the device (eligible streams; T is then of no consequence)
 *   perf file.ogg K T host|dev ahead   the packet-by-packet loop of the first form, served from batches of K packets decoded ahead
 *                            (lw_ogg_stream_set_read_ahead): the reference's loop unchanged, at the batched rate
 *
 * Build: cc -O2 -Iinclude examples/perf.c -Llewton_amd/_lib -llewton_amd -Wl,-rpath,$PWD/lewton_amd/_lib -o examples/perf
 */
#include "lewton_amd.h"

#include <stdio.h>
#include <stdlib.h>
#include <time.h>

static double now(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char **argv)
{
	if (argc < 2) {
		fprintf(stderr, "No arg found. Please specify a file to open.\n");
		return 2;
	}
	size_t K = argc > 2 ? (size_t)strtoul(argv[2], NULL, 10) : 0;
	const int threads = argc > 3 ? atoi(argv[3]) : 0;
	const int dev_entropy = argc > 4 && argv[4][0] == 'd';
	const size_t read_ahead = argc > 5 ? K : 0;
	if (read_ahead)
		K = 0; /* the loop below is the packet-by-packet one */
	int err = 0;
	printf("Opening file: %s\n", argv[1]);
	lw_ogg_reader *rdr = lw_ogg_reader_open_file(argv[1], &err);
	if (!rdr) {
		fprintf(stderr, "Can't open file (%d)\n", err);
		return 1;
	}
	lw_ogg_stream *srr = lw_ogg_stream_open(rdr, 0, &err); /* OggStreamReader::new */
	if (!srr) {
		fprintf(stderr, "Error: %d\n", err);
		return 1;
	}
	if (dev_entropy)
		lw_ogg_stream_set_entropy_on_device(srr, 1);
	if (read_ahead)
		lw_ogg_stream_set_read_ahead(srr, read_ahead, threads);
	lw_ident_info info;
	lw_ident_get_info(lw_ogg_stream_ident(srr), &info);
	printf("Sample rate: %u\n", info.audio_sample_rate);

	const size_t cap1 = (size_t)info.audio_channels << info.blocksize_1; /* elements of one packet's block */
	const size_t cap = cap1 * (K ? K : 1);
	int16_t *buf = (int16_t *)malloc(cap * sizeof(int16_t));
	uint32_t *ns = (uint32_t *)malloc((K ? K : 1) * sizeof(uint32_t));
	int32_t *st = (int32_t *)malloc((K ? K : 1) * sizeof(int32_t));
	size_t n = 0;
	double len_play = 0.0;
	const double t0 = now();
	double t_first = 0.0; /* end of the first call: it includes creating the device context (HIP start-up, table upload) */
	size_t n_first = 0;
	for (;;) {
		int rc;
		if (K) {
			size_t got = 0;
			rc = lw_ogg_stream_read_dec_packets(srr, LW_FMT_I16_PLANAR, K, threads, buf, cap, ns, st, &got);
			if (rc == LW_OK && got == 0) { /* chain boundary: cross it with the single-packet call */
				size_t m = 0;
				rc = lw_ogg_stream_read_dec_packet(srr, LW_FMT_I16_PLANAR, buf, cap, &m);
				if (rc == LW_OK) {
					n++;
					lw_ident_get_info(lw_ogg_stream_ident(srr), &info);
					len_play += (double)m / info.audio_sample_rate;
				}
			} else if (rc == LW_OK) {
				for (size_t i = 0; i < got; i++) {
					if (st[i] != LW_OK) {
						fprintf(stderr, "Error: packet %zu: %d\n", n + i, st[i]);
						return 1;
					}
					len_play += (double)ns[i] / info.audio_sample_rate;
				}
				n += got;
			}
		} else {
			size_t m = 0;
			rc = lw_ogg_stream_read_dec_packet(srr, LW_FMT_I16_PLANAR, buf, cap, &m);
			if (rc == LW_OK) {
				n++;
				len_play += (double)m / info.audio_sample_rate;
			}
		}
		if (t_first == 0.0) {
			t_first = now();
			n_first = n;
		}
		if (rc == LW_OGG_EOF)
			break;
		if (rc == LW_ERR_CAPACITY) { /* next link of a chained file has more channels or larger blocks */
			fprintf(stderr, "Error: chained stream needs a larger buffer\n");
			return 1;
		}
		if (rc != LW_OK) {
			fprintf(stderr, "Error: %d %s\n", rc, lw_last_device_error());
			return 1;
		}
	}
	const double dt = now() - t0;
	printf("The piece is %g s long (%zu packets).\n", len_play, n);
	printf("Decoded in %g s (%.0f packets/s, %.1fx real time).\n", dt, (double)n / dt, len_play / dt);
	if (n > n_first)
		printf("Without the first call (device start-up, %g s): %.0f packets/s.\n", t_first - t0,
				(double)(n - n_first) / (t0 + dt - t_first));
	lw_ogg_stream_close(srr);
	free(buf);
	free(ns);
	free(st);
	return 0;
}
