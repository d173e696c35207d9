// Sanitizer harness for the Ogg demultiplexer (lewton_amd/csrc/lw_ogg.cpp): the container is untrusted input.  Built by
// tests/test_fuzz_host.py with g++ -fsanitize=address,undefined from the PRODUCT source; the packet decoder behind the
// stream layer is stubbed out (the page / packet reader and the seek are what is exercised here).
// Input file: u32 n_cases, then per case: u32 len, bytes.
#include "../../lewton_amd/csrc/lw_ogg.cpp"

#include <cstdio>
#include <vector>

// ---- stubs for the rest of the C ABI (never reached with a NULL decoder; they only have to link)
void lw_set_device_error(const std::string &) {}
extern "C" {
const char *lw_last_device_error(void) { return ""; }
lw_ident *lw_read_header_ident(const uint8_t *, size_t, int *err) { if (err) *err = LW_HDR_NOT_VORBIS; return nullptr; }
int lw_ident_get_info(const lw_ident *, lw_ident_info *) { return LW_ERR_NULL_ARG; }
void lw_ident_free(lw_ident *) {}
lw_setup *lw_read_header_setup(const uint8_t *, size_t, uint8_t, uint8_t, uint8_t, int *err) { if (err) *err = LW_HDR_BAD_FORMAT; return nullptr; }
void lw_setup_free(lw_setup *) {}
lw_comment *lw_read_header_comment(const uint8_t *, size_t, int *err) { if (err) *err = LW_HDR_BAD_FORMAT; return nullptr; }
void lw_comment_free(lw_comment *) {}
lw_decoder *lw_decoder_create(const lw_ident *, const lw_setup *, int, int *err) { if (err) *err = LW_ERR_DEVICE; return nullptr; }
void lw_decoder_destroy(lw_decoder *) {}
lw_pwr *lw_pwr_new(lw_decoder *) { return nullptr; }
void lw_pwr_reset(lw_pwr *) {}
void lw_pwr_free(lw_pwr *) {}
int lw_get_decoded_sample_count(const lw_ident *, const lw_setup *, const uint8_t *, size_t, size_t *) { return LW_ERR_NULL_ARG; }
int lw_read_audio_packet(lw_decoder *, const uint8_t *, size_t, lw_pwr *, int, void *, size_t, size_t *) { return LW_ERR_DEVICE; }
lw_ring *lw_ring_create(lw_decoder *, size_t, size_t, int, int *err) { if (err) *err = LW_ERR_DEVICE; return nullptr; }
void lw_ring_destroy(lw_ring *) {}
int lw_ring_stage(lw_ring *, const lw_packet *, size_t, int) { return LW_ERR_DEVICE; }
int lw_ring_launch(lw_ring *) { return LW_ERR_DEVICE; }
int lw_ring_collect(lw_ring *, const lw_packet_result **, size_t *, const void **, size_t *) { return LW_ERR_DEVICE; }
int lw_ring_release(lw_ring *) { return LW_ERR_DEVICE; }
int lw_ring_drain(lw_ring *) { return LW_OK; }
int lw_ring_set_entropy_on_device(lw_ring *, int) { return LW_ERR_DEVICE; }
void lw_pwr_get_state(const lw_pwr *, lw_pwr_state *) {}
void lw_pwr_set_state(lw_pwr *, const lw_pwr_state *) {}
lw_batch *lw_batch_create(lw_decoder *, size_t, int, int *) { return nullptr; }
void lw_batch_destroy(lw_batch *) {}
int lw_batch_entropy(lw_batch *, const lw_packet *, size_t, int) { return LW_ERR_DEVICE; }
int lw_batch_upload(lw_batch *, void *) { return LW_ERR_DEVICE; }
int lw_batch_synth_to_host(lw_batch *, void *, size_t, void *) { return LW_ERR_DEVICE; }
size_t lw_batch_out_elems(const lw_batch *) { return 0; }
const lw_packet_result *lw_batch_results(const lw_batch *) { return nullptr; }
}

static bool rd(FILE *f, uint32_t &v) { return fread(&v, 4, 1, f) == 1; }

int main(int argc, char **argv)
{
	if (argc < 2)
		return 2;
	FILE *f = fopen(argv[1], "rb");
	if (!f)
		return 2;
	uint32_t n_cases = 0;
	if (!rd(f, n_cases))
		return 2;
	size_t packets = 0, bytes = 0, errors = 0, seeks = 0;
	for (uint32_t c = 0; c < n_cases; c++) {
		uint32_t len = 0;
		if (!rd(f, len) || len > (64u << 20))
			return 2;
		// exact-size heap copy: reads past the end of the container are caught
		std::vector<uint8_t> data(len);
		if (len && fread(data.data(), 1, len, f) != len)
			return 2;
		for (int pass = 0; pass < 2; pass++) {
			lw_ogg_reader *r = lw_ogg_reader_open_memory(data.data(), data.size(), pass);
			if (!r)
				return 1;
			lw_ogg_packet k;
			int rc;
			unsigned n = 0;
			while ((rc = lw_ogg_read_packet(r, &k)) == LW_OK) {
				packets++;
				for (size_t i = 0; i < k.len; i += 97)
					bytes += k.data[i]; // touch the payload
				if (k.len)
					bytes += k.data[k.len - 1];
				if (++n % 7 == 3) {
					// page-granular seek to a position derived from the data, then carry on reading
					const uint64_t goal = (uint64_t)k.absgp_page / 2 + n;
					if (lw_ogg_seek_absgp(r, (int)(n & 1), k.stream_serial, goal) == LW_OK)
						seeks++;
					if (n > 200)
						break;
				}
			}
			if (rc != LW_OGG_EOF && rc != LW_OK)
				errors++;
			// the stream layer on top: header bootstrap must fail cleanly (stubs reject every header)
			lw_ogg_delete_unread_packets(r);
			lw_ogg_seek_absgp(r, 0, 0, 0);
			int err = 0;
			lw_ogg_stream *s = lw_ogg_stream_open(r, 0, &err); // takes ownership of r
			if (s)
				lw_ogg_stream_close(s);
		}
	}
	fclose(f);
	printf("cases %u, packets %zu, errors %zu, seeks %zu, checksum %zu\n", n_cases, packets, errors, seeks, bytes);
	return 0;
}
