// lewton's C API (src/capi.rs:13-147) on top of the library's own C ABI.
#include "../../include/lewton.h"
#include "../../include/lewton_amd.h"

#include <cstdlib>
#include <vector>

struct LewtonContext {
	lw_ident *ident = nullptr;
	lw_setup *setup = nullptr;
	lw_decoder *dec = nullptr; // created at the first decode (context creation needs no GPU, like the reference's)
	lw_pwr *pwr = nullptr;
	uint8_t channels = 0;
	size_t cap = 0; // 1 << blocksize_1
};

struct LewtonSamples {
	size_t channels = 0, count = 0;
	std::vector<float> data; // [channel][count]
};

namespace {

// read_xiph_lacing, capi.rs:21-35: sum of bytes up to and including the first one below 255
bool read_xiph_lacing(const uint8_t *&p, size_t &n, uint64_t &out)
{
	uint64_t r = 0;
	for (;;) {
		if (n == 0)
			return false;
		const uint64_t v = *p++;
		n--;
		r += v;
		if (v < 255) {
			out = r;
			return true;
		}
	}
}

} // namespace

extern "C" {

LewtonContext *lewton_context_from_extradata(const uint8_t *data, size_t len)
{
	if (!data)
		return nullptr;
	// "We must start with a 2 as per matroska encapsulation spec" (capi.rs:39-42)
	if (len == 0 || data[0] != 2)
		return nullptr;
	const uint8_t *p = data + 1;
	size_t n = len - 1;
	uint64_t ident_len = 0, comment_len = 0;
	if (!read_xiph_lacing(p, n, ident_len) || !read_xiph_lacing(p, n, comment_len))
		return nullptr;
	if (ident_len > n || comment_len > n - ident_len)
		return nullptr; // the reference's slice indexing would panic here
	int err = 0;
	lw_ident *id = lw_read_header_ident(p, (size_t)ident_len, &err);
	if (!id)
		return nullptr;
	p += ident_len + comment_len; // the comment header is skipped, not parsed (capi.rs:50)
	n -= (size_t)(ident_len + comment_len);
	lw_ident_info info;
	lw_ident_get_info(id, &info);
	lw_setup *st = lw_read_header_setup(p, n, info.audio_channels, info.blocksize_0, info.blocksize_1, &err);
	if (!st) {
		lw_ident_free(id);
		return nullptr;
	}
	auto *cx = new LewtonContext();
	cx->ident = id;
	cx->setup = st;
	cx->channels = info.audio_channels;
	cx->cap = (size_t)1 << info.blocksize_1;
	return cx;
}

void lewton_context_reset(LewtonContext *ctx)
{
	if (ctx && ctx->pwr)
		lw_pwr_reset(ctx->pwr);
}

int lewton_decode_packet(LewtonContext *ctx, const uint8_t *pkt, size_t len, LewtonSamples **sample_out)
{
	if (!pkt || !ctx || !sample_out)
		return 1;
	if (!ctx->dec) {
		int err = 0, device = 0;
		if (const char *e = std::getenv("LEWTON_AMD_DEVICE"))
			device = std::atoi(e);
		ctx->dec = lw_decoder_create(ctx->ident, ctx->setup, device, &err);
		if (!ctx->dec)
			return 2;
	}
	if (!ctx->pwr) {
		ctx->pwr = lw_pwr_new(ctx->dec);
		if (!ctx->pwr)
			return 2;
	}
	auto *s = new LewtonSamples();
	s->channels = ctx->channels;
	s->data.resize((size_t)ctx->channels * ctx->cap);
	size_t m = 0;
	if (lw_read_audio_packet(ctx->dec, pkt, len, ctx->pwr, LW_FMT_F32_PLANAR, s->data.data(), ctx->cap, &m) != LW_OK) {
		delete s;
		return 2;
	}
	s->count = m;
	s->data.resize((size_t)ctx->channels * m); // packed [channel][m]
	*sample_out = s;
	return 0;
}

size_t lewton_samples_count(const LewtonSamples *samples)
{
	return samples && samples->channels ? samples->count : 0;
}

const float *lewton_samples_f32(const LewtonSamples *samples, size_t channel)
{
	if (!samples || channel >= samples->channels)
		return nullptr;
	// a Vec's pointer is never NULL, also for an empty channel vector (capi.rs:134-137)
	static const float kEmpty = 0.0f;
	return samples->count ? samples->data.data() + channel * samples->count : &kEmpty;
}

void lewton_samples_drop(LewtonSamples *samples)
{
	delete samples;
}

void lewton_context_drop(LewtonContext *ctx)
{
	if (!ctx)
		return;
	if (ctx->pwr)
		lw_pwr_free(ctx->pwr);
	if (ctx->dec)
		lw_decoder_destroy(ctx->dec);
	lw_setup_free(ctx->setup);
	lw_ident_free(ctx->ident);
	delete ctx;
}

} // extern "C"
