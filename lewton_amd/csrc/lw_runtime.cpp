// Runtime + C ABI of the MI355X audio-packet decode path (product code, compiled with hipcc).
//
// Implements include/lewton_amd.h: header objects, the device context (tables in HBM), the
// device-resident PreviousWindowRight pool, batches with pinned staging, and the drop-in
// single-packet call.  There is no CPU fallback for the synthesis stage: without a usable GPU every
// device call returns LW_ERR_DEVICE.
#include "../../include/lewton_amd.h"

#include "lw_entropy.hpp"
#include "lw_fast.hpp"
#include "lw_host.hpp"
#include "lw_kernels.hpp"
#include "lw_pool.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#define LW_ERR_UNSUPPORTED_STREAM LW_AUDIO_BAD_FORMAT
// packets a worker claims at a time: small enough that 64 threads share a 4096-packet batch evenly to the end
#define LW_ENTROPY_CHUNK 4

namespace {

thread_local std::string g_dev_err;

bool hip_ok(hipError_t e, const char *what)
{
	if (e == hipSuccess)
		return true;
	g_dev_err = std::string(what) + ": " + hipGetErrorString(e);
	(void)hipGetLastError();
	return false;
}

#define HIP_TRY(expr)                        \
	do {                                     \
		if (!hip_ok((expr), #expr))          \
			return LW_ERR_DEVICE;            \
	} while (0)

extern const float kInverseDbTable[256];

} // namespace

struct lw_ident {
	std::shared_ptr<lw::Ident> p;
};
struct lw_setup {
	std::shared_ptr<lw::Setup> p;
};
struct lw_comment {
	std::unique_ptr<lw::Comment> p;
};

struct lw_decoder {
	std::shared_ptr<lw::Ident> id;
	std::shared_ptr<lw::Setup> setup;
	int device = 0;
	int n_cus = 256;
	LwDevTables T{};
	void *d_blob = nullptr; // one allocation holding every table
	bool any_coupling = false;
	bool any_floor0 = false; // some floor is of type 0: batches carry explicit floor curves
	bool symbols_ok = false; // Tier B (device-side inverse VQ) is possible for this stream
	std::string symbols_why;
	LwVqTables V{};
	void *d_vq_blob = nullptr;
	std::vector<uint32_t> vq_book_ends; // cumulative float offsets of the book tables in V.vq (ascending table size)
	uint32_t max_posts = 2;
	std::vector<uint64_t> mode_floor_bytes; // per mode: bytes of floor input over all channels (SURVEY 8(d) accounting)
	// PreviousWindowRight pool: [slots][2][ch][n1/2] floats
	std::mutex mu;
	float *d_state = nullptr;
	size_t state_cap = 0;
	std::vector<int> free_slots;
	LwFastPlan fast;               // specialised long-block kernel: eligibility, units, LDS image
	uint8_t *d_fast_image = nullptr;
	LwFastUnit *d_fast_units = nullptr;
	lw_batch *one = nullptr; // internal batch for lw_read_audio_packet
	void *one_out = nullptr; // pinned host output for the single-packet path
	size_t one_out_bytes = 0;
};

struct lw_pwr {
	lw_decoder *dec = nullptr;
	int slot = -1;
	bool present = false;
	uint32_t len = 0;   // per-channel length
	uint8_t parity = 0; // which of the two buffers holds the valid state
};

struct lw_batch {
	lw_decoder *dec = nullptr;
	size_t max_packets = 0;
	int fmt = 0;
	uint8_t *h_slab = nullptr, *d_slab = nullptr; // all host->device buffers below are slices of these
	size_t slab_bytes = 0;
	LwPacketRec *h_recs = nullptr;
	uint16_t *h_floor = nullptr;
	float *h_res = nullptr;
	float *h_fcurve = nullptr, *d_fcurve = nullptr; // explicit floor curves (floor 0), layout of the residues
	// packets of the generic kernels, by size class (block size <= / > 2^9): dense launch grids instead of 8192
	// workgroups that mostly find out they have nothing to do
	uint32_t *h_gen = nullptr, *d_gen = nullptr; // [3][max_packets]: small blocks, large blocks, k_ola_generic's packets
	uint32_t n_gen_small = 0, n_gen_large = 0, n_gen_ola = 0;
	LwSegment *h_seg = nullptr, *d_seg = nullptr; // workgroups of the fused small-block kernel over the overlap-add list
	uint32_t n_seg = 0;
	bool has_tdonly = false; // the specialised kernel's work list contains LW_RF_TDONLY packets
	// Tier B: codeword symbols instead of residue vectors (inverse VQ in k_residue_vq)
	bool symbols = false;
	uint32_t *h_sym = nullptr, *d_sym = nullptr, *h_sym_off = nullptr, *d_sym_off = nullptr;
	size_t sym_cap_words = 0, sym_words = 0;
	LwPacketRec *d_recs = nullptr;
	uint16_t *d_floor = nullptr;
	float *d_res = nullptr;
	float *d_decoupled = nullptr, *d_td = nullptr, *d_tap = nullptr;
	void *d_out = nullptr;
	size_t d_out_elems = 0;
	LwFastItem *h_items = nullptr, *d_items = nullptr;           // [max_packets] main pass
	LwFastItem *h_halo_items = nullptr, *d_halo_items = nullptr; // [max_packets] halo pre-pass
	float *d_halo = nullptr;
	size_t halo_cap = 0, n_items = 0, n_halo_items = 0;
	std::vector<uint32_t> fast_idx, fast_slot, fast_order;
	uint32_t fast_per_round = 1, fast_rounds = 1, fast_dense = 0, fast_late_from = 1;
	// (debug: LW_PACE_GROUP=<waves per pacing group> overrides)
	size_t n = 0, res_floats = 0, out_elems = 0;
	uint32_t max_n = 0;
	bool has_generic = false, has_fast = false, force_generic = false;
	std::vector<lw_packet_result> results;
	uint64_t alg_bytes = 0;
	std::string last_kernels;
	std::vector<lw::Prologue> prologues;
	std::vector<int> status;
	std::vector<int32_t> slot_last; // per state slot: last ok packet index in this batch (-1 none)
	std::vector<uint32_t> slot_seen; // per state slot: epoch of the batch that last touched it
	uint32_t epoch = 0;
	std::vector<lw_pwr *> touched;
};

namespace {

size_t elem_size(int fmt)
{
	return fmt == LW_FMT_F32_PLANAR ? 4 : 2;
}

int decoder_set_device(const lw_decoder *d)
{
	HIP_TRY(hipSetDevice(d->device));
	return LW_OK;
}

// grow the state pool to at least `slots` (caller holds d->mu)
int grow_state(lw_decoder *d, size_t slots)
{
	if (slots <= d->state_cap)
		return LW_OK;
	size_t cap = std::max<size_t>(slots, d->state_cap ? d->state_cap * 2 : 64);
	const size_t per = (size_t)2 * d->T.state_stride;
	float *nb = nullptr;
	HIP_TRY(hipDeviceSynchronize());
	HIP_TRY(hipMalloc((void **)&nb, cap * per * sizeof(float)));
	if (d->d_state) {
		HIP_TRY(hipMemcpy(nb, d->d_state, d->state_cap * per * sizeof(float), hipMemcpyDeviceToDevice));
		(void)hipFree(d->d_state);
	}
	for (size_t s = cap; s-- > d->state_cap;)
		d->free_slots.push_back((int)s);
	d->d_state = nb;
	d->state_cap = cap;
	return LW_OK;
}

} // namespace

extern "C" {

const char *lw_version(void)
{
	return "lewton_amd 0.1 (gfx950)";
}

const char *lw_last_device_error(void)
{
	return g_dev_err.c_str();
}

// ---- headers ----------------------------------------------------------------------------------
lw_ident *lw_read_header_ident(const uint8_t *packet, size_t len, int *err)
{
	int e = 0;
	if (!packet && len) { // (NULL, 0) is an empty packet: the reader fails on its first bit like the reference's
		if (err)
			*err = LW_ERR_NULL_ARG;
		return nullptr;
	}
	auto p = lw::read_header_ident(packet, len, e);
	if (err)
		*err = e;
	if (!p)
		return nullptr;
	auto *h = new lw_ident;
	h->p = std::move(p);
	return h;
}

int lw_ident_get_info(const lw_ident *id, lw_ident_info *out)
{
	if (!id || !out)
		return LW_ERR_NULL_ARG;
	out->audio_channels = id->p->channels;
	out->audio_sample_rate = id->p->sample_rate;
	out->bitrate_maximum = id->p->br_max;
	out->bitrate_nominal = id->p->br_nom;
	out->bitrate_minimum = id->p->br_min;
	out->blocksize_0 = id->p->bs0;
	out->blocksize_1 = id->p->bs1;
	return LW_OK;
}

void lw_ident_free(lw_ident *id)
{
	delete id;
}

lw_setup *lw_read_header_setup(const uint8_t *packet, size_t len, uint8_t ch, uint8_t bs0, uint8_t bs1, int *err)
{
	int e = 0;
	if (!packet && len) { // (NULL, 0) is an empty packet: the reader fails on its first bit like the reference's
		if (err)
			*err = LW_ERR_NULL_ARG;
		return nullptr;
	}
	auto p = lw::read_header_setup(packet, len, ch, bs0, bs1, e);
	if (err)
		*err = e;
	if (!p)
		return nullptr;
	auto *h = new lw_setup;
	h->p = std::move(p);
	return h;
}

void lw_setup_free(lw_setup *s)
{
	delete s;
}

lw_comment *lw_read_header_comment(const uint8_t *packet, size_t len, int *err)
{
	int e = 0;
	if (!packet && len) { // (NULL, 0) is an empty packet: the reader fails on its first bit like the reference's
		if (err)
			*err = LW_ERR_NULL_ARG;
		return nullptr;
	}
	auto p = lw::read_header_comment(packet, len, e);
	if (err)
		*err = e;
	if (!p)
		return nullptr;
	auto *h = new lw_comment;
	h->p = std::move(p);
	return h;
}

const char *lw_comment_vendor(const lw_comment *c, size_t *len)
{
	if (len)
		*len = c->p->vendor.size();
	return c->p->vendor.data();
}

size_t lw_comment_count(const lw_comment *c)
{
	return c->p->list.size();
}

int lw_comment_get(const lw_comment *c, size_t i, const char **key, size_t *key_len, const char **val, size_t *val_len)
{
	if (!c || i >= c->p->list.size())
		return LW_ERR_CAPACITY;
	*key = c->p->list[i].first.data();
	*key_len = c->p->list[i].first.size();
	*val = c->p->list[i].second.data();
	*val_len = c->p->list[i].second.size();
	return LW_OK;
}

void lw_comment_free(lw_comment *c)
{
	delete c;
}

// a setup header is parsed for a channel count (header.rs:1029: the mux lists have one entry per channel); pairing it with
// another stream's ident header would index those lists out of range
static bool setup_matches_ident(const lw::Ident &id, const lw::Setup &s)
{
	for (const auto &m : s.mappings)
		if (m.mux.size() != id.channels)
			return false;
	return true;
}

int lw_get_decoded_sample_count(const lw_ident *id, const lw_setup *s, const uint8_t *packet, size_t len, size_t *count)
{
	if (!id || !s || (!packet && len) || !count)
		return LW_ERR_NULL_ARG;
	if (!setup_matches_ident(*id->p, *s->p))
		return LW_ERR_STATE_MISMATCH;
	return lw::decoded_sample_count(*id->p, *s->p, packet, len, *count);
}

static uint32_t floor_stride_of(const lw::Setup &s)
{
	uint32_t mp = 2;
	for (const auto &fl : s.floors)
		if (fl.type == 1)
			mp = std::max<uint32_t>(mp, (uint32_t)fl.f1.x_list.size());
	return (mp + 1) & ~1u;
}

uint32_t lw_setup_floor_stride(const lw_setup *s)
{
	return s ? floor_stride_of(*s->p) : 0;
}

int lw_entropy_decode_host(const lw_ident *id, const lw_setup *s, const uint8_t *packet, size_t len, uint16_t *floor_out,
		float *residue_out, size_t residue_cap_floats, uint8_t *blocksize_log2, uint8_t *mode, uint8_t *flags,
		uint64_t *bits_consumed, float *floor_curve_out)
{
	if (!id || !s || (!packet && len) || !floor_out || !residue_out)
		return LW_ERR_NULL_ARG;
	if (!setup_matches_ident(*id->p, *s->p))
		return LW_ERR_STATE_MISMATCH;
	lw::BitReader br(packet, len);
	lw::Prologue p;
	int rc = lw::read_prologue(*id->p, *s->p, br, p);
	if (rc)
		return rc;
	if ((size_t)id->p->channels * (p.n / 2) > residue_cap_floats)
		return LW_ERR_CAPACITY;
	lw::EntropyScratch scr;
	rc = lw::entropy_decode(*id->p, *s->p, packet, len, p, floor_out, floor_stride_of(*s->p), residue_out, scr,
			bits_consumed, floor_curve_out);
	if (blocksize_log2)
		*blocksize_log2 = p.bs;
	if (mode)
		*mode = p.mode;
	if (flags)
		*flags = (uint8_t)((p.blockflag ? 1 : 0) | (p.prev_flag ? 2 : 0) | (p.next_flag ? 4 : 0));
	return rc;
}

int lw_setup_supports_device_vq(const lw_ident *id, const lw_setup *s, const char **why)
{
	static thread_local std::string msg;
	const char *w = "";
	const bool ok = id && s && lw::symbols_supported(*id->p, *s->p, &w);
	msg = w;
	if (why)
		*why = msg.c_str();
	return ok ? 1 : 0;
}

int lw_entropy_symbols_host(const lw_ident *id, const lw_setup *s, const uint8_t *packet, size_t len, uint16_t *floor_out,
		uint64_t *symbols, size_t cap_symbols, size_t *n_symbols, uint32_t pass_off[9], uint8_t *blocksize_log2,
		uint8_t *mode, uint8_t *flags, float *floor_curve_out)
{
	if (!id || !s || (!packet && len) || !floor_out || !symbols || !n_symbols || !pass_off)
		return LW_ERR_NULL_ARG;
	if (!setup_matches_ident(*id->p, *s->p))
		return LW_ERR_STATE_MISMATCH;
	if (!lw::symbols_supported(*id->p, *s->p, nullptr))
		return LW_ERR_UNSUPPORTED;
	lw::Prologue p;
	lw::EntropyScratch scr;
	lw::SymbolSink sink;
	std::vector<uint64_t> tmp;
	sink.clear();
	const int rc = lw::entropy_decode(*id->p, *s->p, packet, len, p, floor_out, floor_stride_of(*s->p), nullptr, scr, nullptr,
			floor_curve_out, &sink);
	if (blocksize_log2)
		*blocksize_log2 = p.bs;
	if (mode)
		*mode = p.mode;
	if (flags)
		*flags = (uint8_t)((p.blockflag ? 1 : 0) | (p.prev_flag ? 2 : 0) | (p.next_flag ? 4 : 0));
	if (rc)
		return rc;
	sink.sort_by_pass(tmp);
	*n_symbols = sink.ops.size();
	for (int q = 0; q < 9; q++)
		pass_off[q] = sink.pass_off[q];
	if (sink.ops.size() > cap_symbols)
		return LW_ERR_CAPACITY;
	if (!sink.ops.empty())
		std::memcpy(symbols, sink.ops.data(), sink.ops.size() * 8);
	return LW_OK;
}

int lw_setup_codebook_vq(const lw_setup *s, unsigned book, float *dst, size_t cap_floats, uint32_t *dims, uint32_t *entries)
{
	if (!s || book >= s->p->codebooks.size())
		return LW_ERR_NULL_ARG;
	const lw::Codebook &cb = s->p->codebooks[book];
	if (dims)
		*dims = cb.dims;
	if (entries)
		*entries = cb.entries;
	if (!cb.has_vq)
		return LW_ERR_UNSUPPORTED;
	if (dst) {
		if (cap_floats < cb.vq.size())
			return LW_ERR_CAPACITY;
		std::memcpy(dst, cb.vq.data(), cb.vq.size() * sizeof(float));
	}
	return LW_OK;
}

int lw_setup_submap_info(const lw_setup *s, unsigned mode, unsigned submap, uint8_t *residue_type, uint32_t *partition_size,
		uint8_t *channels, size_t cap_channels, size_t *n_channels)
{
	if (!s || mode >= s->p->modes.size())
		return LW_ERR_NULL_ARG;
	const lw::Mapping &mp = s->p->mappings[s->p->modes[mode].mapping];
	if (submap >= mp.submap_residue.size())
		return LW_ERR_NULL_ARG;
	const lw::Residue &rs = s->p->residues[mp.submap_residue[submap]];
	if (residue_type)
		*residue_type = rs.type;
	if (partition_size)
		*partition_size = rs.partition_size;
	size_t n = 0;
	for (size_t c = 0; c < mp.mux.size(); c++)
		if (mp.mux[c] == submap) {
			if (channels && n < cap_channels)
				channels[n] = (uint8_t)c;
			n++;
		}
	if (n_channels)
		*n_channels = n;
	return channels && n > cap_channels ? LW_ERR_CAPACITY : LW_OK;
}

size_t lw_debug_fast_image(const lw_ident *id, const lw_setup *s, uint8_t *dst, size_t cap, uint32_t *offsets16)
{
	if (!id || !s)
		return 0;
	LwFastPlan plan;
	lw::build_fast_plan(*id->p, *s->p, plan);
	if (!plan.eligible)
		return 0;
	std::memcpy(plan.image.data() + plan.off.inv_db, kInverseDbTable, sizeof(float) * 256);
	if (dst)
		std::memcpy(dst, plan.image.data(), std::min(cap, plan.image.size()));
	if (offsets16)
		std::memcpy(offsets16, &plan.off, sizeof(uint32_t) * 16);
	return plan.image.size();
}

int lw_huffman_check(const uint8_t *lengths, size_t n_entries, const uint8_t *bits, size_t bits_len, uint32_t *syms,
		size_t max_syms, size_t *n_syms)
{
	lw::Huffman h;
	const int rc = (int)h.build(lengths, n_entries);
	if (rc)
		return rc;
	if (bits && syms && n_syms) {
		// through the residue loops' reader: table path while 8 bytes lie ahead, Huffman::decode for the rest
		lw::BitReader r(bits, bits_len);
		lw::CodeReader cr(r);
		size_t k = 0;
		while (k < max_syms) {
			uint32_t sym;
			if (!cr.next(h, sym))
				break;
			syms[k++] = sym;
		}
		*n_syms = k;
	}
	return 0;
}

// ---- device context -------------------------------------------------------------------------
int lw_device_count(void)
{
	int n = 0;
	if (!hip_ok(hipGetDeviceCount(&n), "hipGetDeviceCount"))
		return 0;
	return n;
}

lw_decoder *lw_decoder_create(const lw_ident *idh, const lw_setup *sh, int device, int *err)
{
	int dummy;
	if (!err)
		err = &dummy;
	*err = LW_OK;
	if (!idh || !sh) {
		*err = LW_ERR_NULL_ARG;
		return nullptr;
	}
	const lw::Ident &id = *idh->p;
	const lw::Setup &s = *sh->p;
	if (!setup_matches_ident(id, s)) {
		*err = LW_ERR_STATE_MISMATCH;
		return nullptr;
	}
	int ndev = 0;
	if (!hip_ok(hipGetDeviceCount(&ndev), "hipGetDeviceCount") || device < 0 || device >= ndev ||
			!hip_ok(hipSetDevice(device), "hipSetDevice")) {
		if (g_dev_err.empty())
			g_dev_err = "no such HIP device";
		*err = LW_ERR_DEVICE;
		return nullptr;
	}
	auto d = std::make_unique<lw_decoder>();
	d->id = idh->p;
	d->setup = sh->p;
	d->device = device;
	if (hipDeviceGetAttribute(&d->n_cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || d->n_cus <= 0)
		d->n_cus = 256;

	// ---- build one blob with all tables
	std::vector<uint8_t> blob;
	auto put = [&](const void *p, size_t bytes) {
		const size_t off = (blob.size() + 255) & ~(size_t)255;
		blob.resize(off + bytes);
		std::memcpy(blob.data() + off, p, bytes);
		return off;
	};
	size_t offA[2], offB[2], offC[2], offW[2], offR[2];
	for (int b = 0; b < 2; b++) {
		const lw::BlocksizeTables &t = id.tab[b];
		offA[b] = put(t.A.data(), t.A.size() * 4);
		offB[b] = put(t.B.data(), t.B.size() * 4);
		offC[b] = put(t.C.data(), t.C.size() * 4);
		offW[b] = put(t.window.data(), t.window.size() * 4);
		offR[b] = put(t.bitrev.data(), t.bitrev.size() * 4);
	}
	const size_t off_db = put(kInverseDbTable, sizeof(float) * 256);
	const size_t nfl = s.floors.size(), nmodes = s.modes.size(), ch = id.channels;
	d->mode_floor_bytes.assign(nmodes, 0);
	for (size_t m = 0; m < nmodes; m++) {
		const lw::Mapping &mp = s.mappings[s.modes[m].mapping];
		const uint64_t half = ((uint64_t)1 << (s.modes[m].blockflag ? id.bs1 : id.bs0)) / 2;
		for (size_t c = 0; c < ch; c++) {
			const lw::Floor &fl = s.floors[mp.submap_floor[mp.mux[c]]];
			d->mode_floor_bytes[m] += fl.type == 0 ? half * 4 + 2 : (uint64_t)fl.f1.x_list.size() * 2; // explicit curve | posts
		}
	}
	std::vector<uint16_t> fx(nfl * LW_XSTRIDE, 0);
	std::vector<uint8_t> fF(nfl, 0);
	for (size_t f = 0; f < nfl; f++) {
		if (s.floors[f].type == 0)
			d->any_floor0 = true;
		const lw::Floor1 &f1 = s.floors[f].f1;
		fF[f] = (uint8_t)f1.sorted_x.size();
		d->max_posts = std::max<uint32_t>(d->max_posts, (uint32_t)f1.sorted_x.size());
		for (size_t i = 0; i < f1.sorted_x.size(); i++)
			fx[f * LW_XSTRIDE + i] = (uint16_t)std::min<uint32_t>(f1.sorted_x[i], 65535u);
	}
	std::vector<uint8_t> mode_floor(nmodes * ch, 0);
	std::vector<uint16_t> couple_off(nmodes + 1, 0);
	std::vector<uint8_t> couple;
	for (size_t m = 0; m < nmodes; m++) {
		const lw::Mapping &mp = s.mappings[s.modes[m].mapping];
		for (size_t c = 0; c < ch; c++)
			mode_floor[m * ch + c] = mp.submap_floor[mp.mux[c]];
		couple_off[m] = (uint16_t)(couple.size() / 2);
		for (size_t k = 0; k < mp.mag.size(); k++) {
			couple.push_back(mp.mag[k]);
			couple.push_back(mp.ang[k]);
		}
		if (!mp.mag.empty())
			d->any_coupling = true;
	}
	couple_off[nmodes] = (uint16_t)(couple.size() / 2);
	if (couple.empty())
		couple.push_back(0);
	// per mode and channel: the other channel of the one coupling step it takes part in (fused small-block kernel: a wave
	// decouples its own channel on the fly); pair_coupling = no channel of any mode is in more than one step
	std::vector<int8_t> mode_partner(nmodes * ch, -1);
	std::vector<uint8_t> mode_role(nmodes * ch, 0);
	bool pair_coupling = ch <= 127;
	for (size_t m = 0; m < nmodes && pair_coupling; m++) {
		const lw::Mapping &mp = s.mappings[s.modes[m].mapping];
		for (size_t k = 0; k < mp.mag.size(); k++) {
			const size_t mg = mp.mag[k], an = mp.ang[k];
			if (mg == an || mode_partner[m * ch + mg] >= 0 || mode_partner[m * ch + an] >= 0) {
				pair_coupling = false;
				break;
			}
			mode_partner[m * ch + mg] = (int8_t)an;
			mode_role[m * ch + mg] = 1;
			mode_partner[m * ch + an] = (int8_t)mg;
			mode_role[m * ch + an] = 2;
		}
	}
	const size_t off_fx = put(fx.data(), fx.size() * 2);
	const size_t off_fF = put(fF.data(), fF.size());
	const size_t off_mf = put(mode_floor.data(), mode_floor.size());
	const size_t off_co = put(couple_off.data(), couple_off.size() * 2);
	const size_t off_cp = put(couple.data(), couple.size());
	const size_t off_mp = put(mode_partner.data(), mode_partner.size());
	const size_t off_mr = put(mode_role.data(), mode_role.size());

	if (!hip_ok(hipMalloc(&d->d_blob, blob.size()), "hipMalloc(tables)") ||
			!hip_ok(hipMemcpy(d->d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice), "hipMemcpy(tables)")) {
		*err = LW_ERR_DEVICE;
		if (d->d_blob)
			(void)hipFree(d->d_blob);
		return nullptr;
	}
	const uint8_t *base = (const uint8_t *)d->d_blob;
	for (int b = 0; b < 2; b++) {
		d->T.bs[b].A = (const float *)(base + offA[b]);
		d->T.bs[b].B = (const float *)(base + offB[b]);
		d->T.bs[b].C = (const float *)(base + offC[b]);
		d->T.bs[b].window = (const float *)(base + offW[b]);
		d->T.bs[b].bitrev = (const uint32_t *)(base + offR[b]);
		d->T.bs[b].bs = b ? id.bs1 : id.bs0;
		d->T.bs[b].n = 1u << d->T.bs[b].bs;
	}
	d->T.inv_db = (const float *)(base + off_db);
	d->T.floor_x = (const uint16_t *)(base + off_fx);
	d->T.floor_F = base + off_fF;
	d->T.mode_floor = base + off_mf;
	d->T.couple_off = (const uint16_t *)(base + off_co);
	d->T.couple = base + off_cp;
	d->T.mode_partner = (const int8_t *)(base + off_mp);
	d->T.mode_role = base + off_mr;
	d->T.pair_coupling = pair_coupling ? 1u : 0u;
	d->T.sid = nullptr;
	d->T.ch = (uint32_t)ch;
	d->T.fstride = floor_stride_of(s);
	d->T.n_modes = (uint32_t)nmodes;
	d->T.n_floors = (uint32_t)nfl;
	d->T.state_chan_stride = (1u << id.bs1) / 2;
	d->T.state_stride = d->T.ch * d->T.state_chan_stride;
	// ---- Tier B tables: dense VQ tables, per-mode submap descriptors
	{
		const char *why = "";
		d->symbols_ok = lw::symbols_supported(id, s, &why);
		d->symbols_why = why;
		if (d->symbols_ok) {
			std::vector<uint8_t> vb;
			auto putv = [&](const void *p, size_t bytes) {
				const size_t off = (vb.size() + 255) & ~(size_t)255;
				vb.resize(off + bytes);
				std::memcpy(vb.data() + off, p, bytes);
				return off;
			};
			// the tables of the books used by residues, smallest first: k_residue_vq stages a prefix of the pool in LDS (the
			// gathers of 8..32 bytes out of 128-byte L2 lines are what bounds that kernel otherwise)
			std::vector<float> pool;
			std::vector<uint32_t> boff(256, 0);
			std::vector<uint16_t> bdims(256, 0);
			std::vector<bool> in_residue(s.codebooks.size(), false);
			for (const lw::Residue &rs : s.residues)
				for (const lw::ResidueBook &rb : rs.books)
					for (unsigned pass = 0; pass < 8; pass++)
						if (rb.vals_used & (1u << pass))
							in_residue[rb.val_i[pass]] = true;
			std::vector<size_t> order;
			for (size_t k = 0; k < s.codebooks.size(); k++) {
				bdims[k] = s.codebooks[k].dims;
				if (s.codebooks[k].has_vq && in_residue[k])
					order.push_back(k);
			}
			std::stable_sort(order.begin(), order.end(),
					[&](size_t a, size_t b) { return s.codebooks[a].vq.size() < s.codebooks[b].vq.size(); });
			d->vq_book_ends.clear();
			for (size_t k : order) {
				const lw::Codebook &cb = s.codebooks[k];
				pool.resize((pool.size() + 3) & ~(size_t)3); // rows of 2 / 4 / 8 floats stay 8 / 16 / 32-byte aligned
				boff[k] = (uint32_t)pool.size();
				pool.insert(pool.end(), cb.vq.begin(), cb.vq.end());
				d->vq_book_ends.push_back((uint32_t)pool.size());
			}
			if (pool.empty())
				pool.push_back(0.0f);
			std::vector<LwSubmapDesc> sd(nmodes * 16);
			std::vector<LwChanMap> cm(nmodes * ch);
			for (size_t m = 0; m < nmodes; m++) {
				const lw::Mapping &mp = s.mappings[s.modes[m].mapping];
				unsigned before = 0;
				for (size_t sm = 0; sm < mp.submap_residue.size(); sm++) {
					const lw::Residue &rs = s.residues[mp.submap_residue[sm]];
					LwSubmapDesc &e = sd[m * 16 + sm];
					e.type = rs.type;
					e.psize = (uint16_t)rs.partition_size;
					e.vbase_ch = (uint16_t)before;
					e.pad = 0;
					uint8_t n = 0;
					for (size_t c = 0; c < ch; c++)
						if (mp.mux[c] == sm)
							n++;
					e.sub_ch = n;
					uint8_t pos = 0;
					for (size_t c = 0; c < ch; c++)
						if (mp.mux[c] == sm)
							cm[m * ch + c] = LwChanMap{(uint8_t)before, n, pos++, rs.type};
					before += n;
				}
			}
			const size_t o_vq = putv(pool.data(), pool.size() * 4), o_bo = putv(boff.data(), boff.size() * 4);
			const size_t o_bd = putv(bdims.data(), bdims.size() * 2), o_sd = putv(sd.data(), sd.size() * sizeof(LwSubmapDesc));
			const size_t o_ch = putv(cm.data(), cm.size() * sizeof(LwChanMap));
			if (!hip_ok(hipMalloc(&d->d_vq_blob, vb.size()), "hipMalloc(vq tables)") ||
					!hip_ok(hipMemcpy(d->d_vq_blob, vb.data(), vb.size(), hipMemcpyHostToDevice), "hipMemcpy(vq tables)")) {
				*err = LW_ERR_DEVICE;
				(void)hipFree(d->d_blob);
				return nullptr;
			}
			const uint8_t *vbase = (const uint8_t *)d->d_vq_blob;
			d->V.vq = (const float *)(vbase + o_vq);
			d->V.book_off = (const uint32_t *)(vbase + o_bo);
			d->V.book_dims = (const uint16_t *)(vbase + o_bd);
			d->V.submap = (const LwSubmapDesc *)(vbase + o_sd);
			d->V.chmap = (const LwChanMap *)(vbase + o_ch);
		}
	}
	lw::build_fast_plan(id, s, d->fast);
	if (d->fast.eligible) {
		std::memcpy(d->fast.image.data() + d->fast.off.inv_db, kInverseDbTable, sizeof(float) * 256);
		const size_t ub = d->fast.units.size() * sizeof(LwFastUnit);
		if (!hip_ok(hipMalloc((void **)&d->d_fast_image, d->fast.image.size()), "hipMalloc(fast image)") ||
				!hip_ok(hipMemcpy(d->d_fast_image, d->fast.image.data(), d->fast.image.size(), hipMemcpyHostToDevice),
					"hipMemcpy(fast image)") ||
				!hip_ok(hipMalloc((void **)&d->d_fast_units, ub), "hipMalloc(fast units)") ||
				!hip_ok(hipMemcpy(d->d_fast_units, d->fast.units.data(), ub, hipMemcpyHostToDevice), "hipMemcpy(fast units)")) {
			*err = LW_ERR_DEVICE;
			(void)hipFree(d->d_blob);
			return nullptr;
		}
	}
	return d.release();
}

void lw_decoder_destroy(lw_decoder *d)
{
	if (!d)
		return;
	(void)hipSetDevice(d->device);
	(void)hipDeviceSynchronize();
	if (d->one)
		lw_batch_destroy(d->one);
	if (d->one_out)
		(void)hipHostFree(d->one_out);
	if (d->d_state)
		(void)hipFree(d->d_state);
	if (d->d_blob)
		(void)hipFree(d->d_blob);
	if (d->d_vq_blob)
		(void)hipFree(d->d_vq_blob);
	if (d->d_fast_image)
		(void)hipFree(d->d_fast_image);
	if (d->d_fast_units)
		(void)hipFree(d->d_fast_units);
	delete d;
}

int lw_decoder_device(const lw_decoder *d)
{
	return d ? d->device : -1;
}

size_t lw_decoder_max_block_elems(const lw_decoder *d)
{
	// a long block with a long predecessor and a short successor yields the most samples: right_start - left_start =
	// (3 n1 - n0) / 4 (audio.rs:1056-1073)
	if (!d)
		return 0;
	const size_t n0 = (size_t)1 << d->id->bs0, n1 = (size_t)1 << d->id->bs1;
	return (size_t)d->T.ch * ((3 * n1 - n0) / 4);
}

// ---- PreviousWindowRight ----------------------------------------------------------------------
lw_pwr *lw_pwr_new(lw_decoder *d)
{
	if (!d)
		return nullptr;
	std::lock_guard<std::mutex> g(d->mu);
	if (hipSetDevice(d->device) != hipSuccess)
		return nullptr;
	if (d->free_slots.empty() && grow_state(d, d->state_cap + 1) != LW_OK)
		return nullptr;
	auto *p = new lw_pwr;
	p->dec = d;
	p->slot = d->free_slots.back();
	d->free_slots.pop_back();
	return p;
}

int lw_pwr_is_empty(const lw_pwr *p)
{
	return p ? !p->present : 1;
}

void lw_pwr_reset(lw_pwr *p)
{
	if (p) {
		p->present = false;
		p->len = 0;
	}
}

size_t lw_pwr_len(const lw_pwr *p)
{
	return (p && p->present) ? p->len : 0;
}

void lw_pwr_get_state(const lw_pwr *p, lw_pwr_state *out)
{
	if (!p || !out)
		return;
	out->present = p->present ? 1 : 0;
	out->parity = p->parity;
	out->len = p->len;
}

void lw_pwr_set_state(lw_pwr *p, const lw_pwr_state *in)
{
	if (!p || !in)
		return;
	p->present = in->present != 0;
	p->parity = in->parity & 1;
	p->len = in->len;
}

lw_pwr *lw_pwr_clone(const lw_pwr *p)
{
	if (!p)
		return nullptr;
	lw_pwr *q = lw_pwr_new(p->dec);
	if (!q)
		return nullptr;
	q->present = p->present;
	q->len = p->len;
	q->parity = p->parity;
	if (p->present) {
		lw_decoder *d = p->dec;
		std::lock_guard<std::mutex> g(d->mu);
		const size_t per = (size_t)2 * d->T.state_stride;
		if (!hip_ok(hipDeviceSynchronize(), "sync") ||
				!hip_ok(hipMemcpy(d->d_state + (size_t)q->slot * per, d->d_state + (size_t)p->slot * per,
							per * sizeof(float), hipMemcpyDeviceToDevice),
					"hipMemcpy(state clone)")) {
			q->present = false;
		}
	}
	return q;
}

void lw_pwr_free(lw_pwr *p)
{
	if (!p)
		return;
	{
		std::lock_guard<std::mutex> g(p->dec->mu);
		p->dec->free_slots.push_back(p->slot);
	}
	delete p;
}

int lw_pwr_copy_to_host(const lw_pwr *p, float *dst)
{
	if (!p || !dst)
		return LW_ERR_NULL_ARG;
	if (!p->present)
		return LW_ERR_CAPACITY;
	lw_decoder *d = p->dec;
	if (int rc = decoder_set_device(d))
		return rc;
	HIP_TRY(hipDeviceSynchronize());
	const float *src = d->d_state + ((size_t)p->slot * 2 + p->parity) * d->T.state_stride;
	HIP_TRY(hipMemcpy2D(dst, p->len * sizeof(float), src, d->T.state_chan_stride * sizeof(float), p->len * sizeof(float),
				d->T.ch, hipMemcpyDeviceToHost));
	return LW_OK;
}

// ---- batches ----------------------------------------------------------------------------------
lw_batch *lw_batch_create(lw_decoder *d, size_t max_packets, int fmt, int *err)
{
	int dummy;
	if (!err)
		err = &dummy;
	*err = LW_OK;
	if (!d || max_packets == 0 || fmt < 0 || fmt > 2) {
		*err = LW_ERR_NULL_ARG;
		return nullptr;
	}
	if (decoder_set_device(d)) {
		*err = LW_ERR_DEVICE;
		return nullptr;
	}
	auto b = std::make_unique<lw_batch>();
	b->dec = d;
	b->max_packets = max_packets;
	b->fmt = fmt;
	const size_t ch = d->T.ch, half1 = d->T.state_chan_stride;
	const size_t rec_b = max_packets * sizeof(LwPacketRec);
	const size_t fl_b = max_packets * ch * d->T.fstride * sizeof(uint16_t);
	const size_t res_b = max_packets * ch * half1 * sizeof(float);
	// every host->device buffer of the batch is a slice of ONE pinned slab mirrored by ONE device slab at the same
	// offsets: a small batch (the single-packet path of lw_read_audio_packet above all) goes up with one hipMemcpyAsync
	// instead of five
	size_t off = 0;
	auto slice = [&](size_t bytes) {
		const size_t at = off;
		off = (off + bytes + 255) & ~(size_t)255;
		return at;
	};
	const size_t o_recs = slice(rec_b), o_floor = slice(fl_b), o_items = slice(max_packets * sizeof(LwFastItem));
	const size_t o_halo = slice(max_packets * sizeof(LwFastItem)), o_gen = slice(3 * max_packets * sizeof(uint32_t));
	const size_t o_seg = slice(max_packets * sizeof(LwSegment));
	const size_t o_res = slice(res_b), o_fc = d->any_floor0 ? slice(res_b) : 0;
	b->slab_bytes = off;
	bool ok = hip_ok(hipHostMalloc((void **)&b->h_slab, off), "hipHostMalloc(batch records)") &&
		hip_ok(hipMalloc((void **)&b->d_slab, off), "hipMalloc(batch records)");
	if (ok) {
		auto H = [&](size_t o) { return b->h_slab + o; };
		auto D = [&](size_t o) { return b->d_slab + o; };
		b->h_recs = (LwPacketRec *)H(o_recs), b->d_recs = (LwPacketRec *)D(o_recs);
		b->h_floor = (uint16_t *)H(o_floor), b->d_floor = (uint16_t *)D(o_floor);
		b->h_items = (LwFastItem *)H(o_items), b->d_items = (LwFastItem *)D(o_items);
		b->h_halo_items = (LwFastItem *)H(o_halo), b->d_halo_items = (LwFastItem *)D(o_halo);
		b->h_gen = (uint32_t *)H(o_gen), b->d_gen = (uint32_t *)D(o_gen);
		b->h_seg = (LwSegment *)H(o_seg), b->d_seg = (LwSegment *)D(o_seg);
		b->h_res = (float *)H(o_res), b->d_res = (float *)D(o_res);
		if (d->any_floor0)
			b->h_fcurve = (float *)H(o_fc), b->d_fcurve = (float *)D(o_fc);
	}
	if (!ok) {
		*err = LW_ERR_DEVICE;
		lw_batch_destroy(b.release());
		return nullptr;
	}
	b->results.resize(max_packets);
	b->prologues.resize(max_packets);
	b->status.resize(max_packets);
	return b.release();
}

void lw_batch_destroy(lw_batch *b)
{
	if (!b)
		return;
	(void)hipSetDevice(b->dec->device);
	(void)hipDeviceSynchronize();
	if (b->h_slab)
		(void)hipHostFree(b->h_slab);
	if (b->h_sym)
		(void)hipHostFree(b->h_sym);
	if (b->h_sym_off)
		(void)hipHostFree(b->h_sym_off);
	void *dev[] = {b->d_slab, b->d_sym, b->d_sym_off, b->d_decoupled, b->d_td, b->d_tap, b->d_out, b->d_halo};
	for (void *p : dev)
		if (p)
			(void)hipFree(p);
	delete b;
}

void lw_batch_set_force_generic(lw_batch *b, int on)
{
	if (b)
		b->force_generic = on != 0;
}

int lw_decoder_supports_device_vq(const lw_decoder *d, const char **why)
{
	if (why)
		*why = d ? d->symbols_why.c_str() : "";
	return d && d->symbols_ok ? 1 : 0;
}

int lw_batch_set_residue_on_device(lw_batch *b, int on)
{
	if (!b)
		return LW_ERR_NULL_ARG;
	if (!on) {
		b->symbols = false;
		return LW_OK;
	}
	if (!b->dec->symbols_ok)
		return LW_ERR_UNSUPPORTED;
	if (int rc = decoder_set_device(b->dec))
		return rc;
	if (!b->h_sym_off) {
		if (!hip_ok(hipHostMalloc((void **)&b->h_sym_off, b->max_packets * sizeof(uint32_t)), "hipHostMalloc(symbol offsets)") ||
				!hip_ok(hipMalloc((void **)&b->d_sym_off, b->max_packets * sizeof(uint32_t)), "hipMalloc(symbol offsets)"))
			return LW_ERR_DEVICE;
	}
	b->symbols = true;
	return LW_OK;
}

size_t lw_batch_size(const lw_batch *b)
{
	return b ? b->n : 0;
}

size_t lw_batch_out_elems(const lw_batch *b)
{
	return b ? b->out_elems : 0;
}

const lw_packet_result *lw_batch_results(const lw_batch *b)
{
	return b ? b->results.data() : nullptr;
}

uint64_t lw_batch_algorithmic_bytes(const lw_batch *b)
{
	return b ? b->alg_bytes : 0;
}

const char *lw_batch_last_kernels(const lw_batch *b)
{
	return b ? b->last_kernels.c_str() : "";
}

int lw_batch_entropy(lw_batch *b, const lw_packet *pkts, size_t n, int n_threads)
{
	if (!b || (!pkts && n))
		return LW_ERR_NULL_ARG;
	if (n > b->max_packets)
		return LW_ERR_CAPACITY;
	lw_decoder *d = b->dec;
	const lw::Ident &id = *d->id;
	const lw::Setup &s = *d->setup;
	const size_t ch = d->T.ch, fstride = d->T.fstride;
	b->n = n;

	// pass 1 (sequential, cheap): prologues -> block sizes -> residue offsets
	size_t res_off = 0;
	uint32_t max_n = 0;
	for (size_t i = 0; i < n; i++) {
		LwPacketRec &r = b->h_recs[i];
		std::memset(&r, 0, sizeof(r));
		r.prev = -1;
		r.state_out = -1;
		r.floor_off = (uint32_t)(i * ch * fstride);
		r.res_off = (uint32_t)res_off;
		if ((!pkts[i].data && pkts[i].len) || !pkts[i].pwr) {
			b->status[i] = LW_ERR_NULL_ARG;
			continue;
		}
		if (pkts[i].pwr->dec != d) {
			b->status[i] = LW_ERR_STATE_MISMATCH;
			continue;
		}
		lw::BitReader br(pkts[i].data, pkts[i].len);
		b->status[i] = lw::read_prologue(id, s, br, b->prologues[i]);
		if (b->status[i] == LW_OK) {
			res_off += ch * (b->prologues[i].n / 2);
			max_n = std::max(max_n, b->prologues[i].n);
		}
	}
	b->res_floats = res_off;
	b->max_n = max_n;

	// pass 2 (parallel): entropy decode straight into the pinned staging buffers
	unsigned nt = n_threads > 0 ? (unsigned)n_threads : std::max(1u, std::thread::hardware_concurrency());
	nt = (unsigned)std::min<size_t>(nt, std::max<size_t>(1, n / 8));
	alignas(128) std::atomic<size_t> next{0}; // (its own cache line: every worker adds to it once per LW_ENTROPY_CHUNK packets)
	alignas(128) char next_pad[8] = {0};
	(void)next_pad;
	// Tier B: a worker reserves room for a packet's symbol block in the pinned pool with one atomic add once the packet
	// is decoded (block sizes are not known before).  Blocks that no longer fit are parked in the worker's own arena and
	// gathered after the pool has been enlarged (first batches only).
	struct SymRef {
		int32_t arena = -1; // -1: already in the pool at h_sym_off[i]
		uint32_t off = 0, words = 0;
	};
	std::vector<SymRef> sym_ref(b->symbols ? n : 0);
	std::vector<std::vector<uint32_t>> arenas(b->symbols ? std::max(1u, nt) : 0);
	std::atomic<unsigned> next_arena{0};
	std::atomic<size_t> pool_used{0};
	std::atomic<bool> overflow{false};
	auto worker = [&]() {
		// scratch vectors keep their capacity from batch to batch (pool threads are persistent)
		static thread_local lw::EntropyScratch scr;
		static thread_local lw::SymbolSink sink;
		static thread_local std::vector<uint64_t> tmp;
		const unsigned my = b->symbols ? next_arena.fetch_add(1) : 0;
		for (;;) {
			const size_t i0 = next.fetch_add(LW_ENTROPY_CHUNK);
			if (i0 >= n)
				break;
			for (size_t i = i0; i < std::min(n, i0 + LW_ENTROPY_CHUNK); i++) {
				if (b->status[i] != LW_OK)
					continue;
				LwPacketRec &r = b->h_recs[i];
				if (b->symbols)
					sink.clear();
				b->status[i] = lw::entropy_decode(id, s, pkts[i].data, pkts[i].len, b->prologues[i],
						b->h_floor + r.floor_off, (unsigned)fstride, b->h_res + r.res_off, scr, nullptr,
						b->h_fcurve ? b->h_fcurve + r.res_off : nullptr, b->symbols ? &sink : nullptr);
				if (b->symbols && b->status[i] == LW_OK) {
					sink.sort_by_pass(tmp);
					SymRef &ref = sym_ref[i];
					ref.words = 10 + 2 * (uint32_t)sink.ops.size();
					const size_t at = pool_used.fetch_add(ref.words);
					uint32_t *w;
					if (at + ref.words <= b->sym_cap_words) {
						b->h_sym_off[i] = (uint32_t)at;
						w = b->h_sym + at;
					} else {
						overflow = true;
						std::vector<uint32_t> &a = arenas[my];
						ref.arena = (int32_t)my;
						ref.off = (uint32_t)a.size();
						a.resize(a.size() + ref.words);
						w = a.data() + ref.off;
					}
					for (int q = 0; q < 9; q++)
						w[q] = sink.pass_off[q];
					w[9] = 0;
					if (!sink.ops.empty())
						std::memcpy(w + 10, sink.ops.data(), sink.ops.size() * 8);
				}
			}
		}
	};
	lw::entropy_pool().run(nt, worker);
	if (b->symbols) {
		const size_t total = pool_used.load();
		if (overflow) {
			// enlarge the pool, keep what is already in it, append the parked blocks
			uint32_t *nh = nullptr, *nd = nullptr;
			const size_t cap = total + total / 2 + 1024;
			if (!hip_ok(hipHostMalloc((void **)&nh, cap * 4), "hipHostMalloc(symbols)") ||
					!hip_ok(hipMalloc((void **)&nd, cap * 4), "hipMalloc(symbols)"))
				return LW_ERR_DEVICE;
			size_t at = 0;
			for (size_t i = 0; i < n; i++) {
				const SymRef &ref = sym_ref[i];
				if (!ref.words)
					continue;
				const uint32_t *src = ref.arena < 0 ? b->h_sym + b->h_sym_off[i] : arenas[ref.arena].data() + ref.off;
				std::memcpy(nh + at, src, (size_t)ref.words * 4);
				b->h_sym_off[i] = (uint32_t)at;
				at += ref.words;
			}
			if (b->h_sym)
				(void)hipHostFree(b->h_sym);
			if (b->d_sym) {
				(void)hipDeviceSynchronize();
				(void)hipFree(b->d_sym);
			}
			b->h_sym = nh;
			b->d_sym = nd;
			b->sym_cap_words = cap;
			b->sym_words = at;
		} else {
			b->sym_words = total;
		}
	}

	// pass 3 (sequential): window geometry, state hand-over, output offsets, error semantics
	if (b->slot_last.size() < d->state_cap) {
		b->slot_last.assign(d->state_cap, -1);
		b->slot_seen.assign(d->state_cap, 0);
	}
	b->epoch++;
	b->touched.clear();
	b->fast_idx.clear();
	b->fast_slot.clear();
	size_t out_off = 0;
	uint64_t alg = 0;
	const size_t esz = elem_size(b->fmt);
	b->has_generic = b->has_fast = false;
	b->n_gen_small = b->n_gen_large = b->n_gen_ola = 0;
	b->has_tdonly = false;
	const uint32_t n0h = (1u << id.bs0) / 2, n1h = (1u << id.bs1) / 2;
	for (size_t i = 0; i < n; i++) {
		LwPacketRec &r = b->h_recs[i];
		lw_packet_result &res = b->results[i];
		res.status = b->status[i];
		res.n_samples = 0;
		res.out_offset = out_off;
		r.out_off = (uint32_t)out_off;
		if (b->status[i] != LW_OK) {
			r.flags = LW_RF_SKIP;
			continue;
		}
		lw_pwr *pw = pkts[i].pwr;
		const lw::Prologue &p = b->prologues[i];
		const lw::WindowInfo w = lw::window_info(id, p.blockflag, p.prev_flag, p.next_flag);
		r.bs = p.bs;
		r.mode = p.mode;
		r.ls = (uint16_t)w.left_start;
		r.rs = (uint16_t)w.right_start;
		r.re = (uint16_t)w.right_end;
		r.flags = (p.blockflag ? LW_RF_LONG : 0) | (w.left_use_bs1 ? LW_RF_SLOPE_BS1 : 0);
		if (b->slot_seen[pw->slot] != b->epoch) {
			b->slot_seen[pw->slot] = b->epoch;
			b->touched.push_back(pw);
		}
		if (pw->present) {
			const uint32_t slope_len = w.left_use_bs1 ? n1h : n0h;
			if (slope_len < pw->len) {
				// audio.rs:1107-1111: error after pwr.data.take() -> the state is gone
				pw->present = false;
				pw->len = 0;
				b->slot_last[pw->slot] = -1;
				res.status = b->status[i] = LW_AUDIO_BAD_FORMAT;
				r.flags = LW_RF_SKIP;
				continue;
			}
			r.plen = (uint16_t)pw->len;
			const int32_t last = b->slot_last[pw->slot];
			if (last >= 0) {
				r.prev = last;
			} else {
				r.prev = -(pw->slot + 2);
				if (pw->parity)
					r.flags |= LW_RF_PARITY_IN;
			}
			res.n_samples = w.right_start - w.left_start;
		} else {
			r.prev = -1; // audio.rs:1140-1152: no previous window -> zero samples
			r.plen = 0;
		}
		// specialised kernel: long block, both neighbours long, stored right part (if any) is a full long half
		// Other long blocks of an eligible stream (window shapes next to short blocks, a stored right part of another
		// length) still get floor, decoupling and IMDCT from the specialised kernel, which writes their whole time-domain
		// block; k_ola_generic does their window / overlap-add / state (LW_RF_TDONLY).
		if (d->fast.eligible && !b->force_generic && p.blockflag && (d->fast.long_mode_mask[p.mode >> 3] & (1u << (p.mode & 7)))) {
			r.flags |= LW_RF_FAST;
			if (!(p.prev_flag && p.next_flag && (r.prev == -1 || r.plen == n1h))) {
				r.flags |= LW_RF_TDONLY;
				b->has_tdonly = true;
			}
			b->fast_idx.push_back((uint32_t)i);
			b->fast_slot.push_back((uint32_t)pw->slot);
		}
		pw->present = true;
		pw->len = w.right_end - w.right_start;
		b->slot_last[pw->slot] = (int32_t)i;
		out_off += (size_t)res.n_samples * ch;
		alg += (uint64_t)ch * (p.n / 2) * 4 + 16 + (uint64_t)res.n_samples * ch * esz + d->mode_floor_bytes[p.mode];
		if (!(r.flags & LW_RF_FAST)) {
			b->has_generic = true;
			if (p.bs <= LW_SMALL_BS)
				b->h_gen[b->n_gen_small++] = (uint32_t)i;
			else
				b->h_gen[b->max_packets + b->n_gen_large++] = (uint32_t)i;
		}
		if (!(r.flags & LW_RF_FAST) || (r.flags & LW_RF_TDONLY)) {
			b->has_generic = true; // k_ola_generic has work
			b->h_gen[2 * b->max_packets + b->n_gen_ola++] = (uint32_t)i;
		}
	}
	// the last ok packet of every stream hands its right part to the stream's state slot
	for (lw_pwr *pw : b->touched) {
		const int32_t last = b->slot_last[pw->slot];
		if (last >= 0) {
			LwPacketRec &r = b->h_recs[last];
			r.state_out = pw->slot;
			const uint8_t outp = pw->parity ^ 1;
			if (outp)
				r.flags |= LW_RF_PARITY_OUT;
			pw->parity = outp;
		}
		b->slot_last[pw->slot] = -1;
	}
	b->out_elems = out_off;
	b->alg_bytes = alg;

	// ---- workgroups of the fused small-block kernel: runs of consecutive entries of the overlap-add list that are consecutive
	// packets of one stream, cut at 16 / ch members (one wave per member and channel).  Used when the batch has small generic
	// blocks, every coupling step is a disjoint pair and a packet's channels fit one workgroup -- and only on request
	// (LW_SMALL_FUSED=1): measured slower than the three generic kernels so far (lw_kernels.hip); otherwise those run as
	// before (b->n_seg == 0).
	b->n_seg = 0;
	if (b->n_gen_small && d->T.pair_coupling && ch <= 8 && !b->force_generic && getenv("LW_SMALL_FUSED")) {
		const uint32_t *e = b->h_gen + 2 * b->max_packets;
		const uint32_t ppw = lw_small_fused_members((uint32_t)ch);
		auto small_generic = [&](const LwPacketRec &r) { return !(r.flags & LW_RF_FAST) && r.bs <= LW_SMALL_BS; };
		uint32_t start = 0;
		auto close = [&](uint32_t end) {
			if (end == start)
				return;
			LwSegment &sg = b->h_seg[b->n_seg++];
			sg.first = start;
			sg.count = (uint16_t)(end - start);
			const int32_t p = b->h_recs[e[start]].prev;
			sg.halo = (p >= 0 && small_generic(b->h_recs[p])) ? 1 : 0;
			start = end;
		};
		// (a segment whose first member's predecessor is a small block of another segment recomputes that block on the
		// waves of one member slot: such a segment holds ppw - 1 members, and at least one)
		auto cap_of = [&](uint32_t first) {
			const int32_t p = b->h_recs[e[first]].prev;
			const bool halo = p >= 0 && small_generic(b->h_recs[p]);
			return halo ? std::max(1u, ppw - 1) : ppw;
		};
		uint32_t cap = b->n_gen_ola ? cap_of(0) : ppw;
		for (uint32_t i = 1; i < b->n_gen_ola; i++)
			if (b->h_recs[e[i]].prev != (int32_t)e[i - 1] || i - start == cap) {
				close(i);
				cap = cap_of(i);
			}
		close(b->n_gen_ola);
	}

	// ---- work plan of the specialised kernel: items sorted by stream so that consecutive packets of a
	// stream sit in consecutive items; a workgroup works through a chunk of rounds * per_round consecutive
	// items and hands right halves over in LDS; a predecessor outside the chunk is recomputed by the halo pre-pass
	b->n_items = b->n_halo_items = 0;
	b->has_fast = !b->fast_idx.empty();
	if (b->has_fast) {
		const size_t nf = b->fast_idx.size();
		for (size_t i = 0; i < n; i++) { // generic successors of fast packets read the td block
			const LwPacketRec &r = b->h_recs[i];
			const bool ola_generic = !(r.flags & LW_RF_FAST) || (r.flags & LW_RF_TDONLY);
			if (!(r.flags & LW_RF_SKIP) && ola_generic && r.prev >= 0 && (b->h_recs[r.prev].flags & LW_RF_FAST))
				b->h_recs[r.prev].flags |= LW_RF_WRITE_TD;
		}
		b->fast_order.resize(nf);
		for (size_t k = 0; k < nf; k++)
			b->fast_order[k] = (uint32_t)k;
		if (!std::is_sorted(b->fast_slot.begin(), b->fast_slot.end())) // (callers usually list their streams one after the other)
			std::stable_sort(b->fast_order.begin(), b->fast_order.end(),
					[&](uint32_t a, uint32_t c) { return b->fast_slot[a] < b->fast_slot[c]; });
		uint32_t per_round = LW_FAST_WAVES / (uint32_t)d->fast.units.size();
		// as few rounds per workgroup as two resident workgroups per CU allow: small batches spread over the whole
		// chip; big batches get long chunks (LDS hand-over, few halo recomputations)
		const size_t per_pass = (size_t)per_round * std::max(1, d->n_cus);
		uint32_t rounds = (uint32_t)std::min<size_t>(LW_FAST_MAX_ROUNDS, std::max<size_t>(1, (nf + per_pass - 1) / per_pass));
		if (const char *e = getenv("LW_FAST_ROUNDS")) { // test hook: force the number of rounds per workgroup
			rounds = (uint32_t)std::min(LW_FAST_MAX_ROUNDS, std::max(1, atoi(e)));
		} else if (rounds == 1) {
			// fewer packets than one full round per CU (the long blocks of a mixed short/long batch, a small batch): fewer
			// packets per workgroup, so that every CU gets some (1 117 long packets in chunks of 16 kept 186 of 256 CUs idle)
			per_round = (uint32_t)std::min<size_t>(per_round, std::max<size_t>(1, (nf + d->n_cus - 1) / std::max(1, d->n_cus)));
		}
		const uint32_t chunk = per_round * rounds;
		b->fast_per_round = per_round;
		b->fast_rounds = rounds;
		b->fast_dense = 1;
		auto fill = [&](LwFastItem &it, uint32_t idx) {
			const LwPacketRec &r = b->h_recs[idx];
			std::memset(&it, 0, sizeof(it));
			it.res_off = r.res_off;
			it.floor_off = r.floor_off;
			it.out_off = r.out_off;
			it.state_out = r.state_out;
			it.mode = r.mode;
			it.flags = (uint8_t)(r.flags & (LW_RF_PARITY_IN | LW_RF_PARITY_OUT | LW_RF_WRITE_TD | LW_RF_TDONLY));
			if (r.flags & LW_RF_TDONLY) {
				it.flags |= LW_RF_WRITE_TD; // left AND right half go to the td block
				it.state_out = -1;          // the state slot is written by k_ola_generic
			}
			it.pkt = idx;
		};
		for (size_t k = 0; k < nf; k++) {
			const uint32_t idx = b->fast_idx[b->fast_order[k]];
			const LwPacketRec &r = b->h_recs[idx];
			LwFastItem &it = b->h_items[k];
			fill(it, idx);
			if (it.res_off != (uint32_t)(k * ch * n1h) || it.floor_off != (uint32_t)(k * ch * fstride))
				b->fast_dense = 0;
			if (r.prev == -1 || (r.flags & LW_RF_TDONLY)) {
				it.src_kind = LW_SRC_NONE; // (a TD-only packet is overlapped later, by k_ola_generic)
			} else if (r.prev <= -2) {
				it.src_kind = LW_SRC_STATE;
				it.src_arg = (uint32_t)(-(r.prev + 2));
			} else if (b->h_recs[r.prev].flags & LW_RF_FAST) {
				if ((k % chunk) != 0 && b->h_items[k - 1].pkt == (uint32_t)r.prev) {
					it.src_kind = LW_SRC_LDS;
					b->h_items[k - 1].flags |= LW_IF_NEXT_LDS;
				} else {
					it.src_kind = LW_SRC_HALO;
					it.src_arg = (uint32_t)b->n_halo_items;
					LwFastItem &h = b->h_halo_items[b->n_halo_items];
					fill(h, (uint32_t)r.prev);
					h.state_out = -1;
					h.halo_out = (uint32_t)b->n_halo_items++;
				}
			} else {
				it.src_kind = LW_SRC_TD;
				it.src_arg = 2u * b->h_recs[r.prev].res_off;
			}
		}
		b->n_items = nf;
	}
	return LW_OK;
}

int lw_batch_upload(lw_batch *b, void *hip_stream)
{
	if (!b)
		return LW_ERR_NULL_ARG;
	if (int rc = decoder_set_device(b->dec))
		return rc;
	hipStream_t st = (hipStream_t)hip_stream;
	const size_t ch = b->dec->T.ch;
	if (b->n == 0)
		return LW_OK;
	if (b->slab_bytes <= 64 * 1024 && !b->symbols) { // small batch: the whole slab in one copy
		HIP_TRY(hipMemcpyAsync(b->d_slab, b->h_slab, b->slab_bytes, hipMemcpyHostToDevice, st));
		return LW_OK;
	}
	HIP_TRY(hipMemcpyAsync(b->d_recs, b->h_recs, b->n * sizeof(LwPacketRec), hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(b->d_floor, b->h_floor, b->n * ch * b->dec->T.fstride * sizeof(uint16_t),
				hipMemcpyHostToDevice, st));
	if (b->res_floats && !b->symbols)
		HIP_TRY(hipMemcpyAsync(b->d_res, b->h_res, b->res_floats * sizeof(float), hipMemcpyHostToDevice, st));
	if (b->symbols) {
		HIP_TRY(hipMemcpyAsync(b->d_sym_off, b->h_sym_off, b->n * sizeof(uint32_t), hipMemcpyHostToDevice, st));
		if (b->sym_words)
			HIP_TRY(hipMemcpyAsync(b->d_sym, b->h_sym, b->sym_words * sizeof(uint32_t), hipMemcpyHostToDevice, st));
	}
	if (b->res_floats && b->d_fcurve)
		HIP_TRY(hipMemcpyAsync(b->d_fcurve, b->h_fcurve, b->res_floats * sizeof(float), hipMemcpyHostToDevice, st));
	if (b->n_gen_small)
		HIP_TRY(hipMemcpyAsync(b->d_gen, b->h_gen, b->n_gen_small * sizeof(uint32_t), hipMemcpyHostToDevice, st));
	if (b->n_gen_large)
		HIP_TRY(hipMemcpyAsync(b->d_gen + b->max_packets, b->h_gen + b->max_packets, b->n_gen_large * sizeof(uint32_t),
					hipMemcpyHostToDevice, st));
	if (b->n_gen_ola)
		HIP_TRY(hipMemcpyAsync(b->d_gen + 2 * b->max_packets, b->h_gen + 2 * b->max_packets, b->n_gen_ola * sizeof(uint32_t),
					hipMemcpyHostToDevice, st));
	if (b->n_seg)
		HIP_TRY(hipMemcpyAsync(b->d_seg, b->h_seg, b->n_seg * sizeof(LwSegment), hipMemcpyHostToDevice, st));
	if (b->n_items)
		HIP_TRY(hipMemcpyAsync(b->d_items, b->h_items, b->n_items * sizeof(LwFastItem), hipMemcpyHostToDevice, st));
	if (b->n_halo_items)
		HIP_TRY(hipMemcpyAsync(b->d_halo_items, b->h_halo_items, b->n_halo_items * sizeof(LwFastItem), hipMemcpyHostToDevice, st));
	return LW_OK;
}

static int batch_launch(lw_batch *b, void *d_out, hipStream_t st, bool all_generic, float *tap)
{
	lw_decoder *d = b->dec;
	if (b->n == 0)
		return LW_OK;
	const bool run_generic = b->has_generic || all_generic;
	const bool run_fast = b->has_fast && !all_generic;
	if (run_generic) {
		const size_t maxres = b->max_packets * d->T.ch * d->T.state_chan_stride;
		if (d->any_coupling && !b->d_decoupled)
			HIP_TRY(hipMalloc((void **)&b->d_decoupled, maxres * sizeof(float)));
		if (!b->d_td)
			HIP_TRY(hipMalloc((void **)&b->d_td, 2 * maxres * sizeof(float)));
	}
	if (run_fast && b->n_halo_items > b->halo_cap) {
		if (b->d_halo) {
			HIP_TRY(hipStreamSynchronize(st));
			(void)hipFree(b->d_halo);
			b->d_halo = nullptr;
		}
		const size_t cap = std::max<size_t>(b->n_halo_items, 64);
		HIP_TRY(hipMalloc((void **)&b->d_halo, cap * d->T.ch * 512 * sizeof(float)));
		b->halo_cap = cap;
	}
	LwBatchDev B{};
	B.recs = b->d_recs;
	B.floors = b->d_floor;
	B.residue = b->d_res;
	B.fcurve = b->d_fcurve;
	B.decoupled = b->d_decoupled;
	B.td = b->d_td;
	B.state = d->d_state;
	B.n_packets = (uint32_t)b->n;
	// dense lists of the generic packets (not used when every packet goes through the generic kernels)
	B.gen_small = all_generic ? nullptr : b->d_gen;
	B.gen_large = all_generic ? nullptr : b->d_gen + b->max_packets;
	B.n_gen_small = b->n_gen_small;
	B.n_gen_large = b->n_gen_large;
	B.gen_ola = all_generic ? nullptr : b->d_gen + 2 * b->max_packets;
	B.n_gen_ola = b->n_gen_ola;
	B.sym = b->symbols ? b->d_sym : nullptr;
	B.sym_off = b->d_sym_off;
	b->last_kernels.clear();
	if (b->symbols) {
		lw_launch_residue_vq(d->T, d->V, B, st, b->max_n, d->vq_book_ends.data(), d->vq_book_ends.size());
		b->last_kernels = "k_residue_vq,";
	}
	const bool fused_small = run_generic && !all_generic && !tap && b->n_seg > 0;
	B.seg = fused_small ? b->d_seg : nullptr;
	B.n_seg = fused_small ? b->n_seg : 0;
	if (fused_small) {
		if (b->n_gen_large) { // large generic blocks still go through k_decouple / k_imdct_generic into B.td
			lw_launch_generic_imdct_large(d->T, B, st, b->max_n, d->any_coupling);
			b->last_kernels += d->any_coupling ? "k_decouple,k_imdct_generic," : "k_imdct_generic,";
		}
	} else if (run_generic) {
		lw_launch_generic_imdct(d->T, B, tap, st, b->max_n, d->any_coupling, all_generic);
		b->last_kernels += d->any_coupling ? "k_decouple,k_imdct_generic," : "k_imdct_generic,";
	}
	if (run_fast) {
		LwFastLaunch L{};
		L.off = d->fast.off;
		L.d_image = d->d_fast_image;
		L.d_items = b->d_items;
		L.n_items = (uint32_t)b->n_items;
		L.d_halo_items = b->d_halo_items;
		L.n_halo_items = (uint32_t)b->n_halo_items;
		L.n_units = (uint32_t)d->fast.units.size();
		L.per_round = b->fast_per_round;
		L.rounds = b->fast_rounds;
		L.dense = b->fast_dense;
		L.late_from = b->fast_late_from;
		L.has_tdonly = b->has_tdonly ? 1u : 0u;
		if (const char *e = getenv("LW_PACE_GROUP"))
			L.late_from = (uint32_t)atoi(e);
		for (size_t i = 0; i < d->fast.units.size() && i < LW_FAST_WAVES; i++)
			L.units[i] = d->fast.units[i];
		L.d_halo = b->d_halo;
		lw_launch_long(d->T, B, L, d_out, b->fmt, st);
		b->last_kernels += b->n_halo_items ? "k_long<halo>,k_long," : "k_long,";
	}
	if (fused_small) {
		lw_launch_small_fused(d->T, B, d_out, b->fmt, st, b->max_n);
		b->last_kernels += "k_small_fused,";
	} else if (run_generic) {
		lw_launch_generic_ola(d->T, B, d_out, b->fmt, st, all_generic);
		b->last_kernels += "k_ola_generic,";
	}
	if (!b->last_kernels.empty())
		b->last_kernels.pop_back();
	HIP_TRY(hipGetLastError());
	return LW_OK;
}

int lw_batch_synth(lw_batch *b, void *d_out, size_t out_capacity_elems, void *hip_stream)
{
	if (!b || (!d_out && b->out_elems))
		return LW_ERR_NULL_ARG;
	if (out_capacity_elems < b->out_elems)
		return LW_ERR_CAPACITY;
	if (int rc = decoder_set_device(b->dec))
		return rc;
	return batch_launch(b, d_out, (hipStream_t)hip_stream, b->force_generic, nullptr);
}

static int ensure_internal_out(lw_batch *b)
{
	if (b->d_out_elems >= b->out_elems && b->d_out)
		return LW_OK;
	if (b->d_out)
		(void)hipFree(b->d_out);
	b->d_out = nullptr;
	const size_t cap = std::max<size_t>(b->out_elems, b->max_packets * b->dec->T.ch * b->dec->T.state_chan_stride);
	HIP_TRY(hipMalloc(&b->d_out, cap * elem_size(b->fmt)));
	b->d_out_elems = cap;
	return LW_OK;
}

int lw_batch_synth_to_host(lw_batch *b, void *h_out, size_t out_capacity_elems, void *hip_stream)
{
	if (!b || (!h_out && b->out_elems))
		return LW_ERR_NULL_ARG;
	if (out_capacity_elems < b->out_elems)
		return LW_ERR_CAPACITY;
	if (int rc = decoder_set_device(b->dec))
		return rc;
	if (int rc = ensure_internal_out(b))
		return rc;
	hipStream_t st = (hipStream_t)hip_stream;
	if (int rc = batch_launch(b, b->d_out, st, b->force_generic, nullptr))
		return rc;
	if (b->out_elems)
		HIP_TRY(hipMemcpyAsync(h_out, b->d_out, b->out_elems * elem_size(b->fmt), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	return LW_OK;
}

int lw_batch_tap(lw_batch *b, size_t idx, int tap, float *dst, size_t cap_floats)
{
	if (!b || !dst)
		return LW_ERR_NULL_ARG;
	if (idx >= b->n || b->status[idx] != LW_OK)
		return LW_ERR_CAPACITY;
	lw_decoder *d = b->dec;
	if (int rc = decoder_set_device(d))
		return rc;
	const LwPacketRec &r = b->h_recs[idx];
	const size_t n = (size_t)1 << r.bs, ch = d->T.ch;
	const size_t want = tap == LW_TAP_POST_MDCT ? ch * n : ch * n / 2;
	if (cap_floats < want)
		return LW_ERR_CAPACITY;
	if (tap == LW_TAP_RESIDUE_PRE_INVERSE && !b->symbols) {
		std::memcpy(dst, b->h_res + r.res_off, want * sizeof(float));
		return LW_OK;
	}
	if (int rc = ensure_internal_out(b))
		return rc;
	if (!b->d_tap)
		HIP_TRY(hipMalloc((void **)&b->d_tap, b->max_packets * ch * d->T.state_chan_stride * sizeof(float)));
	if (int rc = batch_launch(b, b->d_out, nullptr, true, b->d_tap))
		return rc;
	HIP_TRY(hipDeviceSynchronize());
	const float *src;
	if (tap == LW_TAP_RESIDUE_PRE_INVERSE)
		src = b->d_res + r.res_off; // Tier B: the vectors k_residue_vq built on the device
	else if (tap == LW_TAP_RESIDUE_POST_INVERSE)
		src = (d->any_coupling ? b->d_decoupled : b->d_res) + r.res_off;
	else if (tap == LW_TAP_PRE_MDCT)
		src = b->d_tap + r.res_off;
	else
		src = b->d_td + 2 * (size_t)r.res_off;
	HIP_TRY(hipMemcpy(dst, src, want * sizeof(float), hipMemcpyDeviceToHost));
	return LW_OK;
}

int lw_debug_imdct(lw_decoder *d, int blockflag, const float *spectrum, float *out)
{
	if (!d || !spectrum || !out)
		return LW_ERR_NULL_ARG;
	if (int rc = decoder_set_device(d))
		return rc;
	const lw::Setup &s = *d->setup;
	int mode = -1;
	for (size_t m = 0; m < s.modes.size(); m++)
		if ((int)s.modes[m].blockflag == (blockflag ? 1 : 0))
			mode = (int)m;
	if (mode < 0)
		return LW_ERR_CAPACITY;
	int e = 0;
	lw_batch *b = lw_batch_create(d, 1, LW_FMT_F32_PLANAR, &e);
	if (!b)
		return e ? e : LW_ERR_DEVICE;
	const uint32_t bs = blockflag ? d->id->bs1 : d->id->bs0, n = 1u << bs, ch = d->T.ch;
	LwPacketRec &r = b->h_recs[0];
	std::memset(&r, 0, sizeof(r));
	r.prev = -1;
	r.state_out = -1;
	r.bs = (uint8_t)bs;
	r.mode = (uint8_t)mode;
	r.flags = blockflag ? LW_RF_LONG : 0;
	r.rs = (uint16_t)(n / 2);
	r.re = (uint16_t)(n / 2); // nothing to hand over
	std::memset(b->h_res, 0, sizeof(float) * ch * n / 2);
	std::memcpy(b->h_res, spectrum, sizeof(float) * n / 2);
	const lw::Mapping &mp = s.mappings[s.modes[mode].mapping];
	for (uint32_t c = 0; c < ch; c++) {
		uint16_t *rec = b->h_floor + c * d->T.fstride;
		const size_t F = s.floors[mp.submap_floor[mp.mux[c]]].f1.x_list.size();
		for (size_t i = 0; i < F; i++)
			rec[i] = 0;
		rec[0] = LW_POST_ACTIVE | 255u;     // x = 0
		rec[F - 1] = LW_POST_ACTIVE | 255u; // largest x; beyond it the curve stays flat
	}
	b->n = 1;
	b->res_floats = (size_t)ch * n / 2;
	b->max_n = n;
	b->out_elems = 0;
	b->status[0] = LW_OK;
	int rc = lw_batch_upload(b, nullptr);
	if (!rc)
		rc = ensure_internal_out(b);
	if (!rc)
		rc = batch_launch(b, b->d_out, nullptr, true, nullptr);
	if (!rc && !hip_ok(hipDeviceSynchronize(), "sync"))
		rc = LW_ERR_DEVICE;
	if (!rc && !hip_ok(hipMemcpy(out, b->d_td, sizeof(float) * n, hipMemcpyDeviceToHost), "memcpy td"))
		rc = LW_ERR_DEVICE;
	lw_batch_destroy(b);
	return rc;
}

// ---- one packet ---------------------------------------------------------------------------------
int lw_read_audio_packet(lw_decoder *d, const uint8_t *packet, size_t len, lw_pwr *pwr, int fmt, void *out,
		size_t cap_per_channel, size_t *n_samples)
{
	if (!d || (!packet && len) || !pwr || !out || !n_samples)
		return LW_ERR_NULL_ARG;
	if (pwr->dec != d)
		return LW_ERR_STATE_MISMATCH;
	if (fmt < 0 || fmt > 2)
		return LW_ERR_NULL_ARG;
	if (int rc = decoder_set_device(d))
		return rc;
	if (!d->one || d->one->fmt != fmt) {
		if (d->one)
			lw_batch_destroy(d->one);
		int e = 0;
		d->one = lw_batch_create(d, 1, fmt, &e);
		if (!d->one)
			return e ? e : LW_ERR_DEVICE;
	}
	lw_batch *b = d->one;
	lw_packet pk{packet, len, pwr};
	// lw_batch_entropy commits the host half of the PreviousWindowRight (present, len, parity) when it plans the batch; the
	// device half follows when the kernels run.  Any failure in between must leave `pwr` as the reference leaves it on an
	// error: untouched (the one error that consumes the state, audio.rs:1107-1111, is reported through res.status).
	const lw_pwr saved = *pwr;
	if (int rc = lw_batch_entropy(b, &pk, 1, 1)) {
		*pwr = saved;
		return rc;
	}
	const lw_packet_result &res = b->results[0];
	if (res.status != LW_OK)
		return res.status;
	int rc = LW_OK;
	if (res.n_samples > cap_per_channel)
		rc = LW_AUDIO_BUFFER_NOT_ADDRESSABLE;
	if (!rc)
		rc = lw_batch_upload(b, nullptr);
	const size_t need = std::max<size_t>(b->out_elems, 1) * elem_size(fmt);
	if (!rc && d->one_out_bytes < need) {
		if (d->one_out)
			(void)hipHostFree(d->one_out);
		d->one_out = nullptr;
		d->one_out_bytes = 0;
		const size_t bytes = std::max<size_t>(need, (size_t)d->T.state_stride * 2 * 4);
		if (hip_ok(hipHostMalloc(&d->one_out, bytes), "hipHostMalloc(packet output)"))
			d->one_out_bytes = bytes;
		else
			rc = LW_ERR_DEVICE;
	}
	// the kernels write the PCM straight into the pinned host buffer (device-visible): launch + synchronise, no D2H copy
	if (!rc)
		rc = lw_batch_synth(b, d->one_out, b->out_elems, nullptr); // (a first packet yields no samples, only the state)
	if (!rc && !hip_ok(hipStreamSynchronize(nullptr), "hipStreamSynchronize"))
		rc = LW_ERR_DEVICE;
	if (rc) {
		*pwr = saved; // the packet was not decoded: the next call overlaps against the state this one found
		return rc;
	}
	std::memcpy(out, d->one_out, b->out_elems * elem_size(fmt));
	*n_samples = res.n_samples;
	return LW_OK;
}

} // extern "C"

namespace {
// FLOOR1_INVERSE_DB_TABLE: the Vorbis I specification's floor1_inverse_dB_table (spec 10.1; audio.rs:437-501).
// The spec defines it by value; entry i is approximately 1.0649863e-07 * exp(i * 0.06264...) but decoders must
// use the printed constants.
const float kInverseDbTable[256] = {
#include "lw_inverse_db.inc"
};
} // namespace
