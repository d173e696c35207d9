// Runtime + C ABI of the MI355X audio-packet decode path (product code, compiled with hipcc).
//
// Implements include/lewton_amd.h: header objects, the device context (tables in HBM) and the device-resident
// PreviousWindowRight pool.  Batches live in lw_batch.cpp, the drop-in single-packet call in lw_packet.cpp, the staging
// ring in lw_ring.cpp, the stream sharder in lw_shard.cpp, the worker pool in lw_pool.cpp.  There is no CPU fallback for the synthesis stage: without a usable GPU every
// device call returns LW_ERR_DEVICE.
#include "../../include/lewton_amd.h"

#include "lw_internal.hpp"
#include "lw_dev_entropy.hpp"
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

namespace {
thread_local std::string g_dev_err;
extern const float kInverseDbTable[256];
} // namespace

bool lw_hip_ok(hipError_t e, const char *what)
{
	if (e == hipSuccess)
		return true;
	g_dev_err = std::string(what) + ": " + hipGetErrorString(e);
	(void)hipGetLastError();
	return false;
}

void lw_set_device_error(const std::string &msg)
{
	g_dev_err = msg;
}

int lw_decoder_set_device(const lw_decoder *d)
{
	HIP_TRY(hipSetDevice(d->device));
	return LW_OK;
}

int lw_grow_state(lw_decoder *d, size_t slots)
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


extern "C" {

const char *lw_version(void)
{
	return "lewton_amd 0.1 (gfx950)";
}

int lw_default_host_threads(void)
{
	return (int)lw::default_host_threads();
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

size_t lw_debug_short_image(const lw_ident *id, const lw_setup *s, int blockflag, uint8_t *dst, size_t cap, uint8_t *units8,
		size_t *n_units, uint32_t *lanes)
{
	if (!id || !s)
		return 0;
	LwFastPlan fast;
	lw::build_fast_plan(*id->p, *s->p, fast);
	LwShortPlan plan;
	lw::build_blk_plan(*id->p, *s->p, blockflag != 0, fast, plan);
	if (!plan.eligible)
		return 0;
	if (!plan.image.empty())
		std::memcpy(plan.image.data() + lw_blk_inv_db_offset(plan.lanes), kInverseDbTable, sizeof(float) * 256);
	if (dst)
		std::memcpy(dst, plan.image.data(), std::min(cap, plan.image.size()));
	if (units8 && n_units) {
		const size_t n = std::min(*n_units, plan.units.size());
		std::memcpy(units8, plan.units.data(), n * sizeof(LwFastUnit));
	}
	if (n_units)
		*n_units = plan.units.size();
	if (lanes)
		*lanes = plan.lanes;
	return plan.image.size();
}

size_t lw_debug_plan_census(const lw_ident *id, const lw_setup *s, char *dst, size_t cap)
{
	if (!id || !s)
		return 0;
	const lw::Ident &I = *id->p;
	const lw::Setup &S = *s->p;
	LwFastPlan fast;
	lw::build_fast_plan(I, S, fast);
	LwShortPlan blk[2];
	lw::build_blk_plan(I, S, false, fast, blk[0]);
	lw::build_blk_plan(I, S, true, fast, blk[1]);
	bool any_long = false, any_short = false;
	for (const lw::Mode &m : S.modes)
		(m.blockflag ? any_long : any_short) = true;
	if (lw::lw_unified_classes(I, S))
		any_short = false; // (equal block sizes: every mode is planned with the long blocks)
	auto blk_name = [](const LwShortPlan &p) -> std::string {
		if (p.lanes > 64)
			return p.lanes == 128 ? "k_big<12>" : "k_big<13>";
		return "k_short<" + std::to_string(p.lanes) + ">";
	};
	// the routing of lw_batch_entropy (csrc/lw_batch.cpp), per block class
	std::string lng, sht, edge = "none";
	bool use_l10 = false, use_l12 = false;
	if (!any_long)
		lng = "none";
	else if (fast.eligible)
		lng = "k_long";
	else if (blk[1].eligible) {
		if (blk[1].lanes == 32 && blk[1].units.size() <= LW_FAST_WAVES) {
			lng = "k_long10";
			use_l10 = true;
		} else if (blk[1].lanes == 128 && !blk[1].units_split.empty() && !blk[1].image.empty()) {
			lng = "k_long12";
			use_l12 = true;
		} else
			lng = blk_name(blk[1]);
	} else
		lng = std::string("generic (") + (I.bs1 == LW_FAST_BS ? fast.why_not : blk[1].why_not) + ")";
	{ // stream shapes the kernel takes only behind the canonicalising pre-pass (LwPrepPlan)
		const LwPrepPlan *pp = fast.eligible ? &fast.prep : blk[1].eligible ? &blk[1].prep : nullptr;
		if (any_long && pp && pp->on)
			lng += std::string(" + k_prep (") + pp->why + ")";
		else if (any_long && fast.eligible && !fast.pre.empty())
			lng += std::string(", coupling steps inside the waves (") + fast.prep.why + ")";
	}
	if (!any_short)
		sht = "none";
	else if (blk[0].eligible) {
		if (!any_long && blk[0].lanes == 32 && I.bs0 == I.bs1 && !fast.eligible && blk[0].units.size() <= LW_FAST_WAVES)
			sht = "k_long10";
		else
			sht = blk_name(blk[0]);
	} else
		sht = std::string("generic (") + blk[0].why_not + ")";
	if (any_short && blk[0].eligible && blk[0].prep.on)
		sht += std::string(" + k_prep (") + blk[0].prep.why + ")";
	if (any_long && any_short) { // long blocks with a short slope
		const bool short_ok10 = (use_l10 || use_l12) && blk[0].eligible && (blk[0].bs == 8 || blk[0].bs == 9);
		if ((fast.eligible && blk[0].eligible && blk[0].bs == 8) || short_ok10)
			edge = "edge form";
		else if (fast.eligible)
			edge = "k_long + k_ola_generic";
		else if (use_l10 || use_l12) // (the wave kernel's time-domain block, the generic overlap-add: LW_RF_TDONLY)
			edge = use_l12 ? "k_long12 + k_ola_generic" : "k_long10 + k_ola_generic";
		else
			edge = "generic";
	}
	lw::DevEntropyImage img;
	const char *why = "";
	const bool dev = lw::dev_entropy_build(I, S, floor_stride_of(S), img, &why);
	const std::string out = "long=" + lng + " | short=" + sht + " | transitions=" + edge + " | entropy=" +
		(dev ? std::string("device") : std::string("host (") + why + ")");
	if (dst && cap) {
		const size_t n = std::min(cap - 1, out.size());
		std::memcpy(dst, out.data(), n);
		dst[n] = 0;
	}
	return out.size();
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
	if (!lw_hip_ok(hipGetDeviceCount(&n), "hipGetDeviceCount"))
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
	if (!lw_hip_ok(hipGetDeviceCount(&ndev), "hipGetDeviceCount") || device < 0 || device >= ndev ||
			!lw_hip_ok(hipSetDevice(device), "hipSetDevice")) {
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
	d->n_cus_device = d->n_cus;
	{
		hipDeviceProp_t prop;
		d->is_gfx950 = hipGetDeviceProperties(&prop, device) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
	}

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
	// (rows nfl, nfl + 1: the unit floors of the two block classes, LwPrepPlan -- posts at x = 0 and x = n / 2)
	std::vector<uint16_t> fx((nfl + 2) * LW_XSTRIDE, 0);
	std::vector<uint8_t> fF(nfl + 2, 0);
	for (size_t cls = 0; cls < 2; cls++) {
		fx[(nfl + cls) * LW_XSTRIDE + 1] = (uint16_t)((1u << (cls ? id.bs1 : id.bs0)) / 2);
		fF[nfl + cls] = 2;
	}
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
	d->h_mode_floor = mode_floor; // host copies: the planner packs them into the short-block kernel's task descriptors
	d->h_floor_F = fF;
	d->h_mode_partner = mode_partner;
	d->h_mode_role = mode_role;
	const size_t off_fx = put(fx.data(), fx.size() * 2);
	const size_t off_fF = put(fF.data(), fF.size());
	const size_t off_mf = put(mode_floor.data(), mode_floor.size());
	const size_t off_co = put(couple_off.data(), couple_off.size() * 2);
	const size_t off_cp = put(couple.data(), couple.size());
	const size_t off_mp = put(mode_partner.data(), mode_partner.size());
	const size_t off_mr = put(mode_role.data(), mode_role.size());

	if (!lw_hip_ok(hipMalloc(&d->d_blob, blob.size()), "hipMalloc(tables)") ||
			!lw_hip_ok(hipMemcpy(d->d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice), "hipMemcpy(tables)")) {
		*err = LW_ERR_DEVICE;
		lw_decoder_destroy(d.release()); // (every failure below as well: frees whatever has been allocated so far)
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
	{ // entropy stage on the device: the flattened setup image (eligible streams only; the host stage serves the others)
		lw::DevEntropyImage img;
		const char *why = "";
		if (lw::dev_entropy_build(id, s, (unsigned)d->T.fstride, img, &why)) {
			if (!lw_hip_ok(hipMalloc(&d->d_ent_blob, img.blob.size()), "hipMalloc(entropy image)") ||
					!lw_hip_ok(hipMemcpy(d->d_ent_blob, img.blob.data(), img.blob.size(), hipMemcpyHostToDevice),
						"hipMemcpy(entropy image)")) {
				*err = LW_ERR_DEVICE;
				lw_decoder_destroy(d.release());
				return nullptr;
			}
			d->E = lw::dev_entropy_view(img, (const uint8_t *)d->d_ent_blob);
			d->dev_entropy_ok = true;
		} else {
			d->dev_entropy_why = why;
		}
	}
	lw::build_fast_plan(id, s, d->fast);
	if (d->fast.eligible) {
		std::memcpy(d->fast.image.data() + d->fast.off.inv_db, kInverseDbTable, sizeof(float) * 256);
		const size_t ub = d->fast.units.size() * sizeof(LwFastUnit);
		if (!lw_hip_ok(hipMalloc((void **)&d->d_fast_image, d->fast.image.size()), "hipMalloc(fast image)") ||
				!lw_hip_ok(hipMemcpy(d->d_fast_image, d->fast.image.data(), d->fast.image.size(), hipMemcpyHostToDevice),
					"hipMemcpy(fast image)") ||
				!lw_hip_ok(hipMalloc((void **)&d->d_fast_units, ub), "hipMalloc(fast units)") ||
				!lw_hip_ok(hipMemcpy(d->d_fast_units, d->fast.units.data(), ub, hipMemcpyHostToDevice), "hipMemcpy(fast units)")) {
			*err = LW_ERR_DEVICE;
			lw_decoder_destroy(d.release());
			return nullptr;
		}
	}
	// k_short<L>: class 0 = the short blocks, class 1 = the long blocks of a stream k_long does not cover
	for (int cls = 0; cls < 2; cls++) {
		LwShortPlan &bp = d->blkp[cls];
		lw::build_blk_plan(id, s, cls != 0, d->fast, bp);
		if (!bp.eligible || bp.image.empty()) // (k_big: no image)
			continue;
		std::memcpy(bp.image.data() + lw_blk_inv_db_offset(bp.lanes), kInverseDbTable, sizeof(float) * 256);
		if (!lw_hip_ok(hipMalloc((void **)&d->d_blk_image[cls], bp.image.size()), "hipMalloc(block kernel image)") ||
				!lw_hip_ok(hipMemcpy(d->d_blk_image[cls], bp.image.data(), bp.image.size(), hipMemcpyHostToDevice),
					"hipMemcpy(block kernel image)")) {
			*err = LW_ERR_DEVICE;
			lw_decoder_destroy(d.release());
			return nullptr;
		}
		if (!bp.sid12.empty() && (!lw_hip_ok(hipMalloc((void **)&d->d_l12_sid, bp.sid12.size()), "hipMalloc(k_long12 interval table)") ||
					!lw_hip_ok(hipMemcpy(d->d_l12_sid, bp.sid12.data(), bp.sid12.size(), hipMemcpyHostToDevice),
						"hipMemcpy(k_long12 interval table)"))) {
			*err = LW_ERR_DEVICE;
			lw_decoder_destroy(d.release());
			return nullptr;
		}
	}
	// canonicalising pre-pass (k_prep): the (mode, channel) actions of the block classes that need it, one merged table
	{
		const LwPrepPlan *pp[2] = {d->blkp[0].eligible ? &d->blkp[0].prep : nullptr,
			d->fast.eligible ? &d->fast.prep : d->blkp[1].eligible ? &d->blkp[1].prep : nullptr};
		d->h_prep_action.assign(nmodes * ch, LW_PREP_NONE);
		d->h_prep_mode.assign(nmodes, 0);
		for (int cls = 0; cls < 2; cls++) {
			if (!pp[cls] || !pp[cls]->on)
				continue;
			d->prep_cls[cls] = true;
			d->prep_floors = d->prep_floors || pp[cls]->premul;
			for (size_t m = 0; m < nmodes; m++)
				for (size_t c = 0; c < ch; c++)
					if (pp[cls]->action[m * ch + c] != LW_PREP_NONE) {
						d->h_prep_action[m * ch + c] = pp[cls]->action[m * ch + c];
						d->h_prep_mode[m] = 1;
					}
		}
		if ((d->prep_cls[0] || d->prep_cls[1]) &&
				(!lw_hip_ok(hipMalloc((void **)&d->d_prep_action, d->h_prep_action.size()), "hipMalloc(pre-pass actions)") ||
				 !lw_hip_ok(hipMemcpy(d->d_prep_action, d->h_prep_action.data(), d->h_prep_action.size(), hipMemcpyHostToDevice),
					 "hipMemcpy(pre-pass actions)"))) {
			*err = LW_ERR_DEVICE;
			lw_decoder_destroy(d.release());
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
	if (d->d_ent_blob)
		(void)hipFree(d->d_ent_blob);
	if (d->d_fast_image)
		(void)hipFree(d->d_fast_image);
	if (d->d_fast_units)
		(void)hipFree(d->d_fast_units);
	for (uint8_t *p : d->d_blk_image)
		if (p)
			(void)hipFree(p);
	if (d->d_l12_sid)
		(void)hipFree(d->d_l12_sid);
	if (d->d_prep_action)
		(void)hipFree(d->d_prep_action);
	delete d;
}

int lw_decoder_device(const lw_decoder *d)
{
	return d ? d->device : -1;
}

// Several decoders on ONE GPU.  k_long / k_long10 / k_long12 / k_mix* take a whole compute unit per workgroup (152 KB of its
// 160 KB LDS and all of its vector registers), so next to another tenant's long-running kernel (k_entropy: 196 us per 4096
// packets) a 15 us launch waits for whole CUs to drain: 130-150 us.  With a share, the streams made for this
// decoder carry a CU mask and the planner sizes this decoder's launches for its own CUs.  A queue's mask bit i is CU i / 8 of
// XCD i % 8, and a queue must keep CUs on EVERY XCD -- its workgroups go round the XCDs whatever the mask says; a mask that
// empties an XCD is ignored as a whole (both measured: tools/micro/cumask.hip, profiles/r06_cumask.txt) -- so a share is the
// same CUs [n j / k, n (j + 1) / k) of each XCD's n = 32, not whole XCDs: tenants never meet on a CU, they do share the L2s.
int lw_decoder_set_cu_share(lw_decoder *d, unsigned part, unsigned parts)
{
	if (!d || parts == 0 || part >= parts)
		return LW_ERR_NULL_ARG;
	const unsigned per_xcd = (unsigned)d->n_cus_device / LW_XCDS;
	if (parts > per_xcd)
		return LW_ERR_UNSUPPORTED;
	// the mask layout (bit i = CU i / 8 of XCD i % 8; a queue keeps CUs on every XCD) was measured on MI355X = gfx950 with eight
	// XCDs of equally many CUs (tools/micro/cumask.hip, profiles/r06_cumask.txt): anything else gets no share
	if (parts > 1 && (!d->is_gfx950 || (unsigned)d->n_cus_device % LW_XCDS != 0))
		return LW_ERR_UNSUPPORTED;
	// ... and none once this process has copied on the device's copier stream (a tenant's ring without a share): the two kinds of
	// stream in one process were seen to keep it from exiting (include/lewton_amd.h)
	if (parts > 1 && (lw_tenant_streams(d->device) & LW_TENANT_COPIER_STREAM))
		return LW_ERR_UNSUPPORTED;
	std::lock_guard<std::mutex> g(d->mu);
	d->cu_mask.clear();
	d->n_cus = d->n_cus_device;
	if (parts == 1)
		return LW_OK;
	d->cu_mask.assign((size_t)(d->n_cus_device + 31) / 32, 0u);
	int mine = 0;
	for (unsigned i = 0; i < per_xcd * LW_XCDS; i++)
		if ((i / LW_XCDS) * parts / per_xcd == part) {
			d->cu_mask[i / 32] |= 1u << (i % 32);
			mine++;
		}
	d->n_cus = std::max(1, mine);
	return LW_OK;
}

int lw_decoder_cu_count(const lw_decoder *d)
{
	return d ? d->n_cus : 0;
}

int lw_decoder_device_cu_count(const lw_decoder *d)
{
	return d ? d->n_cus_device : 0;
}

// Other decoders' rings run on this decoder's GPU as well (lw_sharder_create with a device named several times, or several
// lewton-style decoders in one process): the rings made for it AFTERWARDS hand their PCM copies to the device's copier
// thread and run their launches' kernels in launch order (lw_ring.cpp: Copier).
int lw_decoder_set_shared_device(lw_decoder *d, int on)
{
	if (!d)
		return LW_ERR_NULL_ARG;
	std::lock_guard<std::mutex> g(d->mu);
	d->shares_device = on != 0;
	return LW_OK;
}

int lw_decoder_shares_device(const lw_decoder *d)
{
	return d && d->shares_device;
}

} // extern "C"

// Which kinds of tenant stream this process has made on each device (include/lewton_amd.h, "Known hazard"): once set, never cleared
static std::atomic<int> g_tenant_streams[256];

int lw_tenant_streams(int device)
{
	return device >= 0 && device < 256 ? g_tenant_streams[device].load() : 0;
}

void lw_tenant_streams_note(int device, int kind)
{
	if (device >= 0 && device < 256)
		g_tenant_streams[device].fetch_or(kind);
}

// A stream for this decoder's launches, on the decoder's share of the device.  Without a share: a non-blocking stream.  With one:
// hipExtStreamCreateWithCUMask takes no flags and makes a DEFAULT (blocking) stream -- a CU-masked stream synchronises implicitly
// with the NULL stream, so any null-stream work in the process (a synchronous hipMemcpy, torch's legacy default stream) orders
// itself against every such tenant's launches; the library keeps its own null-stream calls off the rings' paths
// (lw_batch_device_status clears the edge flags on the batch's own stream).
hipError_t lw_decoder_stream_create(lw_decoder *d, hipStream_t *s)
{
	std::vector<uint32_t> mask;
	{
		std::lock_guard<std::mutex> g(d->mu);
		mask = d->cu_mask;
	}
	if (mask.empty())
		return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
	if (lw_tenant_streams(d->device) & LW_TENANT_COPIER_STREAM)
		return hipErrorNotSupported; // (lw_ring_create checks first and says LW_ERR_UNSUPPORTED)
	lw_tenant_streams_note(d->device, LW_TENANT_MASKED_STREAMS);
	return hipExtStreamCreateWithCUMask(s, (uint32_t)mask.size(), mask.data());
}

extern "C" {

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
	if (d->free_slots.empty() && lw_grow_state(d, d->state_cap + 1) != LW_OK)
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
		if (!lw_hip_ok(hipDeviceSynchronize(), "sync") ||
				!lw_hip_ok(hipMemcpy(d->d_state + (size_t)q->slot * per, d->d_state + (size_t)p->slot * per,
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
	if (int rc = lw_decoder_set_device(d))
		return rc;
	HIP_TRY(hipDeviceSynchronize());
	const float *src = d->d_state + ((size_t)p->slot * 2 + p->parity) * d->T.state_stride;
	HIP_TRY(hipMemcpy2D(dst, p->len * sizeof(float), src, d->T.state_chan_stride * sizeof(float), p->len * sizeof(float),
				d->T.ch, hipMemcpyDeviceToHost));
	return LW_OK;
}

int lw_decoder_supports_device_entropy(const lw_decoder *d, const char **why)
{
	if (why)
		*why = d ? d->dev_entropy_why.c_str() : "";
	return d && d->dev_entropy_ok ? 1 : 0;
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

