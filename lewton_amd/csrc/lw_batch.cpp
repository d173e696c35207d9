// Batches of the MI355X audio-packet decode path (product code, compiled with hipcc): pinned staging, the threaded host
// entropy stage, the work plan of the kernels (window geometry, state hand-over, task lists), upload and launches, debug
// taps.  Implements the lw_batch_* part of include/lewton_amd.h.
#include "lw_internal.hpp"
#include "lw_pool.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>

// packets a worker claims at a time: small enough that 64 threads share a 4096-packet batch evenly to the end
#define LW_ENTROPY_CHUNK 4

// floats per raw edge in the edge buffer (lw_fast.hpp): blocksize_0 / 4 = 8 x the lanes of a short block of k_short (64 next to k_long)
static inline size_t lw_edge_values(const lw_decoder *d)
{
	return d->blkp[0].eligible && d->blkp[0].lanes <= 16 ? 8u * d->blkp[0].lanes : LW_EDGE_VALUES;
}

extern "C" {

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
	if (lw_decoder_set_device(d)) {
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
	const size_t o_recs = slice(rec_b), o_floor = slice(fl_b), o_items = slice((max_packets + max_packets / 2 + 64) * sizeof(LwFastItem));
	const size_t o_halo = slice(max_packets * sizeof(LwFastItem)), o_gen = slice(4 * max_packets * sizeof(uint32_t));
	const size_t o_ola = slice(max_packets * sizeof(LwOlaDesc));
	const size_t o_tasks = slice(max_packets * ch * sizeof(LwGenTask));
	// k_short: at most two slots per short packet (a recomputed predecessor in front of it) plus one per long block with a
	// short left slope, in tasks of LW_SHORT_SLOTS
	size_t max_tasks[2] = {0, 0}, o_slots[2] = {0, 0};
	for (int cls = 0; cls < 2; cls++)
		if (d->blkp[cls].eligible) {
			// (sized in slots: at most three per packet plus one task of padding, whatever the passes per task)
			const size_t per_wave = std::max<size_t>(1, 64 / d->blkp[cls].lanes); // (k_big: one slot per pass)
			max_tasks[cls] = 3 * max_packets + 2 * per_wave * d->blkp[cls].passes;
			o_slots[cls] = slice(max_tasks[cls] * sizeof(LwShortSlot));
		}
	const size_t o_res = slice(res_b), o_fc = d->any_floor0 ? slice(res_b) : 0;
	b->slab_bytes = off;
	bool ok = lw_hip_ok(hipHostMalloc((void **)&b->h_slab, off), "hipHostMalloc(batch records)") &&
		lw_hip_ok(hipMalloc((void **)&b->d_slab, off), "hipMalloc(batch records)");
	if (ok) {
		auto H = [&](size_t o) { return b->h_slab + o; };
		auto D = [&](size_t o) { return b->d_slab + o; };
		b->h_recs = (LwPacketRec *)H(o_recs), b->d_recs = (LwPacketRec *)D(o_recs);
		b->h_floor = (uint16_t *)H(o_floor), b->d_floor = (uint16_t *)D(o_floor);
		b->h_items = (LwFastItem *)H(o_items), b->d_items = (LwFastItem *)D(o_items);
		b->h_halo_items = (LwFastItem *)H(o_halo), b->d_halo_items = (LwFastItem *)D(o_halo);
		b->h_gen = (uint32_t *)H(o_gen), b->d_gen = (uint32_t *)D(o_gen);
		b->h_ola = (LwOlaDesc *)H(o_ola), b->d_ola = (LwOlaDesc *)D(o_ola);
		b->h_tasks = (LwGenTask *)H(o_tasks), b->d_tasks = (LwGenTask *)D(o_tasks);
		for (int cls = 0; cls < 2; cls++)
			if (max_tasks[cls]) {
				b->h_slots[cls] = (LwShortSlot *)H(o_slots[cls]), b->d_slots[cls] = (LwShortSlot *)D(o_slots[cls]);
				b->max_tasks[cls] = max_tasks[cls];
			}
		b->h_res = (float *)H(o_res), b->d_res = (float *)D(o_res);
		if (d->any_floor0)
			b->h_fcurve = (float *)H(o_fc), b->d_fcurve = (float *)D(o_fc);
	}
	if (ok) { // the device error word (lw_batch_device_status)
		ok = lw_hip_ok(hipHostMalloc((void **)&b->h_err, 64, hipHostMallocMapped), "hipHostMalloc(error word)") &&
			lw_hip_ok(hipHostGetDevicePointer((void **)&b->d_err, b->h_err, 0), "hipHostGetDevicePointer(error word)");
		if (ok)
			*b->h_err = 0;
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
	if (b->h_pk)
		(void)hipHostFree(b->h_pk);
	if (b->h_pool)
		(void)hipHostFree(b->h_pool);
	if (b->h_err)
		(void)hipHostFree(b->h_err);
	void *ent[] = {b->d_pk, b->d_pool};
	for (void *p : ent)
		if (p)
			(void)hipFree(p);
	void *dev[] = {b->d_slab, b->d_edge, b->d_decoupled, b->d_floor_alt, b->d_td, b->d_tap, b->d_out, b->d_halo};
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

void lw_debug_batch_set_mix(lw_batch *b, int mode)
{
	if (b)
		b->mix_mode = mode;
}

void lw_debug_batch_set_long10(lw_batch *b, int mode)
{
	if (b)
		b->l10_mode = mode;
}

void lw_debug_batch_break_mix(lw_batch *b, unsigned spin)
{
	if (b)
		b->mix_break_spin = spin;
}

static std::atomic<unsigned> g_break_mix_spin{0}; // lw_debug_break_mix: the same for every batch launched while it is set

void lw_debug_break_mix(unsigned spin)
{
	g_break_mix_spin.store(spin);
}

/* After the launches of lw_batch_synth have COMPLETED (the caller has synchronised the stream): LW_OK, or LW_ERR_DEVICE when a
 * kernel raised the batch's device error word -- the PCM of this batch must not be used.  Clears the word and what the failed
 * launch may have left behind (k_mix's edge flags), so that the batch can be launched again. */
int lw_batch_device_status(lw_batch *b)
{
	if (!b)
		return LW_ERR_NULL_ARG;
	if (!b->h_err || __atomic_load_n(b->h_err, __ATOMIC_ACQUIRE) == 0)
		return LW_OK;
	__atomic_store_n(b->h_err, 0u, __ATOMIC_RELEASE);
	lw_set_device_error("k_mix: a short block's wave never saw the raw edges of its long neighbours (grid not resident?); batch dropped");
	if (lw_decoder_set_device(b->dec) == LW_OK && b->d_edge) {
		// on the stream the failed launch ran on (completed by now; a re-launch on it comes behind this) -- not the NULL stream, with
		// which CU-masked (blocking) streams of every tenant in the process would synchronise
		const size_t entries = b->max_packets * 2 * b->dec->T.ch;
		(void)lw_hip_ok(hipMemsetAsync(b->d_edge + entries * lw_edge_values(b->dec), 0, entries * sizeof(uint32_t), (hipStream_t)b->last_stream),
				"hipMemsetAsync(edge flags)");
		(void)lw_hip_ok(hipStreamSynchronize((hipStream_t)b->last_stream), "hipStreamSynchronize(edge flags)");
	}
	for (size_t i = 0; i < b->n; i++)
		if (b->results[i].status == LW_OK) {
			b->results[i].status = LW_ERR_DEVICE;
			b->results[i].n_samples = 0;
		}
	return LW_ERR_DEVICE;
}

void lw_debug_batch_set_rounds(lw_batch *b, int rounds)
{
	if (b)
		b->forced_rounds = rounds > 0 ? rounds : 0;
}

void lw_debug_batch_set_halo(lw_batch *b, int mode)
{
	if (b)
		b->halo_mode = mode;
}

/* Entropy stage on the device (lw_dev_entropy.h, k_entropy): lw_batch_entropy then only reads the packet prologues, copies
 * the packets into pinned staging and plans the batch; floors and residues are decoded by one GPU wave per packet. */
int lw_batch_set_entropy_on_device(lw_batch *b, int on)
{
	if (!b)
		return LW_ERR_NULL_ARG;
	if (!on) {
		b->dev_entropy = false;
		return LW_OK;
	}
	lw_decoder *d = b->dec;
	if (!d->dev_entropy_ok)
		return LW_ERR_UNSUPPORTED; // lw_decoder_supports_device_entropy says why
	if (int rc = lw_decoder_set_device(d))
		return rc;
	if (!b->h_pk) {
		HIP_TRY(hipHostMalloc((void **)&b->h_pk, b->max_packets * sizeof(LwEntPacket)));
		HIP_TRY(hipMalloc((void **)&b->d_pk, b->max_packets * sizeof(LwEntPacket)));
	}
	b->dev_entropy = true;
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

uint64_t lw_batch_state_bytes(const lw_batch *b)
{
	return b ? b->state_bytes : 0;
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
	// The device entropy stage grows its pinned / device packet pool in here, and this function is entered from threads the
	// library did not create on the decoder's GPU (the Ogg reader's staging thread, any caller of lw_ring_stage): a fresh
	// thread's current HIP device is 0, so the pool of a decoder on GPU N would land on GPU 0 (and the hipDeviceSynchronize
	// before it is freed would wait for the wrong device).
	if (b->dev_entropy)
		if (int rc = lw_decoder_set_device(d))
			return rc;

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
			if (b->dev_entropy)
				b->h_pk[i].start_bit = (uint8_t)br.pos;
		}
	}
	b->res_floats = res_off;
	b->max_n = max_n;
	if (b->dev_entropy) {
		// entropy stage on the device: the packets go up as they are, word-aligned and followed by at least 3 zero words (the
		// device reader requests its window one refill ahead); for the eligible setups the prologue alone decides a packet's status
		size_t words = 0;
		for (size_t i = 0; i < n; i++) {
			if (b->status[i] != LW_OK)
				continue;
			b->h_pk[i].word_off = (uint32_t)words;
			b->h_pk[i].len = (uint32_t)pkts[i].len;
			words += (pkts[i].len + 3) / 4 + 3;
		}
		if (words > b->pool_cap_words) {
			const size_t cap = words + words / 2 + 1024;
			if (b->d_pool) {
				if (!lw_hip_ok(hipDeviceSynchronize(), "sync before growing the packet pool"))
					return LW_ERR_DEVICE;
				(void)hipFree(b->d_pool);
				b->d_pool = nullptr;
			}
			if (b->h_pool)
				(void)hipHostFree(b->h_pool);
			b->h_pool = nullptr;
			b->pool_cap_words = 0;
			if (!lw_hip_ok(hipHostMalloc((void **)&b->h_pool, cap * 4), "hipHostMalloc(packet pool)") ||
					!lw_hip_ok(hipMalloc((void **)&b->d_pool, cap * 4), "hipMalloc(packet pool)"))
				return LW_ERR_DEVICE;
			b->pool_cap_words = cap;
		}
		b->pool_words = words;
		for (size_t i = 0; i < n; i++) {
			if (b->status[i] != LW_OK)
				continue;
			uint8_t *dst = (uint8_t *)(b->h_pool + b->h_pk[i].word_off);
			const size_t len = pkts[i].len, padded = ((len + 3) / 4 + 3) * 4;
			if (len)
				std::memcpy(dst, pkts[i].data, len);
			std::memset(dst + len, 0, padded - len);
		}
	}

	// pass 2 (parallel): entropy decode straight into the pinned staging buffers
	unsigned nt = n_threads > 0 ? (unsigned)n_threads : lw::default_host_threads();
	nt = (unsigned)std::min<size_t>(nt, std::max<size_t>(1, n / 8));
	alignas(128) std::atomic<size_t> next{0}; // (its own cache line: every worker adds to it once per LW_ENTROPY_CHUNK packets)
	alignas(128) char next_pad[8] = {0};
	(void)next_pad;
	auto worker = [&]() {
		// scratch vectors keep their capacity from batch to batch (pool threads are persistent)
		static thread_local lw::EntropyScratch scr;
		for (;;) {
			const size_t i0 = next.fetch_add(LW_ENTROPY_CHUNK);
			if (i0 >= n)
				break;
			for (size_t i = i0; i < std::min(n, i0 + LW_ENTROPY_CHUNK); i++) {
				if (b->status[i] != LW_OK)
					continue;
				LwPacketRec &r = b->h_recs[i];
				b->status[i] = lw::entropy_decode(id, s, pkts[i].data, pkts[i].len, b->prologues[i],
						b->h_floor + r.floor_off, (unsigned)fstride, b->h_res + r.res_off, scr, nullptr,
						b->h_fcurve ? b->h_fcurve + r.res_off : nullptr);
			}
		}
	};
	if (!b->dev_entropy)
		lw::entropy_pool().run(nt, worker);

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
	uint64_t alg = 0, state = 0;
	const size_t esz = lw_elem_size(b->fmt);
	b->has_generic = b->has_fast = false;
	b->n_gen_small = b->n_gen_large = b->n_gen_ola = 0;
	b->has_tdonly = false;
	b->n_prep = 0;
	for (int cls = 0; cls < 2; cls++) {
		b->blk_idx[cls].clear();
		b->blk_slot[cls].clear();
		b->n_tasks[cls] = 0;
	}
	// k_short<L> covers the short blocks (class 0) and, where k_long does not apply, the long blocks with two long slopes (class 1)
	const bool blk_ok[2] = {d->blkp[0].eligible && !b->force_generic, d->blkp[1].eligible && !b->force_generic};
	// blocksize_1 = 10: the class-1 blocks go to k_long10 on k_long's work list (below) instead of k_short<32>'s slots
	b->use_l10 = blk_ok[1] && d->blkp[1].lanes == 32 && b->l10_mode != 0 && !d->fast.eligible &&
		d->blkp[1].units.size() <= LW_FAST_WAVES;
	b->l10_cls = 1;
	if (!b->use_l10 && blk_ok[0] && d->blkp[0].lanes == 32 && id.bs0 == id.bs1 && b->l10_mode != 0 && !d->fast.eligible &&
			d->blkp[0].units.size() <= LW_FAST_WAVES) {
		// blocksize_0 = blocksize_1 = 10 and no mode with the block flag (what libvorbis writes for equal block sizes is ONE mode,
		// flag 0): every block is a 1024-point block with two full slopes -- k_long10's shape, on the "short" class's units and image
		bool any_long = false;
		for (const lw::Mode &m : s.modes)
			any_long = any_long || m.blockflag;
		if (!any_long) {
			b->use_l10 = true;
			b->l10_cls = 0;
		}
	}
	// blocksize_1 = 12: likewise to k_long12 (one wave per channel: the split units) instead of k_big<12>
	b->use_l12 = blk_ok[1] && d->blkp[1].lanes == 128 && !d->blkp[1].units_split.empty() && !d->blkp[1].image.empty() &&
		b->l10_mode != 0 && !d->fast.eligible;
	// short blocks of 256 points next to k_long, of 256 / 512 points next to k_long10 / k_long12: long blocks with short slopes stay
	// in the long-block kernel's EDGE form (lw_fast.hpp)
	const bool short_ok10 = ((b->use_l10 && b->l10_cls == 1) || b->use_l12) && b->l10_mode != 1 && blk_ok[0] && (d->blkp[0].bs == 8 || d->blkp[0].bs == 9);
	const bool short_ok = (d->fast.eligible && blk_ok[0] && d->blkp[0].bs == 8) || short_ok10;
	// ... and where they have none (short blocks of another size, or on the generic kernels): the wave kernel's time-domain block
	// and the generic overlap-add (LW_RF_TDONLY); l10_mode 1 (test hook) sends such blocks to the generic kernels altogether
	const bool td_ok10 = ((b->use_l10 && b->l10_cls == 1) || b->use_l12) && b->l10_mode != 1 && !short_ok10 && id.bs0 != id.bs1;
	b->edge_mode = false; // set when the batch has a long block with a short slope (an all-(1,1) batch keeps the plain k_long)
	const uint32_t n0h = (1u << id.bs0) / 2, n1h = (1u << id.bs1) / 2;
	// equal block sizes with a flagged mode: one block shape (full slopes on both sides whatever the flags say), one class -- every
	// packet is planned as a long block between long blocks (lw_unified_classes)
	const bool unified = lw::lw_unified_classes(id, s);
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
		lw::Prologue &p = b->prologues[i];
		if (unified)
			p.blockflag = p.prev_flag = p.next_flag = true; // (same window geometry, same tables: window_info below)
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
				state += (uint64_t)ch * pw->len * 4; // the stream's stored right part comes in from the state pool
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
			if (short_ok) {
				// the stream's short blocks run through k_short: a long block with a short slope keeps everything but the
				// 128-sample overlap with its short neighbour in k_long (LW_XF_EDGE_*).  A stored right part of another length
				// than the slope's (legal, never produced by an encoder) sends the packet to the generic kernels.
				if (r.prev == -1 || r.plen == (p.prev_flag ? n1h : n0h)) {
					r.flags |= LW_RF_FAST;
					r.xflags = (uint8_t)((p.prev_flag ? 0u : LW_XF_EDGE_L) | (p.next_flag ? 0u : LW_XF_EDGE_R));
					b->edge_mode |= r.xflags != 0;
					b->fast_idx.push_back((uint32_t)i);
					b->fast_slot.push_back((uint32_t)pw->slot);
				}
			} else {
				r.flags |= LW_RF_FAST;
				if (!(p.prev_flag && p.next_flag && (r.prev == -1 || r.plen == n1h))) {
					r.flags |= LW_RF_TDONLY;
					b->has_tdonly = true;
				}
				b->fast_idx.push_back((uint32_t)i);
				b->fast_slot.push_back((uint32_t)pw->slot);
			}
		} else if (blk_ok[1] && p.blockflag && (d->blkp[1].short_mode_mask[p.mode >> 3] & (1u << (p.mode & 7))) && td_ok10 &&
				!(p.prev_flag && p.next_flag && (r.prev == -1 || r.plen == n1h))) {
			// a long block of k_long10 / k_long12 with a short slope (or a stored right part of another length) where the short blocks
			// have no edge form: floor, inverse coupling and transform stay in the wave kernel, which writes the whole time-domain block;
			// k_ola_generic does the window / overlap-add / state (LW_RF_TDONLY, as next to k_long)
			r.flags |= LW_RF_FAST | LW_RF_TDONLY;
			b->has_tdonly = true;
			b->blk_idx[1].push_back((uint32_t)i);
			b->blk_slot[1].push_back((uint32_t)pw->slot);
		} else if (blk_ok[1] && p.blockflag && (d->blkp[1].short_mode_mask[p.mode >> 3] & (1u << (p.mode & 7))) &&
				(short_ok10 ? (r.prev == -1 || r.plen == (p.prev_flag ? n1h : n0h))
				            : (p.prev_flag && p.next_flag && (r.prev == -1 || r.plen == n1h)))) {
			// a long block of a stream without k_long: k_short<L> (class 1) or k_long10, two long slopes; other window shapes go to
			// the generic kernels -- unless the stream's short blocks run through k_short next to k_long10 / k_long12: then as above (EDGE)
			r.flags |= LW_RF_FAST;
			if (short_ok10) {
				r.xflags = (uint8_t)((p.prev_flag ? 0u : LW_XF_EDGE_L) | (p.next_flag ? 0u : LW_XF_EDGE_R));
				b->edge_mode |= r.xflags != 0;
			}
			b->blk_idx[1].push_back((uint32_t)i);
			b->blk_slot[1].push_back((uint32_t)pw->slot);
		} else if (blk_ok[0] && !p.blockflag && (d->blkp[0].short_mode_mask[p.mode >> 3] & (1u << (p.mode & 7))) &&
				(r.prev == -1 || r.plen == n0h)) {
			r.flags |= LW_RF_FAST; // (LW_RF_FAST without LW_RF_LONG: a short block of k_short<L>, class 0)
			b->blk_idx[0].push_back((uint32_t)i);
			b->blk_slot[0].push_back((uint32_t)pw->slot);
		}
		if ((r.flags & LW_RF_FAST) && d->h_prep_mode[p.mode]) // its stream shape goes through the canonicalising pre-pass first
			b->h_gen[3 * b->max_packets + b->n_prep++] = (uint32_t)i;
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
			state += (uint64_t)ch * pw->len * 4; // ... and the new one goes out to it
			const uint8_t outp = pw->parity ^ 1;
			if (outp)
				r.flags |= LW_RF_PARITY_OUT;
			pw->parity = outp;
		}
		b->slot_last[pw->slot] = -1;
	}
	b->out_elems = out_off;
	b->alg_bytes = alg;
	b->state_bytes = state;
	if (!b->fast_idx.empty() || !b->blk_idx[0].empty() || !b->blk_idx[1].empty())
		for (size_t i = 0; i < n; i++) { // generic successors of packets of the specialised kernels read the td block
			const LwPacketRec &r = b->h_recs[i];
			const bool ola_generic = !(r.flags & LW_RF_FAST) || (r.flags & LW_RF_TDONLY);
			if (!(r.flags & LW_RF_SKIP) && ola_generic && r.prev >= 0 && (b->h_recs[r.prev].flags & LW_RF_FAST))
				b->h_recs[r.prev].flags |= LW_RF_WRITE_TD;
		}
	const int wave_cls = b->use_l12 ? 1 : b->l10_cls; // the block class whose packets k_long10 / k_long12 take
	// what marks a packet of the specialised wave-pipeline kernel in rec.flags (the single-class case: its blocks carry no block flag)
	const uint32_t fast_mark = (b->use_l10 && b->l10_cls == 0) ? (uint32_t)LW_RF_FAST : (uint32_t)(LW_RF_FAST | LW_RF_LONG);
	if (b->use_l10 || b->use_l12) { // (k_long10 / k_long12: their packets on the specialised kernel's lists from here on)
		b->fast_idx.swap(b->blk_idx[wave_cls]);
		b->fast_slot.swap(b->blk_slot[wave_cls]);
		b->blk_idx[wave_cls].clear();
		b->blk_slot[wave_cls].clear();
	}
	// ---- slots of k_short<L> (lw_fast.hpp), per block class: the blocks sorted by stream so that consecutive blocks of a stream sit
	// in consecutive slots of a wave and hand their right part over through LDS; a block whose predecessor of the same class is
	// not the slot in front of it (wave boundary) gets that predecessor recomputed in the slot in front (LW_SS_HALO).
	// With 256-point short blocks next to k_long (short_ok): a short block followed by a long one with a short left slope also
	// does that block's first 128 samples; such a long block whose predecessor is NOT a block of k_short gets a slot of its own
	// that only carries the stored right part (LW_SS_EDGE).
	if ((blk_ok[0] || blk_ok[1]) && (!b->blk_idx[0].empty() || !b->blk_idx[1].empty() || (short_ok && !b->fast_idx.empty()))) {
		b->succ.assign(n, -1);
		for (size_t i = 0; i < n; i++)
			if (!(b->h_recs[i].flags & LW_RF_SKIP) && b->h_recs[i].prev >= 0)
				b->succ[b->h_recs[i].prev] = (int32_t)i;
		const bool klong = d->fast.eligible || b->use_l10 || b->use_l12;
		// a packet of k_long / a block of k_short<L> of class `cls`
		auto is_long_fast = [&](const LwPacketRec &r) {
			return klong && (r.flags & (fast_mark | LW_RF_SKIP)) == fast_mark;
		};
		auto is_blk = [&](const LwPacketRec &r, int cls) {
			if ((r.flags & (LW_RF_FAST | LW_RF_SKIP)) != LW_RF_FAST || is_long_fast(r))
				return false;
			return ((r.flags & LW_RF_LONG) ? 1 : 0) == cls;
		};
		struct Ev {
			uint32_t slot, idx;
		};
		std::vector<Ev> ev;
		for (int cls = 0; cls < 2; cls++) {
			if (!blk_ok[cls] || (cls == wave_cls && (b->use_l10 || b->use_l12)))
				continue;
			// passes per wave: more slots per recomputed predecessor -- but only while the launch keeps the waves the chip holds
			// at a time (five per CU: LDS) (a wave's passes run one after the other: 4096 blocks of 1024 points in 820 waves of 3
			// passes were slower than in 4096 waves of one)
			const bool big = d->blkp[cls].lanes > 64; // k_big: a workgroup per task, one slot per pass, at least two (halo + block)
			const size_t per_wave = big ? 1 : 64 / d->blkp[cls].lanes;
			uint32_t passes = big ? 2 : 1;
			if (big) {
				// k_big holds three waves per SIMD (its registers): 12 waves = 12 / (lanes / 64) workgroups per CU.  The fewest passes
				// with which the whole launch is resident at once (the launch lasts passes x ~15 us whatever the number of
				// workgroups, and a second round of them would double it; measured: profiles/r04_k_big_variants.txt); a task of p
				// slots takes p - 1 blocks when it starts with a recomputed predecessor
				const size_t cap = (size_t)std::max(1, d->n_cus) * (12 / (d->blkp[cls].lanes / 64));
				while (passes < d->blkp[cls].passes && (b->blk_idx[cls].size() + passes - 2) / (passes - 1) > cap)
					passes++;
			} else
			while (passes < d->blkp[cls].passes &&
					b->blk_idx[cls].size() / (per_wave * (passes + 1) - 1) >= 5 * (size_t)std::max(1, d->n_cus))
				passes++;
			b->blk_passes[cls] = passes;
			const size_t per_task = per_wave * passes;
			// events: the class's blocks, and (class 0, short_ok) long blocks with a short left slope whose predecessor is not a
			// short block of k_short
			ev.clear();
			for (size_t k = 0; k < b->blk_idx[cls].size(); k++)
				ev.push_back(Ev{b->blk_slot[cls][k], b->blk_idx[cls][k]});
			if (cls == 0 && short_ok)
				for (size_t k = 0; k < b->fast_idx.size(); k++) {
					const LwPacketRec &r = b->h_recs[b->fast_idx[k]];
					if ((r.xflags & LW_XF_EDGE_L) && r.prev != -1 && !(r.prev >= 0 && is_blk(b->h_recs[r.prev], 0)))
						ev.push_back(Ev{b->fast_slot[k], b->fast_idx[k]});
				}
			if (ev.empty())
				continue;
			std::stable_sort(ev.begin(), ev.end(), [](const Ev &a, const Ev &c) { return a.slot != c.slot ? a.slot < c.slot : a.idx < c.idx; });
			LwShortSlot *slots = b->h_slots[cls];
			size_t n_slots = 0; // slots used so far (tasks are consecutive groups of per_task)
			// capacity (lw_batch_create sizes the list for three slots per packet plus padding): every event below takes at most two
			// slots and at most one task's worth of padding in front of them
			const size_t slot_cap = b->max_tasks[cls];
			bool slots_full = false;
			auto room = [&](size_t want) { // the next `want` slots lie in one task
				const size_t used = n_slots % per_task;
				if (n_slots + per_task + want > slot_cap) {
					slots_full = true;
					return;
				}
				if (used + want > per_task)
					while (n_slots % per_task) {
						std::memset(&slots[n_slots], 0, sizeof(LwShortSlot));
						slots[n_slots].next_edge = 0xFFFFFFFFu;
						slots[n_slots].state_out = -1;
						n_slots++;
					}
			};
			auto blank = [&](uint32_t idx, uint8_t kind) -> LwShortSlot & {
				LwShortSlot &sl = slots[n_slots++];
				std::memset(&sl, 0, sizeof(sl));
				const LwPacketRec &r = b->h_recs[idx];
				sl.res_off = r.res_off;
				sl.floor_off = r.floor_off;
				sl.out_off = r.out_off;
				sl.state_out = -1;
				sl.next_edge = 0xFFFFFFFFu;
				sl.kind = kind;
				sl.prev_kind = LW_SP_NONE;
				sl.pkt = idx;
				return sl;
			};
			// where the stored right part in front of packet `r` lives when it is not a slot of this launch
			auto outside_prev = [&](const LwPacketRec &r, LwShortSlot &sl) {
				if (r.prev <= -2) {
					sl.prev_kind = LW_SP_STATE;
					sl.prev_arg = (uint32_t)(-(r.prev + 2));
					sl.flags |= r.flags & LW_RF_PARITY_IN;
				} else if (r.prev >= 0) {
					const LwPacketRec &pr = b->h_recs[r.prev];
					if (short_ok && is_long_fast(pr)) { // (its right slope is short: the stored part has the short slope's length)
						sl.prev_kind = LW_SP_EDGE;
						sl.prev_arg = (uint32_t)r.prev;
					} else { // generic kernels, or k_long<TD>: the whole time-domain block is in td
						sl.prev_kind = LW_SP_TD;
						sl.prev_arg = 2u * pr.res_off + pr.rs;
						sl.prev_stride = (uint16_t)(1u << pr.bs);
					}
				}
			};
			auto set_next = [&](uint32_t idx, LwShortSlot &sl) { // the long successor with a short left slope, if any
				const int32_t nx = b->succ[idx];
				if (nx < 0 || !short_ok)
					return;
				const LwPacketRec &nr = b->h_recs[nx];
				if (is_long_fast(nr) && (nr.xflags & LW_XF_EDGE_L)) {
					sl.next_edge = (uint32_t)nx;
					sl.next_out = nr.out_off;
					sl.next_m = (uint32_t)(nr.rs - nr.ls);
				}
			};
			int64_t last_pkt = -1; // packet of the slot placed last (a block or a halo), -1 after anything else
			for (const Ev &e : ev) {
				if (slots_full)
					break;
				const LwPacketRec &r = b->h_recs[e.idx];
				if (is_long_fast(r)) { // LW_SS_EDGE
					room(1);
					if (slots_full)
						break;
					LwShortSlot &sl = blank(e.idx, LW_SS_EDGE);
					outside_prev(r, sl);
					sl.next_edge = e.idx;
					sl.next_out = r.out_off;
					sl.next_m = (uint32_t)(r.rs - r.ls);
					last_pkt = -1;
					continue;
				}
				const bool pred_same = r.prev >= 0 && is_blk(b->h_recs[r.prev], cls);
				bool lane = pred_same && last_pkt == (int64_t)r.prev && (n_slots % per_task) != 0;
				if (pred_same && !lane) { // recompute the predecessor in the slot in front
					room(2);
					if (slots_full)
						break;
					blank((uint32_t)r.prev, LW_SS_HALO);
					lane = true;
				} else {
					room(1); // (a slot behind its predecessor in the same task: only the capacity check)
					if (slots_full)
						break;
				}
				LwShortSlot &sl = blank(e.idx, LW_SS_BLOCK);
				if (lane)
					sl.prev_kind = LW_SP_LANE;
				else
					outside_prev(r, sl);
				sl.state_out = r.state_out;
				sl.flags |= r.flags & LW_RF_PARITY_OUT;
				if (r.flags & LW_RF_WRITE_TD)
					sl.flags |= LW_SF_WRITE_TD;
				set_next(e.idx, sl);
				last_pkt = (int64_t)e.idx;
			}
			if (slots_full) { // (cannot happen while lw_batch_create's bound holds; never write past the list)
				lw_set_device_error("block kernel: slot list capacity exceeded");
				return LW_ERR_CAPACITY;
			}
			while (n_slots % per_task) { // pad the last task
				std::memset(&slots[n_slots], 0, sizeof(LwShortSlot));
				slots[n_slots].next_edge = 0xFFFFFFFFu;
				slots[n_slots].state_out = -1;
				n_slots++;
			}
			b->n_tasks[cls] = n_slots / per_task;
		}
	}
	// tasks of the short-block transform kernel: record + the per-channel look-ups, one load on the device
	for (uint32_t t = 0; t < b->n_gen_small; t++) {
		const LwPacketRec &r = b->h_recs[b->h_gen[t]];
		for (uint32_t c = 0; c < ch; c++) {
			LwGenTask &g = b->h_tasks[(size_t)t * ch + c];
			g.rec = r;
			g.c = (uint8_t)c;
			g.fl = d->h_mode_floor[(size_t)r.mode * ch + c];
			g.F = d->h_floor_F[g.fl];
			g.partner = d->T.pair_coupling ? d->h_mode_partner[(size_t)r.mode * ch + c] : (int8_t)-1;
			g.role = d->T.pair_coupling ? d->h_mode_role[(size_t)r.mode * ch + c] : (uint8_t)0;
			g.pad[0] = g.pad[1] = g.pad[2] = 0;
		}
	}
	// descriptors of the generic overlap-add tasks (records are final now: state hand-over and parities included)
	for (uint32_t t = 0; t < b->n_gen_ola; t++) {
		const LwPacketRec &r = b->h_recs[b->h_gen[2 * b->max_packets + t]];
		LwOlaDesc &o = b->h_ola[t];
		o.cur_off = 2u * r.res_off;
		o.out_off = r.out_off;
		o.state_out = r.state_out;
		o.n = (uint16_t)(1u << r.bs); // (8192-point blocks: 0x2000 fits)
		o.ls = r.ls;
		o.rs = r.rs;
		o.re = r.re;
		o.plen = r.plen;
		o.flags = (uint8_t)(r.flags & (LW_RF_SLOPE_BS1 | LW_RF_PARITY_OUT));
		o.pad = 0;
		if (r.prev == -1) {
			o.prev_kind = 0;
			o.prev_off = 0;
			o.prev_stride = 0;
		} else if (r.prev >= 0) {
			const LwPacketRec &pr = b->h_recs[r.prev];
			o.prev_kind = 1;
			o.prev_off = 2u * pr.res_off + pr.rs;
			o.prev_stride = (uint16_t)(1u << pr.bs);
		} else {
			const uint32_t slot = (uint32_t)(-(r.prev + 2)), par = (r.flags & LW_RF_PARITY_IN) ? 1u : 0u;
			o.prev_kind = 2;
			o.prev_off = (uint32_t)(((size_t)slot * 2 + par) * d->T.state_stride);
			o.prev_stride = (uint16_t)d->T.state_chan_stride;
		}
	}

	// ---- work plan of the specialised kernel: items sorted by stream so that consecutive packets of a
	// stream sit in consecutive items; a workgroup works through a chunk of rounds * per_round consecutive
	// items and hands right halves over in LDS; a predecessor outside the chunk is recomputed by the halo pre-pass
	b->n_items = b->n_halo_items = 0;
	b->has_fast = !b->fast_idx.empty();
	if (b->has_fast) {
		const size_t nf = b->fast_idx.size();
		b->fast_order.resize(nf);
		for (size_t k = 0; k < nf; k++)
			b->fast_order[k] = (uint32_t)k;
		if (!std::is_sorted(b->fast_slot.begin(), b->fast_slot.end())) // (callers usually list their streams one after the other)
			std::stable_sort(b->fast_order.begin(), b->fast_order.end(),
					[&](uint32_t a, uint32_t c) { return b->fast_slot[a] < b->fast_slot[c]; });
		const size_t n_fast_units = b->use_l10 ? d->blkp[b->l10_cls].units.size() : b->use_l12 ? d->blkp[1].units_split.size() : d->fast.units.size();
		const uint32_t per_round_max = LW_FAST_WAVES / (uint32_t)n_fast_units;
		uint32_t per_round = per_round_max, rounds = 1;
		// launch shape for `n_it` items: as few rounds per workgroup as two resident workgroups per CU allow -- small batches spread
		// over the whole chip; big batches get long chunks (LDS hand-over, few recomputed predecessors)
		auto shape = [&](size_t n_it) {
			per_round = per_round_max;
			const size_t per_pass = (size_t)per_round * std::max(1, d->n_cus);
			rounds = (uint32_t)std::min<size_t>(LW_FAST_MAX_ROUNDS, std::max<size_t>(1, (n_it + per_pass - 1) / per_pass));
			if (b->forced_rounds) { // lw_debug_batch_set_rounds (tests: hand-over paths across rounds and workgroups)
				rounds = (uint32_t)std::min<int>(LW_FAST_MAX_ROUNDS, b->forced_rounds);
			} else {
				// spread the packets evenly over the CUs: with fewer packets than one full round per CU (the long blocks of a mixed
				// short/long batch, a small batch: 1 117 long packets in chunks of 16 kept 186 of 256 CUs idle) or a packets-per-round
				// count that does not divide the batch (5.1 with three units per packet: 5 packets x 4 rounds = 20 per workgroup put
				// 4096 packets on 205 of 256 CUs) a workgroup takes fewer packets per round instead
				const size_t cus = (size_t)std::max(1, d->n_cus);
				per_round = (uint32_t)std::min<size_t>(per_round, std::max<size_t>(1, (n_it + cus * rounds - 1) / (cus * rounds)));
			}
			return per_round * rounds;
		};
		auto pkt_of = [&](size_t k) { return b->fast_idx[b->fast_order[k]]; };
		// does item k's previous right half come from another packet of this launch's kernel (then it has to sit in the item in front)?
		auto needs_pred = [&](const LwPacketRec &r) {
			return !(r.prev == -1 || (r.flags & LW_RF_TDONLY) || (r.xflags & LW_XF_EDGE_L)) && r.prev >= 0 &&
				(b->h_recs[r.prev].flags & fast_mark) == fast_mark;
		};
		uint32_t chunk = shape(nf);
		const uint32_t rounds_plain = rounds;
		// A chunk that starts inside a stream needs its predecessor's right half.  Round 6: the predecessor is recomputed IN the launch,
		// as an item of its own in front (no samples, no state: the form of a stream's first packet, whose right half goes to the next
		// wave through LDS) -- one more wave-slot per chunk start instead of a pre-pass launch in front of every such launch (ONE
		// stream x 4096 long packets: 255 chunk starts; the pre-pass cost 6 of the launch's 21 us).  The pre-pass (k_long<RIGHT_ONLY>,
		// LW_SRC_HALO) remains for forced launch shapes (tests), one-item chunks and batches where every other item would be one.
		size_t n_inline = 0;
		bool inline_halo = b->halo_mode != 0 && !b->forced_rounds;
		for (int iter = 0; inline_halo && iter < 4; iter++) {
			size_t pos = 0, nh = 0;
			int64_t last = -1;
			for (size_t k = 0; k < nf; k++) {
				const LwPacketRec &r = b->h_recs[pkt_of(k)];
				if (needs_pred(r) && (pos % chunk == 0 || last != (int64_t)r.prev)) {
					nh++;
					pos++;
				}
				pos++;
				last = (int64_t)pkt_of(k);
			}
			const uint32_t c2 = shape(nf + nh);
			// (not where the extra items would cost every workgroup another round -- a batch that fills the chip's waves exactly:
			// ONE stream x 8192 long packets took 37 us with them against 35 us with the pre-pass)
			if (c2 < 2 || nh * 3 > nf || nf + nh > b->max_packets + b->max_packets / 2 + 64 || rounds > rounds_plain) {
				inline_halo = false;
				chunk = shape(nf);
				break;
			}
			n_inline = nh;
			if (c2 == chunk)
				break;
			chunk = c2; // (another chunk length moves the chunk starts: count again)
			if (iter == 3) {
				inline_halo = false;
				chunk = shape(nf);
			}
		}
		if (!inline_halo)
			n_inline = 0;
		// a sparse launch (one round, at most half of a workgroup's waves in use) lasts as long as ONE wave's dependent chain:
		// split every channel pair over two waves (LW_UNIT_SPLIT_*) -- each half does one channel's floor, transform and samples
		b->fast_split = !b->use_l10 && !b->use_l12 && !b->forced_rounds && !b->has_tdonly && rounds == 1 && d->fast.units_split.size() > d->fast.units.size() &&
			per_round * d->fast.units_split.size() <= LW_FAST_WAVES;
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
			it.flags = (uint8_t)((r.flags & (LW_RF_PARITY_IN | LW_RF_PARITY_OUT | LW_RF_WRITE_TD | LW_RF_TDONLY)) |
					(r.xflags & (LW_XF_EDGE_L | LW_XF_EDGE_R)));
			if (r.prev == -1)
				it.flags |= LW_IF_SILENT;
			if (r.flags & LW_RF_TDONLY) {
				it.flags |= LW_RF_WRITE_TD; // left AND right half go to the td block
				it.state_out = -1;          // the state slot is written by k_ola_generic
			}
			it.pkt = idx;
		};
		size_t pos = 0; // items placed so far
		for (size_t k = 0; k < nf; k++) {
			const uint32_t idx = pkt_of(k);
			const LwPacketRec &r = b->h_recs[idx];
			const bool pred_in_front = pos % chunk != 0 && pos > 0 && b->h_items[pos - 1].pkt == (uint32_t)r.prev;
			if (inline_halo && needs_pred(r) && !pred_in_front) {
				// the predecessor once more, for its right half only: no previous window (no samples), no state, no edges, no td block
				LwFastItem &h = b->h_items[pos++];
				fill(h, (uint32_t)r.prev);
				h.state_out = -1;
				h.src_kind = LW_SRC_NONE;
				h.flags = LW_IF_SILENT;
				b->fast_dense = 0;
			}
			LwFastItem &it = b->h_items[pos];
			fill(it, idx);
			if (it.res_off != (uint32_t)(pos * ch * n1h) || it.floor_off != (uint32_t)(pos * ch * fstride))
				b->fast_dense = 0;
			if (r.prev == -1 || (r.flags & LW_RF_TDONLY) || (r.xflags & LW_XF_EDGE_L)) {
				it.src_kind = LW_SRC_NONE; // (a TD-only packet is overlapped later, by k_ola_generic; a short left slope by k_short)
			} else if (r.prev <= -2) {
				it.src_kind = LW_SRC_STATE;
				it.src_arg = (uint32_t)(-(r.prev + 2));
			} else if ((b->h_recs[r.prev].flags & fast_mark) == fast_mark) {
				if ((pos % chunk) != 0 && b->h_items[pos - 1].pkt == (uint32_t)r.prev) {
					it.src_kind = LW_SRC_LDS;
					b->h_items[pos - 1].flags |= LW_IF_NEXT_LDS;
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
			pos++;
		}
		b->n_items = pos;
		b->n_inline_halo = inline_halo ? pos - nf : 0;
		(void)n_inline;
	}
	return LW_OK;
}

int lw_batch_upload(lw_batch *b, void *hip_stream)
{
	if (!b)
		return LW_ERR_NULL_ARG;
	if (int rc = lw_decoder_set_device(b->dec))
		return rc;
	hipStream_t st = (hipStream_t)hip_stream;
	const size_t ch = b->dec->T.ch;
	if (b->n == 0)
		return LW_OK;
	b->ent_done = false;
	if (b->h_err) // a new upload starts clean: an error nobody asked about (lw_batch_synth without lw_batch_device_status) is not the next launch's
		__atomic_store_n(b->h_err, 0u, __ATOMIC_RELEASE);
	if (b->d_edge) { // k_mix's flags: an edge written for a reader that a NEW plan no longer has must not look ready
		const size_t entries = b->max_packets * 2 * ch;
		HIP_TRY(hipMemsetAsync(b->d_edge + entries * lw_edge_values(b->dec), 0, entries * sizeof(uint32_t), st));
	}
	if (b->dev_entropy) {
		HIP_TRY(hipMemcpyAsync(b->d_pk, b->h_pk, b->n * sizeof(LwEntPacket), hipMemcpyHostToDevice, st));
		if (b->pool_words)
			HIP_TRY(hipMemcpyAsync(b->d_pool, b->h_pool, b->pool_words * 4, hipMemcpyHostToDevice, st));
	}
	if (b->slab_bytes <= 64 * 1024 && !b->dev_entropy) { // small batch: the whole slab in one copy
		HIP_TRY(hipMemcpyAsync(b->d_slab, b->h_slab, b->slab_bytes, hipMemcpyHostToDevice, st));
		return LW_OK;
	}
	HIP_TRY(hipMemcpyAsync(b->d_recs, b->h_recs, b->n * sizeof(LwPacketRec), hipMemcpyHostToDevice, st));
	if (!b->dev_entropy)
		HIP_TRY(hipMemcpyAsync(b->d_floor, b->h_floor, b->n * ch * b->dec->T.fstride * sizeof(uint16_t),
					hipMemcpyHostToDevice, st));
	if (b->res_floats && !b->dev_entropy)
		HIP_TRY(hipMemcpyAsync(b->d_res, b->h_res, b->res_floats * sizeof(float), hipMemcpyHostToDevice, st));
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
	if (b->n_prep)
		HIP_TRY(hipMemcpyAsync(b->d_gen + 3 * b->max_packets, b->h_gen + 3 * b->max_packets, b->n_prep * sizeof(uint32_t),
					hipMemcpyHostToDevice, st));
	if (b->n_gen_ola)
		HIP_TRY(hipMemcpyAsync(b->d_ola, b->h_ola, b->n_gen_ola * sizeof(LwOlaDesc), hipMemcpyHostToDevice, st));
	if (b->n_gen_small)
		HIP_TRY(hipMemcpyAsync(b->d_tasks, b->h_tasks, (size_t)b->n_gen_small * ch * sizeof(LwGenTask), hipMemcpyHostToDevice, st));
	for (int cls = 0; cls < 2; cls++)
		if (b->n_tasks[cls])
			HIP_TRY(hipMemcpyAsync(b->d_slots[cls], b->h_slots[cls],
						b->n_tasks[cls] * (std::max(1u, 64 / b->dec->blkp[cls].lanes) * b->blk_passes[cls]) * sizeof(LwShortSlot),
						hipMemcpyHostToDevice, st));
	if (b->n_items)
		HIP_TRY(hipMemcpyAsync(b->d_items, b->h_items, b->n_items * sizeof(LwFastItem), hipMemcpyHostToDevice, st));
	if (b->n_halo_items)
		HIP_TRY(hipMemcpyAsync(b->d_halo_items, b->h_halo_items, b->n_halo_items * sizeof(LwFastItem), hipMemcpyHostToDevice, st));
	return LW_OK;
}

// entropy stage on the device: one wave per packet decodes floors and residues (once per upload)
static int device_entropy(lw_batch *b, hipStream_t st)
{
	if (!b->dev_entropy || b->ent_done || b->n == 0)
		return LW_OK;
	lw_decoder *d = b->dec;
	HIP_TRY(lw_launch_entropy(d->E, b->d_pk, b->d_recs, b->d_pool, b->d_floor, b->d_res, (uint32_t)b->n, st));
	HIP_TRY(hipGetLastError());
	b->ent_done = true;
	return LW_OK;
}

static int batch_launch(lw_batch *b, void *d_out, hipStream_t st, bool all_generic, float *tap)
{
	lw_decoder *d = b->dec;
	b->last_stream = (void *)st;
	if (b->n == 0)
		return LW_OK;
	const bool run_generic = b->has_generic || all_generic;
	const bool run_fast = b->has_fast && !all_generic;
	const bool run_short = (b->n_tasks[0] > 0 || b->n_tasks[1] > 0) && !all_generic;
	if (b->edge_mode && (run_fast || run_short) && !b->d_edge) {
		const size_t entries = b->max_packets * 2 * d->T.ch; // raw edges, and one flag each for k_mix (zero between launches)
		HIP_TRY(hipMalloc((void **)&b->d_edge, entries * (lw_edge_values(d) * sizeof(float) + sizeof(uint32_t))));
		HIP_TRY(hipMemsetAsync(b->d_edge + entries * lw_edge_values(d), 0, entries * sizeof(uint32_t), st));
	}
	if (run_short && (b->has_generic || all_generic) && !b->d_td) { // (k_short may read / write td blocks next to generic packets)
		const size_t maxres = b->max_packets * d->T.ch * d->T.state_chan_stride;
		HIP_TRY(hipMalloc((void **)&b->d_td, 2 * maxres * sizeof(float)));
	}
	const bool run_prep = b->n_prep > 0 && !all_generic;
	if (run_generic || run_prep) {
		const size_t maxres = b->max_packets * d->T.ch * d->T.state_chan_stride;
		if (((run_generic && d->any_coupling) || run_prep) && !b->d_decoupled)
			HIP_TRY(hipMalloc((void **)&b->d_decoupled, maxres * sizeof(float)));
		if (run_prep && d->prep_floors && !b->d_floor_alt)
			HIP_TRY(hipMalloc((void **)&b->d_floor_alt, b->max_packets * d->T.ch * d->T.fstride * sizeof(uint16_t)));
		if (run_generic && !b->d_td)
			HIP_TRY(hipMalloc((void **)&b->d_td, 2 * maxres * sizeof(float)));
	}
	if (run_fast && b->n_halo_items > b->halo_cap) {
		if (b->d_halo) {
			HIP_TRY(hipStreamSynchronize(st));
			(void)hipFree(b->d_halo);
			b->d_halo = nullptr;
		}
		const size_t cap = std::max<size_t>(b->n_halo_items, 64);
		HIP_TRY(hipMalloc((void **)&b->d_halo, cap * d->T.ch * std::max<size_t>(512, d->T.state_chan_stride / 2) * sizeof(float))); // [slots][ch][n1 / 4]
		b->halo_cap = cap;
	}
	const uint32_t break_spin = b->mix_break_spin ? b->mix_break_spin : g_break_mix_spin.load(); // (test hooks)
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
	B.ola = all_generic ? nullptr : b->d_ola;
	B.gen_tasks = all_generic || tap || d->any_floor0 ? nullptr : b->d_tasks;
	b->last_kernels.clear();
	if (b->dev_entropy) {
		if (int rc = device_entropy(b, st))
			return rc;
		b->last_kernels = "k_entropy,";
	}
	// the block classes whose stream shape the specialised kernels do not take as it is: k_prep writes their packets' residues after
	// every coupling step (x floor curve where the floor cannot be staged) and, if so, their floor records with the unit floor; those
	// classes' kernels then read the second pair of buffers (Bc[class])
	LwBatchDev Bc[2] = {B, B};
	if (run_prep) {
		HIP_TRY(lw_launch_prep(d->T, B, b->d_gen + 3 * b->max_packets, b->n_prep, d->d_prep_action, d->prep_floors ? b->d_floor_alt : nullptr, st));
		b->last_kernels += "k_prep,";
		for (int cls = 0; cls < 2; cls++)
			if (d->prep_cls[cls]) {
				Bc[cls].residue = b->d_decoupled;
				if (d->prep_floors)
					Bc[cls].floors = b->d_floor_alt;
			}
	}
	const int fast_cls = b->use_l10 ? b->l10_cls : 1; // the block class of the wave-pipeline kernel's packets
	if (run_generic) {
		HIP_TRY(lw_launch_generic_imdct(d->T, B, tap, st, b->max_n, d->any_coupling, all_generic));
		if (all_generic || b->n_gen_small + b->n_gen_large) // (else only k_ola_generic has work: LW_RF_TDONLY packets)
			b->last_kernels += d->any_coupling && (!d->T.pair_coupling || tap) ? "k_decouple,k_imdct_generic," : "k_imdct_generic,";
	}
	LwFastLaunch L{};
	if (run_fast) {
		L.off = d->fast.off;
		L.d_image = d->d_fast_image;
		L.d_items = b->d_items;
		L.n_items = (uint32_t)b->n_items;
		L.d_halo_items = b->d_halo_items;
		L.n_halo_items = (uint32_t)b->n_halo_items;
		const std::vector<LwFastUnit> &units = b->use_l10 ? d->blkp[b->l10_cls].units : b->use_l12 ? d->blkp[1].units_split
			: b->fast_split ? d->fast.units_split : d->fast.units;
		if (b->use_l10 || b->use_l12)
			L.d_image = d->d_blk_image[b->use_l12 ? 1 : b->l10_cls];
		L.d_sid12 = d->d_l12_sid;
		L.n_units = (uint32_t)units.size();
		L.per_round = b->fast_per_round;
		L.rounds = b->fast_rounds;
		L.dense = b->fast_dense;
		L.late_from = b->fast_late_from;
		L.has_tdonly = b->has_tdonly ? 1u : 0u;
		L.split = b->fast_split ? 1u : 0u;
		L.edge_mode = b->edge_mode ? 1u : 0u;
		L.d_edge = b->d_edge;
		L.edge_n = (uint32_t)lw_edge_values(d);
		for (size_t i = 0; i < units.size() && i < LW_FAST_WAVES; i++)
			L.units[i] = units[i];
		L.d_halo = b->d_halo;
		if (!b->use_l10 && !b->use_l12 && !d->fast.pre.empty()) { // k_long<..., PRE>: the units' own coupling programs
			L.pre_on = 1;
			for (size_t i = 0; i < d->fast.pre.size() && i < LW_FAST_WAVES; i++)
				L.pre[i] = d->fast.pre[i];
		}
	}
	auto short_launch = [&](int cls) {
		const LwShortPlan &bp = d->blkp[cls];
		LwShortLaunch S{};
		S.d_image = d->d_blk_image[cls];
		S.d_slots = b->d_slots[cls];
		S.lanes = bp.lanes;
		S.passes = b->blk_passes[cls];
		S.n_tasks = (uint32_t)b->n_tasks[cls];
		S.n_units = (uint32_t)bp.units.size();
		for (size_t i = 0; i < bp.units.size() && i < LW_FAST_WAVES; i++)
			S.units[i] = bp.units[i];
		S.d_edge = b->d_edge;
		for (uint32_t i = 0; i < LW_FAST_MAX_FLOORS; i++)
			S.fl_of[i] = bp.fl_of[i];
		return S;
	};
	// a mixed short / long batch small enough for the chip to hold at once: both kernels' work in ONE launch (k_mix)
	bool mixed = false;
	const bool same_bufs = !run_prep || d->prep_cls[0] == d->prep_cls[1]; // (one launch for both classes: one pair of buffers)
	if (run_fast && !b->use_l10 && !b->use_l12 && run_short && b->mix_mode != 0 && !b->n_tasks[1] && b->n_tasks[0] && b->d_edge && same_bufs) {
		const LwShortLaunch S = short_launch(0);
		if (lw_mix_applicable(L, S, d->n_cus)) {
			uint32_t *flags = (uint32_t *)(b->d_edge + b->max_packets * 2 * d->T.ch * lw_edge_values(d));
			HIP_TRY(lw_launch_mix(d->T, Bc[1], L, S, flags, b->d_err, break_spin, break_spin != 0, d_out, b->fmt, st));
			b->last_kernels += b->n_halo_items ? "k_long<halo>,k_mix," : "k_mix,";
			mixed = true;
		}
	}
	if (run_fast && b->use_l10 && run_short && b->mix_mode != 0 && !b->n_tasks[1] && b->n_tasks[0] && b->d_edge && same_bufs) {
		const LwShortLaunch S = short_launch(0);
		if (lw_mix10_applicable(L, S, d->n_cus)) { // a mixed batch the chip holds at once: long and short blocks in ONE launch (k_mix10)
			uint32_t *flags = (uint32_t *)(b->d_edge + b->max_packets * 2 * d->T.ch * lw_edge_values(d));
			HIP_TRY(lw_launch_mix10(d->T, Bc[1], L, S, flags, b->d_err, break_spin, break_spin != 0, d_out, b->fmt, st));
			b->last_kernels += b->n_halo_items ? "k_long10<halo>,k_mix10," : "k_mix10,";
			mixed = true;
		}
	}
	if (run_fast && b->use_l10 && mixed) {
		// (done above)
	} else if (run_fast && b->use_l10) {
		HIP_TRY(lw_launch_long10(d->T, Bc[fast_cls], L, d_out, b->fmt, st));
		b->last_kernels += b->n_halo_items ? "k_long10<halo>,k_long10," : "k_long10,";
	} else if (run_fast && b->use_l12) {
		HIP_TRY(lw_launch_long12(d->T, Bc[1], L, d_out, b->fmt, st));
		b->last_kernels += b->n_halo_items ? "k_long12<halo>,k_long12," : "k_long12,";
	} else if (run_fast && !mixed) {
		HIP_TRY(lw_launch_long(d->T, Bc[1], L, d_out, b->fmt, st));
		b->last_kernels += b->n_halo_items ? "k_long<halo>,k_long," : "k_long,";
	}
	if (run_short && !mixed)
		for (int cls = 1; cls >= 0; cls--) { // (the two classes never touch: generic packets lie between them)
			if (!b->n_tasks[cls])
				continue;
			const LwShortLaunch S = short_launch(cls);
			if (S.lanes > 64) {
				HIP_TRY(lw_launch_big(d->T, Bc[cls], S, d_out, b->fmt, st));
				b->last_kernels += "k_big,";
			} else {
				HIP_TRY(lw_launch_short(d->T, Bc[cls], S, d_out, b->fmt, st));
				b->last_kernels += "k_short,";
			}
		}
	if (run_generic) {
		lw_launch_generic_ola(d->T, B, d_out, b->fmt, st, all_generic);
		b->last_kernels += "k_ola_generic,";
	}
	if (!b->last_kernels.empty())
		b->last_kernels.pop_back();
	HIP_TRY(hipGetLastError());
	return LW_OK;
}

int lw_batch_device_entropy(lw_batch *b, void *hip_stream)
{
	if (!b)
		return LW_ERR_NULL_ARG;
	if (int rc = lw_decoder_set_device(b->dec))
		return rc;
	return device_entropy(b, (hipStream_t)hip_stream);
}

int lw_batch_synth(lw_batch *b, void *d_out, size_t out_capacity_elems, void *hip_stream)
{
	if (!b || (!d_out && b->out_elems))
		return LW_ERR_NULL_ARG;
	if (out_capacity_elems < b->out_elems)
		return LW_ERR_CAPACITY;
	if (int rc = lw_decoder_set_device(b->dec))
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
	HIP_TRY(hipMalloc(&b->d_out, cap * lw_elem_size(b->fmt)));
	b->d_out_elems = cap;
	return LW_OK;
}

int lw_batch_synth_to_host(lw_batch *b, void *h_out, size_t out_capacity_elems, void *hip_stream)
{
	if (!b || (!h_out && b->out_elems))
		return LW_ERR_NULL_ARG;
	if (out_capacity_elems < b->out_elems)
		return LW_ERR_CAPACITY;
	if (int rc = lw_decoder_set_device(b->dec))
		return rc;
	if (int rc = ensure_internal_out(b))
		return rc;
	hipStream_t st = (hipStream_t)hip_stream;
	if (int rc = batch_launch(b, b->d_out, st, b->force_generic, nullptr))
		return rc;
	if (b->out_elems)
		HIP_TRY(hipMemcpyAsync(h_out, b->d_out, b->out_elems * lw_elem_size(b->fmt), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	return lw_batch_device_status(b);
}

int lw_batch_tap(lw_batch *b, size_t idx, int tap, float *dst, size_t cap_floats)
{
	if (!b || !dst)
		return LW_ERR_NULL_ARG;
	if (idx >= b->n || b->status[idx] != LW_OK)
		return LW_ERR_CAPACITY;
	lw_decoder *d = b->dec;
	if (int rc = lw_decoder_set_device(d))
		return rc;
	const LwPacketRec &r = b->h_recs[idx];
	const size_t n = (size_t)1 << r.bs, ch = d->T.ch;
	const size_t want = tap == LW_TAP_POST_MDCT ? ch * n : ch * n / 2;
	if (cap_floats < want)
		return LW_ERR_CAPACITY;
	if (tap == LW_TAP_RESIDUE_PRE_INVERSE && !b->dev_entropy) {
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
		src = b->d_res + r.res_off; // the vectors k_entropy built on the device
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
	if (int rc = lw_decoder_set_device(d))
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
	if (!rc && !lw_hip_ok(hipDeviceSynchronize(), "sync"))
		rc = LW_ERR_DEVICE;
	if (!rc && !lw_hip_ok(hipMemcpy(out, b->d_td, sizeof(float) * n, hipMemcpyDeviceToHost), "memcpy td"))
		rc = LW_ERR_DEVICE;
	lw_batch_destroy(b);
	return rc;
}

} // extern "C"
