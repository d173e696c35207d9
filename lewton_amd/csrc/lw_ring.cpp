// Staging ring of the MI355X audio-packet decode path (product code): the north-star's "pinned hipMemcpyAsync staging
// ring so that entropy decode of packet N+1 overlaps GPU synthesis of packet N", behind the C ABI (include/lewton_amd.h,
// lw_ring_*).  Built on the public batch API only (lw_batch_*): a slot is one lw_batch (pinned records + device mirror)
// plus a device PCM buffer, a pinned host PCM buffer, a HIP stream and two events.
//
//   stage   (host)    lw_batch_entropy into the slot's pinned staging: bit-serial Huffman / VQ decode on the host threads
//   launch  (queue)   H2D of the records -> synthesis kernels -> D2H of the PCM, all asynchronous on the slot's stream; the
//                     kernels wait for the kernels of the previous launch (consecutive batches may carry the same
//                     streams' window state), the copies of different slots overlap each other and the kernels
//   collect (wait)    blocks until the OLDEST launched slot has its PCM in pinned host memory
//   release           gives the slot back
//
// Slots are used strictly first-in first-out, so every PreviousWindowRight sees its packets in submission order
// (audio.rs:919 touches only its own pwr).  stage() and launch()/collect()/release() may be called from two different
// threads (the Ogg reader's producer thread stages while the caller's thread launches and collects).
//
// Tenants (several decoders on one GPU: lw_decoder_set_shared_device, lw_decoder_set_cu_share): a tenant's ring runs its
// launches' kernels one launch after the other, and its PCM copies are issued by the device's COPIER thread once their kernels
// have finished, on one copy stream for all tenants of the device (see Copier below for the measurement behind it).
#include "../../include/lewton_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

// lw_runtime.cpp: the kinds of tenant stream this process has made on a device (see "Known hazard" in include/lewton_amd.h)
#define LW_TENANT_COPIER_STREAM 1
#define LW_TENANT_MASKED_STREAMS 2
int lw_tenant_streams(int device);
void lw_tenant_streams_note(int device, int kind);

#define LW_RING_MAX_DEVICES 64

namespace {

enum SlotState { SLOT_FREE, SLOT_STAGING, SLOT_STAGED, SLOT_LAUNCHED, SLOT_COLLECTED };

struct Slot {
	lw_batch *batch = nullptr;
	void *d_out = nullptr, *h_out = nullptr;
	hipStream_t stream = nullptr;
	hipEvent_t kernels_done = nullptr, all_done = nullptr;
	SlotState state = SLOT_FREE;
	size_t n = 0, out_elems = 0;
	bool copy_queued = false, copy_failed = false; // (copier rings) the PCM copy is still to be issued / could not be issued
	int dev_rc = LW_OK; // lw_batch_device_status of the launch, latched by the first collect: every collect until the release reports it
};

struct Copier;

} // namespace

struct lw_ring {
	lw_decoder *dec = nullptr;
	int device = 0, fmt = 0;
	size_t max_packets = 0, cap_elems = 0, esz = 2;
	std::vector<Slot> slots;
	// FIFO cursors (slot indices advance modulo slots.size()): next to stage, next to launch, next to collect / release
	size_t i_stage = 0, i_launch = 0, i_collect = 0;
	hipEvent_t last_kernels = nullptr; // kernels_done of the most recent launch (null before the first)
	hipEvent_t last_all_done = nullptr; // all_done of the most recent launch: the PCM copies run one at a time, in order
	bool kernels_fifo = false; // a tenant's ring: the kernels of launch k+1 (k_entropy included) wait for those of launch k
	bool masked = false;       // its slots' streams carry a CU mask (lw_decoder_set_cu_share)
	bool copy_on_slot = false; // the copier issues this ring's copies on the slots' own streams (CU-masked streams in the process)
	Copier *copier = nullptr;  // a tenant's ring: the device's copier issues the PCM copies (null: on the slot's own stream)
	std::mutex mu;
	std::condition_variable cv;
};

namespace {

std::atomic<int> g_policy{-1}; // lw_debug_ring_policy

bool ok(hipError_t e)
{
	if (e == hipSuccess)
		return true;
	(void)hipGetLastError();
	return false;
}

// The PCM copies of the tenants of one device.  A device-to-host copy that is queued BEHIND ITS KERNELS on the slot's stream
// sits in the SDMA engine's queue until those kernels are done, and every copy issued after it -- another ring's, long ready
// -- waits behind it (tools/micro/d2h_streams.hip: a ready 0.31 ms copy on another stream completes 0.3 ms after the END of
// the first stream's kernel, however long that runs).  One ring never notices: its copies are in launch order anyway.  Two
// rings on one device stall each other every few launches (1.3 ms copies in the trace, every third collect of the sharder
// waiting 1.0-1.4 ms: 8.6-9.0 M packets/s where the link allows 12.5 M; profiles/r05_tenants.txt), and ordering the copies
// by events across the rings only makes the queue longer (6.4-7.1 M).  So for tenants the copy is issued by a host thread
// when its kernels HAVE finished: one copier per device, jobs in launch order, one copy stream -- the engine's queue holds
// ready copies only (two logical shards: 11.1-12.4 M packets/s, the rate of ONE ring with the same packets per call).
struct Copier {
	int device = 0;
	std::mutex mu;
	std::condition_variable cv;
	std::deque<std::pair<lw_ring *, Slot *>> jobs;
	// the thread of generation g serves until stop_gen >= g: the last ring to let go raises stop_gen and joins that thread OUTSIDE the
	// table's lock, while a new first ring may already have started generation g + 1 (both under mu)
	uint64_t start_gen = 0, stop_gen = 0;
	int users = 0; // rings holding it (under g_copiers_mu)
	hipStream_t own_stream = nullptr; // the copies of rings on ordinary streams; a ring on CU-masked streams copies on its slots' own
	std::thread th;

	void main(uint64_t gen)
	{
		(void)hipSetDevice(device);
		for (;;) {
			std::pair<lw_ring *, Slot *> j;
			{
				std::unique_lock<std::mutex> g(mu);
				cv.wait(g, [&]() { return stop_gen >= gen || !jobs.empty(); });
				if (jobs.empty())
					return;
				j = jobs.front();
				jobs.pop_front();
			}
			lw_ring *r = j.first;
			Slot *s = j.second;
			hipStream_t on = r->copy_on_slot || !own_stream ? s->stream : own_stream;
			const bool good = ok(hipEventSynchronize(s->kernels_done)) &&
				ok(hipMemcpyAsync(s->h_out, s->d_out, s->out_elems * r->esz, hipMemcpyDeviceToHost, on)) &&
				ok(hipEventRecord(s->all_done, on));
			std::lock_guard<std::mutex> g(r->mu); // (the ring waits for copy_queued of all its slots before it goes away)
			s->copy_failed = !good;
			s->copy_queued = false;
			r->cv.notify_all();
		}
	}
};
std::mutex g_copiers_mu;
Copier *g_copiers[LW_RING_MAX_DEVICES];

Copier *copier_acquire(int device, bool own_stream)
{
	if (device < 0 || device >= LW_RING_MAX_DEVICES)
		return nullptr; // (the caller falls back to copies queued on the slots' own streams)
	std::lock_guard<std::mutex> g(g_copiers_mu);
	Copier *&c = g_copiers[device];
	if (!c) {
		// Kept for the life of the process, and so is its stream: all tenants' copies on ONE stream of the copier's made 11.1-12.4 M
		// packets/s where the same copies issued on the slots' own streams made 8.0-10.6 M (two logical shards; same box, same
		// minute).  Next to CU-MASKED streams that stream is not used: destroying it, leaving it to the runtime's teardown, or even
		// _exit() then hung one run in three (the queues of the process never drained) -- a ring on masked streams copies on them.
		c = new Copier();
		c->device = device;
	}
	if (own_stream && !c->own_stream) {
		if (!ok(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)))
			return nullptr;
		lw_tenant_streams_note(device, LW_TENANT_COPIER_STREAM);
	}
	if (c->users++ == 0) {
		uint64_t gen;
		{
			std::lock_guard<std::mutex> q(c->mu);
			gen = ++c->start_gen;
		}
		c->th = std::thread([p = c, gen]() { p->main(gen); });
	}
	return c;
}

void copier_release(Copier *c)
{
	if (!c)
		return;
	// the thread is joined OUTSIDE the table's lock: it may still be inside a long hipEventSynchronize of the ring that is going
	// away, and creating or destroying a ring on any other device must not wait for that
	std::thread done;
	{
		std::lock_guard<std::mutex> g(g_copiers_mu);
		if (--c->users > 0)
			return;
		{
			std::lock_guard<std::mutex> q(c->mu);
			c->stop_gen = c->start_gen;
		}
		c->cv.notify_all();
		done = std::move(c->th);
	}
	if (done.joinable())
		done.join();
}

// every copy the copier still had to issue for this ring has been issued (or has failed) AND has completed: the ring's buffers
// are no longer read or written by the copier's stream
bool wait_copies_done(lw_ring *r)
{
	{
		std::unique_lock<std::mutex> g(r->mu);
		r->cv.wait(g, [&]() {
			for (const Slot &s : r->slots)
				if (s.copy_queued)
					return false;
			return true;
		});
	}
	return r->copy_on_slot || !r->copier->own_stream || ok(hipStreamSynchronize(r->copier->own_stream)); // (else: the slots' own streams)
}

} // namespace

hipError_t lw_decoder_stream_create(lw_decoder *d, hipStream_t *s); // lw_runtime.cpp: on the decoder's CU share
extern "C" int lw_decoder_cu_count(const lw_decoder *d);
extern "C" int lw_decoder_device_cu_count(const lw_decoder *d);
extern "C" int lw_decoder_shares_device(const lw_decoder *d);

extern "C" {

int lw_decoder_device(const lw_decoder *d);                 // lw_runtime.cpp
size_t lw_decoder_max_block_elems(const lw_decoder *d);     // channels * blocksize_1 / 2

lw_ring *lw_ring_create(lw_decoder *d, size_t n_slots, size_t max_packets, int fmt, int *err)
{
	int dummy;
	if (!err)
		err = &dummy;
	*err = LW_OK;
	if (!d || n_slots < 1 || n_slots > 64 || max_packets == 0 || fmt < 0 || fmt > 2) {
		*err = LW_ERR_NULL_ARG;
		return nullptr;
	}
	auto *r = new lw_ring();
	r->dec = d;
	r->device = lw_decoder_device(d);
	r->fmt = fmt;
	r->max_packets = max_packets;
	r->esz = fmt == LW_FMT_F32_PLANAR ? 4 : 2;
	r->cap_elems = max_packets * lw_decoder_max_block_elems(d);
	r->slots.resize(n_slots);
	bool good = ok(hipSetDevice(r->device));
	{
		r->masked = lw_decoder_cu_count(d) < lw_decoder_device_cu_count(d);
		const bool tenant = r->masked || lw_decoder_shares_device(d);
		const int pol = g_policy.load();
		r->kernels_fifo = pol < 0 ? tenant : (pol & 1) != 0;
		// The two kinds of tenant stream must not meet in one process (include/lewton_amd.h: a process that had copied on the
		// copier's own stream AND run CU-masked streams was seen not to exit): CU-masked streams are refused once the copier's
		// stream exists on this device; the other way round the copier simply issues its copies on the slots' own streams.
		if (good && r->masked && (lw_tenant_streams(r->device) & LW_TENANT_COPIER_STREAM)) {
			*err = LW_ERR_UNSUPPORTED;
			lw_ring_destroy(r);
			return nullptr;
		}
		const bool own = !r->masked && !(lw_tenant_streams(r->device) & LW_TENANT_MASKED_STREAMS);
		if (good && (pol < 0 ? tenant : (pol & 2) != 0))
			r->copier = copier_acquire(r->device, own); // (null -- device ordinal beyond the table, no stream to be had: copies behind the kernels, as for a lone ring)
		r->copy_on_slot = r->masked || !own;
	}
	for (Slot &s : r->slots) {
		if (!good)
			break;
		int e = 0;
		s.batch = lw_batch_create(d, max_packets, fmt, &e);
		good = s.batch && ok(hipMalloc(&s.d_out, r->cap_elems * r->esz)) && ok(hipHostMalloc(&s.h_out, r->cap_elems * r->esz)) &&
			ok(lw_decoder_stream_create(d, &s.stream)) &&
			ok(hipEventCreateWithFlags(&s.kernels_done, hipEventDisableTiming)) &&
			ok(hipEventCreateWithFlags(&s.all_done, hipEventDisableTiming));
	}
	if (!good) {
		*err = LW_ERR_DEVICE;
		lw_ring_destroy(r);
		return nullptr;
	}
	return r;
}

void lw_ring_destroy(lw_ring *r)
{
	if (!r)
		return;
	(void)hipSetDevice(r->device);
	if (r->copier)
		(void)wait_copies_done(r);
	for (Slot &s : r->slots) {
		if (s.stream)
			(void)hipStreamSynchronize(s.stream);
		if (s.batch)
			lw_batch_destroy(s.batch);
		if (s.d_out)
			(void)hipFree(s.d_out);
		if (s.h_out)
			(void)hipHostFree(s.h_out);
		if (s.kernels_done)
			(void)hipEventDestroy(s.kernels_done);
		if (s.all_done)
			(void)hipEventDestroy(s.all_done);
		if (s.stream)
			(void)hipStreamDestroy(s.stream);
	}
	copier_release(r->copier);
	delete r;
}

int lw_ring_set_entropy_on_device(lw_ring *r, int on)
{
	if (!r)
		return LW_ERR_NULL_ARG;
	std::lock_guard<std::mutex> g(r->mu);
	for (Slot &s : r->slots)
		if (s.state != SLOT_FREE)
			return LW_ERR_CAPACITY; // only between batches
	for (Slot &s : r->slots)
		if (int rc = lw_batch_set_entropy_on_device(s.batch, on))
			return rc;
	return LW_OK;
}

size_t lw_ring_slots(const lw_ring *r)
{
	return r ? r->slots.size() : 0;
}

size_t lw_ring_in_flight(lw_ring *r)
{
	if (!r)
		return 0;
	std::lock_guard<std::mutex> g(r->mu);
	size_t n = 0;
	for (const Slot &s : r->slots)
		n += s.state != SLOT_FREE;
	return n;
}

int lw_ring_stage(lw_ring *r, const lw_packet *pkts, size_t n, int n_threads)
{
	if (!r || (!pkts && n))
		return LW_ERR_NULL_ARG;
	if (n > r->max_packets)
		return LW_ERR_CAPACITY;
	Slot *s;
	{
		std::lock_guard<std::mutex> g(r->mu);
		s = &r->slots[r->i_stage];
		if (s->state != SLOT_FREE)
			return LW_ERR_CAPACITY; // every slot is in flight: collect + release first
		s->state = SLOT_STAGING;
	}
	const int rc = lw_batch_entropy(s->batch, pkts, n, n_threads); // the slot is this thread's alone while it is STAGING
	std::lock_guard<std::mutex> g(r->mu);
	if (rc != LW_OK) {
		s->state = SLOT_FREE;
		return rc;
	}
	s->n = n;
	s->out_elems = lw_batch_out_elems(s->batch);
	s->state = SLOT_STAGED;
	r->i_stage = (r->i_stage + 1) % r->slots.size();
	r->cv.notify_all();
	return LW_OK;
}

int lw_ring_launch(lw_ring *r)
{
	if (!r)
		return LW_ERR_NULL_ARG;
	Slot *s;
	{
		std::lock_guard<std::mutex> g(r->mu);
		s = &r->slots[r->i_launch];
		if (s->state != SLOT_STAGED)
			return LW_ERR_CAPACITY; // nothing staged
	}
	if (!ok(hipSetDevice(r->device)))
		return LW_ERR_DEVICE;
	const bool kernels_fifo = r->kernels_fifo;
	int rc = lw_batch_upload(s->batch, s->stream);
	// A tenant's kernels run one launch after the other: on its share of the CUs the next launch's k_entropy would take the
	// CUs this launch's whole-CU workgroups are waiting for (k_long 28 -> 300-700 us, measured) and gain nothing
	if (rc == LW_OK && kernels_fifo && r->last_kernels && !ok(hipStreamWaitEvent(s->stream, r->last_kernels, 0)))
		rc = LW_ERR_DEVICE;
	if (rc == LW_OK) // entropy stage on the device: no stream state involved, so it runs beside the previous launches' kernels
		rc = lw_batch_device_entropy(s->batch, s->stream);
	if (rc == LW_OK && !kernels_fifo && r->last_kernels && !ok(hipStreamWaitEvent(s->stream, r->last_kernels, 0)))
		rc = LW_ERR_DEVICE;
	if (rc == LW_OK)
		rc = lw_batch_synth(s->batch, s->d_out, r->cap_elems, s->stream);
	if (rc == LW_OK && !ok(hipEventRecord(s->kernels_done, s->stream)))
		rc = LW_ERR_DEVICE;
	if (r->copier && s->out_elems) {
		// a tenant's ring: the device's copier issues the copy once the kernels are done (see Copier)
		if (rc == LW_OK) {
			{
				std::lock_guard<std::mutex> g(r->mu);
				s->copy_queued = true;
				s->copy_failed = false;
			}
			{
				std::lock_guard<std::mutex> q(r->copier->mu);
				r->copier->jobs.emplace_back(r, s);
			}
			r->copier->cv.notify_one();
		}
	} else {
		// The PCM copies of consecutive launches run ONE AT A TIME, in launch order: left to themselves the copies of all slots
		// in flight share the link, finish together, the caller (first-in first-out) refills all slots at once, and the batches
		// then move through upload / entropy / synthesis / copy in lock step -- the copy engine idle while the kernels run and
		// the other way round (measured: every third collect waiting 1.2 ms, 7.2 M packets/s; staggered 13 M,
		// profiles/r04_e2e_ring.txt)
		if (rc == LW_OK && s->out_elems && r->last_all_done && !ok(hipStreamWaitEvent(s->stream, r->last_all_done, 0)))
			rc = LW_ERR_DEVICE;
		if (rc == LW_OK && s->out_elems &&
				!ok(hipMemcpyAsync(s->h_out, s->d_out, s->out_elems * r->esz, hipMemcpyDeviceToHost, s->stream)))
			rc = LW_ERR_DEVICE;
		if (rc == LW_OK && !ok(hipEventRecord(s->all_done, s->stream)))
			rc = LW_ERR_DEVICE;
	}
	std::lock_guard<std::mutex> g(r->mu);
	if (rc != LW_OK)
		return rc; // the slot stays STAGED (its host-side bookkeeping is done): the caller may retry or drop the ring
	r->last_kernels = s->kernels_done;
	r->last_all_done = s->all_done;
	s->state = SLOT_LAUNCHED;
	r->i_launch = (r->i_launch + 1) % r->slots.size();
	return LW_OK;
}

int lw_ring_submit(lw_ring *r, const lw_packet *pkts, size_t n, int n_threads)
{
	if (int rc = lw_ring_stage(r, pkts, n, n_threads))
		return rc;
	return lw_ring_launch(r);
}

int lw_ring_collect(lw_ring *r, const lw_packet_result **results, size_t *n, const void **pcm, size_t *pcm_elems)
{
	if (!r)
		return LW_ERR_NULL_ARG;
	Slot *s;
	bool wait;
	{
		std::lock_guard<std::mutex> g(r->mu);
		s = &r->slots[r->i_collect];
		if (s->state != SLOT_LAUNCHED && s->state != SLOT_COLLECTED)
			return LW_ERR_CAPACITY; // nothing launched
		wait = s->state == SLOT_LAUNCHED; // (collect is idempotent until release)
	}
	int dev_rc = LW_OK;
	if (wait) {
		if (r->copier) { // the copier issues this slot's copy once its kernels are done: all_done is recorded only then
			std::unique_lock<std::mutex> g(r->mu);
			r->cv.wait(g, [&]() { return !s->copy_queued; });
			if (s->copy_failed)
				return LW_ERR_DEVICE;
		}
		if (!ok(hipSetDevice(r->device)) || !ok(hipEventSynchronize(s->all_done)))
			return LW_ERR_DEVICE;
		// a kernel of this batch raised its device error word: the batch's results carry LW_ERR_DEVICE, its PCM is void; the
		// slot is COLLECTED all the same (release it as usual)
		dev_rc = lw_batch_device_status(s->batch);
		std::lock_guard<std::mutex> g(r->mu);
		s->dev_rc = dev_rc;
		s->state = SLOT_COLLECTED;
	} else {
		dev_rc = s->dev_rc; // (a second collect of the same slot: the same verdict, not LW_OK over a failed batch's samples)
	}
	if (results)
		*results = lw_batch_results(s->batch);
	if (n)
		*n = s->n;
	if (pcm)
		*pcm = s->h_out;
	if (pcm_elems)
		*pcm_elems = s->out_elems;
	return dev_rc;
}

int lw_ring_release(lw_ring *r)
{
	if (!r)
		return LW_ERR_NULL_ARG;
	std::lock_guard<std::mutex> g(r->mu);
	Slot &s = r->slots[r->i_collect];
	if (s.state != SLOT_COLLECTED)
		return LW_ERR_CAPACITY;
	s.state = SLOT_FREE;
	r->i_collect = (r->i_collect + 1) % r->slots.size();
	r->cv.notify_all();
	return LW_OK;
}

/* Drops everything that was staged or launched and not yet released: waits for the GPU work in flight, frees all slots.
 * The host-side bookkeeping of the PreviousWindowRight objects the dropped batches touched is NOT rolled back (the
 * caller snapshots / restores them, as the Ogg stream layer does with lw_pwr_snapshot). */
int lw_ring_drain(lw_ring *r)
{
	if (!r)
		return LW_ERR_NULL_ARG;
	if (!ok(hipSetDevice(r->device)))
		return LW_ERR_DEVICE;
	int rc = LW_OK;
	if (r->copier && !wait_copies_done(r))
		rc = LW_ERR_DEVICE;
	for (Slot &s : r->slots)
		if (!ok(hipStreamSynchronize(s.stream)))
			rc = LW_ERR_DEVICE;
	std::lock_guard<std::mutex> g(r->mu);
	for (Slot &s : r->slots)
		s.state = SLOT_FREE;
	r->i_stage = r->i_launch = r->i_collect = 0;
	r->cv.notify_all();
	return rc;
}

// measurement hook, process-wide, read by lw_ring_create: -1 = by the ring's decoder (a tenant: kernels in order per ring, copies
// by the device's copier), else bit 0 = kernels of a ring one launch after the other, bit 1 = copies by the device's copier
void lw_debug_ring_policy(int bits)
{
	g_policy.store(bits);
}

const char *lw_ring_last_kernels(const lw_ring *r)
{
	if (!r)
		return "";
	const size_t i = (r->i_launch + r->slots.size() - 1) % r->slots.size();
	return lw_batch_last_kernels(r->slots[i].batch);
}

/* elements the most recently staged batch will produce (known once lw_ring_stage has planned it; no GPU involved) */
size_t lw_ring_last_staged_elems(lw_ring *r)
{
	if (!r)
		return 0;
	std::lock_guard<std::mutex> g(r->mu);
	return r->slots[(r->i_stage + r->slots.size() - 1) % r->slots.size()].out_elems;
}

uint64_t lw_ring_slot_algorithmic_bytes(const lw_ring *r)
{
	if (!r)
		return 0;
	return lw_batch_algorithmic_bytes(r->slots[r->i_collect].batch);
}

} // extern "C"
