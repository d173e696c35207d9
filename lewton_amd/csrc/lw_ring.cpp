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
#include "../../include/lewton_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

enum SlotState { SLOT_FREE, SLOT_STAGING, SLOT_STAGED, SLOT_LAUNCHED, SLOT_COLLECTED };

struct Slot {
	lw_batch *batch = nullptr;
	void *d_out = nullptr, *h_out = nullptr;
	hipStream_t stream = nullptr;
	hipEvent_t kernels_done = nullptr, all_done = nullptr;
	SlotState state = SLOT_FREE;
	size_t n = 0, out_elems = 0;
};

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
	std::mutex mu;
	std::condition_variable cv;
};

namespace {

bool ok(hipError_t e)
{
	if (e == hipSuccess)
		return true;
	(void)hipGetLastError();
	return false;
}

} // namespace

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
	for (Slot &s : r->slots) {
		if (!good)
			break;
		int e = 0;
		s.batch = lw_batch_create(d, max_packets, fmt, &e);
		good = s.batch && ok(hipMalloc(&s.d_out, r->cap_elems * r->esz)) && ok(hipHostMalloc(&s.h_out, r->cap_elems * r->esz)) &&
			ok(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking)) &&
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
	int rc = lw_batch_upload(s->batch, s->stream);
	if (rc == LW_OK) // entropy stage on the device: no stream state involved, so it runs beside the previous launches' kernels
		rc = lw_batch_device_entropy(s->batch, s->stream);
	if (rc == LW_OK && r->last_kernels && !ok(hipStreamWaitEvent(s->stream, r->last_kernels, 0)))
		rc = LW_ERR_DEVICE;
	if (rc == LW_OK)
		rc = lw_batch_synth(s->batch, s->d_out, r->cap_elems, s->stream);
	if (rc == LW_OK && !ok(hipEventRecord(s->kernels_done, s->stream)))
		rc = LW_ERR_DEVICE;
	// The PCM copies of consecutive launches run ONE AT A TIME, in launch order: left to themselves the copies of all slots in
	// flight share the link, finish together, the caller (first-in first-out) refills all slots at once, and the batches then
	// move through upload / entropy / synthesis / copy in lock step -- the copy engine idle while the kernels run and the
	// other way round (measured: every third collect waiting 1.2 ms, 7.2 M packets/s; staggered 13 M, profiles/r04_e2e_ring.txt)
	// (per ring: the same order over ALL rings of a device was measured as well -- two logical shards on one GPU then made 6-7
	// instead of 8.3 M packets/s, profiles/r04_e2e_ring.txt)
	if (rc == LW_OK && s->out_elems && r->last_all_done && !ok(hipStreamWaitEvent(s->stream, r->last_all_done, 0)))
		rc = LW_ERR_DEVICE;
	if (rc == LW_OK && s->out_elems &&
			!ok(hipMemcpyAsync(s->h_out, s->d_out, s->out_elems * r->esz, hipMemcpyDeviceToHost, s->stream)))
		rc = LW_ERR_DEVICE;
	if (rc == LW_OK && !ok(hipEventRecord(s->all_done, s->stream)))
		rc = LW_ERR_DEVICE;
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
		if (!ok(hipSetDevice(r->device)) || !ok(hipEventSynchronize(s->all_done)))
			return LW_ERR_DEVICE;
		// a kernel of this batch raised its device error word: the batch's results carry LW_ERR_DEVICE, its PCM is void; the
		// slot is COLLECTED all the same (release it as usual)
		dev_rc = lw_batch_device_status(s->batch);
		std::lock_guard<std::mutex> g(r->mu);
		s->state = SLOT_COLLECTED;
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
