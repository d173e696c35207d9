// Sharding of independent streams over the GPUs of one node, in one process (product code; include/lewton_amd.h lw_sharder_*).
//
// A stream's decode touches only its own PreviousWindowRight and the immutable headers (audio.rs:919), so streams never
// exchange data: shard g owns the streams with stream_id mod G == g (BASELINE configs[4]).  A shard = one device
// context (lw_decoder: tables and the state pool of its streams in that GPU's HBM), one staging ring (pinned records and pinned
// PCM per slot, a HIP stream per slot) and one worker thread.  lw_sharder_submit splits a list of packets by owner and has every
// shard stage (host entropy decode) and launch (H2D, kernels, D2H, asynchronous) its part on its own thread and device -- all
// shards at once, no collective, nothing crossing xGMI; lw_sharder_collect waits for the oldest call.  A device may be listed several times (logical shards): that is how the N > 1 logic
// is tested on a one-GPU box.  The process-per-GPU form (bench.py under torch.distributed.run) uses the same rule through
// lewton_amd/shard.py; this is the single-process form INTEGRATION.md section 3 describes.
#include "../../include/lewton_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

struct lw_shard_stream {
	lw_sharder *owner = nullptr;
	size_t shard = 0;
	uint64_t id = 0;
	lw_pwr *pwr = nullptr;
};

namespace {

// what one lw_sharder_submit call put on one shard
struct Part {
	std::vector<size_t> idx;        // positions (in the caller's packet list) of this shard's packets, in list order
	std::vector<lw_packet> pk;
	size_t expect = 0;              // how many packets of the call are this shard's (counted by the caller; the worker gathers them)
	size_t out_elems = 0, base = 0; // elements this shard produces / where its first block starts in the caller's buffer
	int rc = LW_OK;
	bool launched = false;          // a ring slot holds this part (must be collected and released)
	uint64_t gen = 0;               // the shard's ring generation the slot belongs to (a drained ring starts a new one)
};

struct Call {
	std::vector<Part> parts;        // one per shard
	size_t n = 0, total_elems = 0;
	int n_threads = 0;
	const lw_shard_packet *pkts = nullptr; // the caller's list, valid while lw_sharder_submit runs (the workers gather their parts from it)
	// collect phase
	void *out = nullptr;
	lw_packet_result *results = nullptr;
	bool keep = false;              // zero-copy collect: the slots stay with the call until lw_sharder_release
	const void **pcm = nullptr;     // zero-copy collect: per shard, the pinned PCM of its slot
	size_t *elems = nullptr;
	bool collected = false;
	// completion of the jobs handed to the workers for this call (one caller waits: its own mutex and condition variable,
	// so that a finishing worker wakes that caller and nobody else)
	std::mutex mu;
	std::condition_variable cv;
	size_t pending = 0;
};

enum JobKind { JOB_STAGE, JOB_COLLECT };

struct Job {
	JobKind kind;
	Call *call;
};

struct Shard {
	size_t index = 0;
	int device = 0;
	lw_decoder *dec = nullptr;
	lw_ring *ring = nullptr;        // pinned records + pinned PCM per slot, H2D / kernels / D2H asynchronous on the slot's stream
	std::thread worker;
	// the worker's own queue, mutex and condition variable: a job for shard g wakes worker g only
	std::mutex mu;
	std::condition_variable cv;
	std::deque<Job> jobs;
	bool quit = false;
	// collect / release / drain of the ring exclude each other (the worker's stage + launch run beside them: the ring allows one
	// staging and one collecting thread).  A drain -- after a failed launch or a failed collect -- frees every slot and resets
	// the ring's cursors, so whatever older calls still hold on this ring is gone: `gen` tells their parts apart from new ones.
	std::mutex ring_mu;
	uint64_t gen = 0;               // under ring_mu
	// the streams opened on this shard (lw_sharder_stream_open / _close): a drain resets their window state
	std::mutex streams_mu;
	std::vector<lw_shard_stream *> streams;
};

} // namespace

// Every shard runs on a staging ring (lw_ring_*): lw_sharder_submit has each shard's worker thread run the host entropy stage
// of its packets into the next ring slot and queue H2D, kernels and D2H (into the slot's PINNED PCM buffer) without waiting;
// it returns when every shard has staged and launched, i.e. while all GPUs work.  The caller's next submit therefore
// overlaps every shard's host stage of call k+1 with the GPU work of call k on all devices -- no GPU idles through the
// host phase of a call, which is what the lock-step form of round 2 did.  lw_sharder_collect waits for the oldest call,
// copies each shard's PCM from pinned memory into the caller's buffer (the shards' workers, in parallel) and frees the
// slots.  Up to LW_SHARD_SLOTS calls may be in flight.
#define LW_SHARD_SLOTS 3

extern "C" size_t lw_ring_last_staged_elems(lw_ring *r); // lw_ring.cpp

struct lw_sharder {
	std::vector<std::unique_ptr<Shard>> shards;
	size_t max_packets = 0;
	int fmt = 0;
	size_t esz = 2;
	std::mutex call_mu;             // serialises the public calls
	std::deque<std::unique_ptr<Call>> calls; // submitted and not yet collected, oldest first (guarded by call_mu)
	std::vector<std::unique_ptr<Call>> spare; // consumed calls: their vectors keep their capacity for the next submit (call_mu)

	std::unique_ptr<Call> fresh_call()
	{
		std::unique_ptr<Call> c;
		if (!spare.empty()) {
			c = std::move(spare.back());
			spare.pop_back();
		} else {
			c = std::make_unique<Call>();
			c->parts.resize(shards.size());
		}
		for (Part &p : c->parts) {
			p.idx.clear();
			p.pk.clear();
			p.expect = p.out_elems = p.base = 0;
			p.rc = LW_OK;
			p.launched = false;
		}
		c->n = c->total_elems = 0;
		c->out = nullptr;
		c->results = nullptr;
		c->keep = c->collected = false;
		c->pcm = nullptr;
		c->elems = nullptr;
		c->pkts = nullptr;
		return c;
	}

	void retire_front()
	{
		spare.push_back(std::move(calls.front()));
		calls.pop_front();
	}

	// Starts a shard's ring over after a device failure.  Everything in flight on the shard is gone with it -- parts of OLDER and
	// of NEWER calls alike report LW_ERR_DEVICE when they are collected (`gen`) -- and the host halves of the shard's window states
	// have advanced past batches that never completed: every stream of the shard is reset (`PreviousWindowRight::new()`), so that
	// its next packet yields no samples (audio.rs:1140-1152) instead of samples overlapped with a stale right part.
	// (s.ring_mu held; the public calls are serialised by call_mu, so no stage of this shard runs beside this)
	void drain_locked(Shard &s)
	{
		(void)lw_ring_drain(s.ring);
		s.gen++;
		std::lock_guard<std::mutex> g(s.streams_mu);
		for (lw_shard_stream *st : s.streams)
			lw_pwr_reset(st->pwr);
	}

	// the calling thread's current HIP device, restored when a public call that ran ring operations on the caller's thread returns
	// (lw_ring_collect / lw_ring_drain select the shard's device; the caller's own HIP code must not find another device current)
	struct KeepDevice {
		int dev = -1;
		KeepDevice() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
		~KeepDevice() { if (dev >= 0) (void)hipSetDevice(dev); }
	};

	void stage(Shard &s, Call &c)
	{
		Part &p = c.parts[s.index];
		p.rc = LW_OK;
		p.out_elems = 0;
		p.launched = false;
		// this shard's packets of the call, in list order (every worker walks the caller's list for its own: the split runs on
		// all shards at once instead of on the caller's thread)
		p.idx.reserve(p.expect);
		p.pk.reserve(p.expect);
		for (size_t i = 0; i < c.n; i++) {
			const lw_shard_stream *st = c.pkts[i].stream;
			if (st->shard != s.index)
				continue;
			p.idx.push_back(i);
			p.pk.push_back(lw_packet{c.pkts[i].data, c.pkts[i].len, st->pwr});
		}
		if (p.pk.empty())
			return;
		p.rc = lw_ring_stage(s.ring, p.pk.data(), p.pk.size(), c.n_threads);
		if (p.rc != LW_OK)
			return;
		p.rc = lw_ring_launch(s.ring);
		std::lock_guard<std::mutex> g(s.ring_mu);
		if (p.rc != LW_OK) {
			drain_locked(s); // a device failure: nothing of this shard survives it (older parts find their generation gone)
			return;
		}
		p.launched = true;
		p.gen = s.gen;
		// sample counts and offsets are known once the batch is planned (lw_ring_stage): the sizes need no GPU
		p.out_elems = lw_ring_last_staged_elems(s.ring);
	}

	// runs on the shard's worker (copy-out form: the shards copy in parallel) or on the caller's thread (zero-copy form)
	void collect(Shard &s, Call &c)
	{
		Part &p = c.parts[s.index];
		if (!p.launched)
			return;
		std::lock_guard<std::mutex> g(s.ring_mu);
		if (p.gen != s.gen) { // the ring was drained after this part was launched: its slot and its PCM are gone
			p.launched = false;
			p.rc = LW_ERR_DEVICE;
			return;
		}
		const lw_packet_result *r = nullptr;
		const void *pcm = nullptr;
		size_t n = 0, elems = 0;
		int rc = lw_ring_collect(s.ring, &r, &n, &pcm, &elems);
		if (rc == LW_OK) {
			if (c.out && elems)
				std::memcpy((char *)c.out + p.base * esz, pcm, elems * esz);
			for (size_t k = 0; k < p.idx.size() && k < n; k++) {
				c.results[p.idx[k]] = r[k];
				if (!c.keep) // (zero-copy form: offsets stay relative to the shard's own block)
					c.results[p.idx[k]].out_offset += p.base;
			}
			if (c.keep) {
				c.pcm[s.index] = pcm;
				c.elems[s.index] = elems;
			} else {
				rc = lw_ring_release(s.ring);
				p.launched = false;
			}
		}
		if (rc != LW_OK) { // the slot would stay LAUNCHED for ever and wedge the ring after a few calls: start the ring over
			drain_locked(s);
			p.launched = false;
			p.rc = rc;
		}
	}

	// caller's thread: lw_ring_release only moves the ring's cursors
	void release(Shard &s, Call &c)
	{
		Part &p = c.parts[s.index];
		if (!p.launched)
			return;
		std::lock_guard<std::mutex> g(s.ring_mu);
		p.launched = false;
		if (p.gen != s.gen)
			return; // (drained meanwhile: nothing left to release)
		if (lw_ring_release(s.ring) != LW_OK) {
			drain_locked(s);
			if (p.rc == LW_OK)
				p.rc = LW_ERR_DEVICE;
		}
	}

	void worker_main(Shard *s)
	{
		(void)hipSetDevice(s->device);
		for (;;) {
			Job j;
			{
				std::unique_lock<std::mutex> g(s->mu);
				s->cv.wait(g, [&]() { return s->quit || !s->jobs.empty(); });
				if (s->quit)
					return;
				j = s->jobs.front();
				s->jobs.pop_front();
			}
			if (j.kind == JOB_STAGE)
				stage(*s, *j.call);
			else
				collect(*s, *j.call);
			std::lock_guard<std::mutex> g(j.call->mu);
			if (--j.call->pending == 0)
				j.call->cv.notify_one();
		}
	}

	// one job per shard that has something to do, wait for all of them
	void all_workers(JobKind kind, Call *c)
	{
		size_t n = 0;
		for (auto &s : shards) {
			const Part &p = c->parts[s->index];
			n += kind == JOB_STAGE ? p.expect != 0 : p.launched;
		}
		if (n == 0)
			return;
		{
			std::lock_guard<std::mutex> g(c->mu);
			c->pending = n;
		}
		for (auto &s : shards) {
			const Part &p = c->parts[s->index];
			if (kind == JOB_STAGE ? p.expect == 0 : !p.launched)
				continue;
			{
				std::lock_guard<std::mutex> g(s->mu);
				s->jobs.push_back(Job{kind, c});
			}
			s->cv.notify_one();
		}
		std::unique_lock<std::mutex> g(c->mu);
		c->cv.wait(g, [&]() { return c->pending == 0; });
	}
};

static std::atomic<int> g_share_cus{0};

extern "C" {

// measurement hook: 1 = logical shards of one device get their own CUs of every XCD each (tools/probe/sharder_probe.py)
void lw_debug_sharder_share_cus(int on)
{
	g_share_cus.store(on);
}

lw_sharder *lw_sharder_create(const lw_ident *id, const lw_setup *setup, const int *devices, size_t n_shards,
		size_t max_packets_per_shard, int fmt, int *err)
{
	int dummy;
	if (!err)
		err = &dummy;
	*err = LW_OK;
	if (!id || !setup || !devices || n_shards == 0 || n_shards > 1024 || max_packets_per_shard == 0 || fmt < 0 || fmt > 2) {
		*err = LW_ERR_NULL_ARG;
		return nullptr;
	}
	auto sh = std::make_unique<lw_sharder>();
	sh->max_packets = max_packets_per_shard;
	sh->fmt = fmt;
	sh->esz = fmt == LW_FMT_F32_PLANAR ? 4 : 2;
	for (size_t g = 0; g < n_shards; g++) {
		auto s = std::make_unique<Shard>();
		s->index = g;
		s->device = devices[g];
		int e = 0;
		s->dec = lw_decoder_create(id, setup, devices[g], &e);
		// logical shards on one device are tenants of that GPU (lw_decoder_set_shared_device): their rings hand the PCM copies to the
		// device's copier.  With the measurement hook each also gets its own CUs of every XCD (lw_decoder_set_cu_share: a shard's
		// whole-CU workgroups then never queue behind the other shard's entropy kernel -- k_long 28 us flat instead of 130-700 --
		// at the same packets/s, the link being the bound; not the default: profiles/r05_tenants.txt).
		unsigned same = 0, before = 0;
		for (size_t k = 0; k < n_shards; k++)
			if (devices[k] == devices[g]) {
				same++;
				before += k < g;
			}
		if (s->dec && same > 1)
			(void)lw_decoder_set_shared_device(s->dec, 1);
		if (s->dec && same > 1 && same <= 32 && g_share_cus.load())
			(void)lw_decoder_set_cu_share(s->dec, before, same);
		if (s->dec)
			s->ring = lw_ring_create(s->dec, LW_SHARD_SLOTS, max_packets_per_shard, fmt, &e);
		const bool ok = s->dec && s->ring;
		sh->shards.push_back(std::move(s));
		if (!ok) {
			*err = e ? e : LW_ERR_DEVICE;
			lw_sharder_destroy(sh.release());
			return nullptr;
		}
	}
	for (auto &s : sh->shards)
		s->worker = std::thread([p = sh.get(), q = s.get()]() { p->worker_main(q); });
	return sh.release();
}

void lw_sharder_destroy(lw_sharder *sh)
{
	if (!sh)
		return;
	for (auto &s : sh->shards) {
		{
			std::lock_guard<std::mutex> g(s->mu);
			s->quit = true;
		}
		s->cv.notify_all();
		if (s->worker.joinable())
			s->worker.join();
		if (s->ring) {
			(void)lw_ring_drain(s->ring);
			lw_ring_destroy(s->ring);
		}
		if (s->dec)
			lw_decoder_destroy(s->dec); // (streams opened on the shard must have been closed: their pwr lives in this decoder)
	}
	delete sh;
}

int lw_sharder_set_entropy_on_device(lw_sharder *sh, int on)
{
	if (!sh)
		return LW_ERR_NULL_ARG;
	std::lock_guard<std::mutex> call(sh->call_mu);
	if (!sh->calls.empty())
		return LW_ERR_CAPACITY; // only between calls
	for (auto &s : sh->shards)
		if (int rc = lw_ring_set_entropy_on_device(s->ring, on))
			return rc; // LW_ERR_UNSUPPORTED: the stream is not eligible (every shard has the same headers: none was switched)
	return LW_OK;
}

size_t lw_sharder_shards(const lw_sharder *sh)
{
	return sh ? sh->shards.size() : 0;
}

int lw_sharder_shard_cus(const lw_sharder *sh, size_t shard)
{
	return sh && shard < sh->shards.size() ? lw_decoder_cu_count(sh->shards[shard]->dec) : 0;
}

size_t lw_sharder_shard_of(const lw_sharder *sh, uint64_t stream_id)
{
	return sh && !sh->shards.empty() ? (size_t)(stream_id % sh->shards.size()) : 0;
}

int lw_sharder_device_of(const lw_sharder *sh, size_t shard)
{
	return sh && shard < sh->shards.size() ? sh->shards[shard]->device : -1;
}

lw_shard_stream *lw_sharder_stream_open(lw_sharder *sh, uint64_t stream_id)
{
	if (!sh)
		return nullptr;
	auto *st = new lw_shard_stream();
	st->owner = sh;
	st->id = stream_id;
	st->shard = lw_sharder_shard_of(sh, stream_id);
	st->pwr = lw_pwr_new(sh->shards[st->shard]->dec);
	if (!st->pwr) {
		delete st;
		return nullptr;
	}
	Shard &s = *sh->shards[st->shard];
	std::lock_guard<std::mutex> g(s.streams_mu);
	s.streams.push_back(st);
	return st;
}

void lw_sharder_stream_close(lw_shard_stream *st)
{
	if (!st)
		return;
	if (st->owner && st->shard < st->owner->shards.size()) {
		Shard &s = *st->owner->shards[st->shard];
		std::lock_guard<std::mutex> g(s.streams_mu);
		s.streams.erase(std::remove(s.streams.begin(), s.streams.end(), st), s.streams.end());
	}
	lw_pwr_free(st->pwr);
	delete st;
}

void lw_sharder_stream_reset(lw_shard_stream *st)
{
	if (st)
		lw_pwr_reset(st->pwr);
}

size_t lw_sharder_in_flight(lw_sharder *sh)
{
	if (!sh)
		return 0;
	std::lock_guard<std::mutex> call(sh->call_mu);
	return sh->calls.size();
}

static int submit_locked(lw_sharder *sh, const lw_shard_packet *pkts, size_t n, int n_threads_per_shard, size_t *out_elems)
{
	if (sh->calls.size() >= LW_SHARD_SLOTS)
		return LW_ERR_CAPACITY; // collect the oldest call first
	// (a cheap pass on the caller's thread: owners and per-shard counts, so that a bad list is refused before anything is staged)
	std::unique_ptr<Call> c = sh->fresh_call();
	c->n = n;
	c->pkts = pkts;
	int bad = LW_OK;
	for (size_t i = 0; i < n && bad == LW_OK; i++) {
		const lw_shard_stream *st = pkts[i].stream;
		if (!st || st->owner != sh)
			bad = LW_ERR_STATE_MISMATCH;
		else if (++c->parts[st->shard].expect > sh->max_packets)
			bad = LW_ERR_CAPACITY;
	}
	if (bad != LW_OK) {
		sh->spare.push_back(std::move(c));
		return bad;
	}
	// all shards run their host entropy stage at once (the worker pool serves their parallel regions side by side): by
	// default they share the CPUs this process may use
	c->n_threads = n_threads_per_shard > 0 ? n_threads_per_shard
			: std::max(1, lw_default_host_threads() / (int)sh->shards.size());
	sh->all_workers(JOB_STAGE, c.get()); // host entropy stage + asynchronous H2D / kernels / D2H of every shard
	int rc = LW_OK;
	size_t total = 0;
	for (Part &p : c->parts) {
		if (p.rc != LW_OK)
			rc = p.rc;
		p.base = total;
		total += p.out_elems;
	}
	c->total_elems = total;
	if (out_elems)
		*out_elems = total;
	c->pkts = nullptr;
	// (on an error the shards that did launch still hold a slot each: the call stays in the queue so that a collect frees them)
	sh->calls.push_back(std::move(c));
	return rc;
}

static int collect_locked(lw_sharder *sh, void *out, size_t cap_elems, lw_packet_result *results, size_t n_results)
{
	if (sh->calls.empty())
		return LW_ERR_CAPACITY; // nothing in flight
	Call &c = *sh->calls.front();
	if (c.collected)
		return LW_ERR_CAPACITY; // held by lw_sharder_collect_pinned: release it first
	if (n_results < c.n || (!results && c.n))
		return LW_ERR_NULL_ARG;
	if (out && cap_elems < c.total_elems) // (out == NULL: the samples are dropped, only statuses and counts come back)
		return LW_ERR_CAPACITY; // nothing consumed: call again with room for lw_sharder_submit's out_elems
	c.out = out;
	c.results = results;
	for (size_t i = 0; i < c.n; i++) // (packets of a shard whose stage failed keep this)
		results[i] = lw_packet_result{LW_ERR_DEVICE, 0, 0};
	sh->all_workers(JOB_COLLECT, &c);
	int rc = LW_OK;
	for (Part &p : c.parts)
		if (p.rc != LW_OK)
			rc = p.rc;
	sh->retire_front();
	return rc;
}

int lw_sharder_collect_pinned(lw_sharder *sh, lw_packet_result *results, size_t n_results, const void **pcm, size_t *elems)
{
	if (!sh || !pcm || !elems)
		return LW_ERR_NULL_ARG;
	std::lock_guard<std::mutex> call(sh->call_mu);
	lw_sharder::KeepDevice keep_dev;
	if (sh->calls.empty())
		return LW_ERR_CAPACITY;
	Call &c = *sh->calls.front();
	if (c.collected)
		return LW_ERR_CAPACITY; // release it first
	if (n_results < c.n || (!results && c.n))
		return LW_ERR_NULL_ARG;
	c.out = nullptr;
	c.results = results;
	c.keep = true;
	c.pcm = pcm;
	c.elems = elems;
	for (size_t g = 0; g < sh->shards.size(); g++) {
		pcm[g] = nullptr;
		elems[g] = 0;
	}
	for (size_t i = 0; i < c.n; i++)
		results[i] = lw_packet_result{LW_ERR_DEVICE, 0, 0};
	for (auto &s : sh->shards) // nothing to copy: the caller's thread waits for every shard's slot itself (no hand-over to the workers)
		sh->collect(*s, c);
	c.collected = true;
	int rc = LW_OK;
	for (Part &p : c.parts)
		if (p.rc != LW_OK)
			rc = p.rc;
	return rc;
}

int lw_sharder_release(lw_sharder *sh)
{
	if (!sh)
		return LW_ERR_NULL_ARG;
	std::lock_guard<std::mutex> call(sh->call_mu);
	lw_sharder::KeepDevice keep_dev;
	if (sh->calls.empty() || !sh->calls.front()->collected)
		return LW_ERR_CAPACITY;
	Call &c = *sh->calls.front();
	for (auto &s : sh->shards)
		sh->release(*s, c);
	int rc = LW_OK;
	for (Part &p : c.parts)
		if (p.rc != LW_OK)
			rc = p.rc;
	sh->retire_front();
	return rc;
}

int lw_sharder_submit(lw_sharder *sh, const lw_shard_packet *pkts, size_t n, int n_threads_per_shard, size_t *out_elems)
{
	if (!sh || (!pkts && n))
		return LW_ERR_NULL_ARG;
	std::lock_guard<std::mutex> call(sh->call_mu);
	return submit_locked(sh, pkts, n, n_threads_per_shard, out_elems);
}

int lw_sharder_collect(lw_sharder *sh, void *out, size_t cap_elems, lw_packet_result *results, size_t n_results)
{
	if (!sh)
		return LW_ERR_NULL_ARG;
	std::lock_guard<std::mutex> call(sh->call_mu);
	return collect_locked(sh, out, cap_elems, results, n_results);
}

int lw_sharder_decode(lw_sharder *sh, const lw_shard_packet *pkts, size_t n, int n_threads_per_shard, void *out,
		size_t cap_elems, lw_packet_result *results)
{
	if (!sh || (!pkts && n) || (!results && n) || (!out && cap_elems))
		return LW_ERR_NULL_ARG;
	std::lock_guard<std::mutex> call(sh->call_mu);
	if (!sh->calls.empty())
		return LW_ERR_CAPACITY; // the synchronous form needs an empty pipeline
	size_t total = 0;
	int rc = submit_locked(sh, pkts, n, n_threads_per_shard, &total);
	if (rc == LW_OK && total > cap_elems)
		rc = LW_ERR_CAPACITY; // (the host halves of the streams' states have advanced: the call cannot be repeated as is)
	if (sh->calls.empty())
		return rc;
	if (rc != LW_OK) { // free the slots the shards hold; the results are dropped
		Call &c = *sh->calls.front();
		std::vector<lw_packet_result> scratch(c.n);
		c.out = nullptr;
		c.results = scratch.data();
		sh->all_workers(JOB_COLLECT, &c);
		sh->retire_front();
		return rc;
	}
	return collect_locked(sh, out, cap_elems, results, n);
}

} // extern "C"
