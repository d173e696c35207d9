// Sharding of independent streams over the GPUs of one node, in one process (product code; include/lewton_amd.h lw_sharder_*).
//
// A stream's decode touches only its own PreviousWindowRight and the immutable headers (audio.rs:919), so streams never
// exchange data: shard g owns the streams with stream_id mod G == g (BASELINE configs[4]).  A shard = one device
// context (lw_decoder: tables and the state pool of its streams in that GPU's HBM), one batch with pinned staging, one HIP
// stream and one worker thread.  lw_sharder_decode splits a list of packets by owner, runs every shard's host entropy stage,
// H2D, kernels and D2H on its own thread and device -- all shards at once, no collective, nothing crossing xGMI -- and
// returns when the last shard is done.  A device may be listed several times (logical shards): that is how the N > 1 logic
// is tested on a one-GPU box.  The process-per-GPU form (bench.py under torch.distributed.run) uses the same rule through
// lewton_amd/shard.py; this is the single-process form INTEGRATION.md section 3 describes.
#include "../../include/lewton_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
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

struct Shard {
	int device = 0;
	lw_decoder *dec = nullptr;
	lw_batch *batch = nullptr;
	hipStream_t stream = nullptr;
	std::thread worker;
	// work of the current call
	std::vector<size_t> idx;       // positions (in the caller's packet list) of this shard's packets, in list order
	std::vector<lw_packet> pk;
	size_t out_elems = 0, base = 0; // elements this shard produces / where its first block starts in the caller's buffer
	int rc = LW_OK;
};

} // namespace

struct lw_sharder {
	std::vector<std::unique_ptr<Shard>> shards;
	size_t max_packets = 0;
	int fmt = 0;
	size_t esz = 2;
	// one call at a time; the workers walk through phase 1 (host entropy stage) and phase 2 (device) of it
	std::mutex mu;
	std::condition_variable cv;
	uint64_t phase = 0;     // 2 * call + {1, 2}; the workers run phase p when they see phase == p
	size_t done = 0;        // workers that finished the current phase
	bool quit = false;
	int n_threads = 0;
	void *out = nullptr;
	std::mutex call_mu;     // serialises lw_sharder_decode callers
	uint64_t call_no = 0;

	void run_phase(Shard &s, bool device_phase)
	{
		if (!device_phase) {
			s.rc = LW_OK;
			s.out_elems = 0;
			if (s.pk.empty())
				return;
			s.rc = lw_batch_entropy(s.batch, s.pk.data(), s.pk.size(), n_threads);
			if (s.rc == LW_OK)
				s.out_elems = lw_batch_out_elems(s.batch);
			return;
		}
		if (s.pk.empty() || s.rc != LW_OK)
			return;
		if (hipSetDevice(s.device) != hipSuccess) {
			s.rc = LW_ERR_DEVICE;
			return;
		}
		s.rc = lw_batch_upload(s.batch, s.stream);
		if (s.rc == LW_OK) // kernels -> internal device buffer -> D2H straight into the caller's buffer at this shard's base
			s.rc = lw_batch_synth_to_host(s.batch, (char *)out + s.base * esz, s.out_elems, s.stream);
	}

	void worker_main(Shard *s)
	{
		(void)hipSetDevice(s->device);
		uint64_t seen = 0;
		for (;;) {
			uint64_t p;
			{
				std::unique_lock<std::mutex> g(mu);
				cv.wait(g, [&]() { return quit || phase != seen; });
				if (quit)
					return;
				p = seen = phase;
			}
			run_phase(*s, (p & 1) == 0);
			std::unique_lock<std::mutex> g(mu);
			done++;
			cv.notify_all();
		}
	}

	void all_workers(uint64_t p)
	{
		std::unique_lock<std::mutex> g(mu);
		phase = p;
		done = 0;
		cv.notify_all();
		cv.wait(g, [&]() { return done == shards.size(); });
	}
};

extern "C" {

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
		s->device = devices[g];
		int e = 0;
		s->dec = lw_decoder_create(id, setup, devices[g], &e);
		if (s->dec)
			s->batch = lw_batch_create(s->dec, max_packets_per_shard, fmt, &e);
		const bool ok = s->dec && s->batch && hipSetDevice(devices[g]) == hipSuccess &&
			hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) == hipSuccess;
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
	{
		std::unique_lock<std::mutex> g(sh->mu);
		sh->quit = true;
		sh->cv.notify_all();
	}
	for (auto &s : sh->shards) {
		if (s->worker.joinable())
			s->worker.join();
		(void)hipSetDevice(s->device);
		if (s->stream) {
			(void)hipStreamSynchronize(s->stream);
			(void)hipStreamDestroy(s->stream);
		}
		if (s->batch)
			lw_batch_destroy(s->batch);
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
	for (auto &s : sh->shards)
		if (int rc = lw_batch_set_entropy_on_device(s->batch, on))
			return rc; // LW_ERR_UNSUPPORTED: the stream is not eligible (every shard has the same headers: none was switched)
	return LW_OK;
}

size_t lw_sharder_shards(const lw_sharder *sh)
{
	return sh ? sh->shards.size() : 0;
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
	return st;
}

void lw_sharder_stream_close(lw_shard_stream *st)
{
	if (!st)
		return;
	lw_pwr_free(st->pwr);
	delete st;
}

void lw_sharder_stream_reset(lw_shard_stream *st)
{
	if (st)
		lw_pwr_reset(st->pwr);
}

int lw_sharder_decode(lw_sharder *sh, const lw_shard_packet *pkts, size_t n, int n_threads_per_shard, void *out,
		size_t cap_elems, lw_packet_result *results)
{
	if (!sh || (!pkts && n) || (!results && n) || (!out && cap_elems))
		return LW_ERR_NULL_ARG;
	std::lock_guard<std::mutex> call(sh->call_mu);
	for (auto &s : sh->shards) {
		s->idx.clear();
		s->pk.clear();
	}
	for (size_t i = 0; i < n; i++) {
		const lw_shard_stream *st = pkts[i].stream;
		if (!st || st->owner != sh)
			return LW_ERR_STATE_MISMATCH;
		Shard &s = *sh->shards[st->shard];
		if (s.idx.size() == sh->max_packets)
			return LW_ERR_CAPACITY;
		s.idx.push_back(i);
		s.pk.push_back(lw_packet{pkts[i].data, pkts[i].len, st->pwr});
	}
	// all shards run their host entropy stage at once (the worker pool serves their parallel regions side by side): by
	// default they share the CPUs this process may use
	sh->n_threads = n_threads_per_shard > 0 ? n_threads_per_shard
			: std::max(1, lw_default_host_threads() / (int)sh->shards.size());
	sh->out = out;
	const uint64_t base_phase = 2 * (++sh->call_no);
	sh->all_workers(base_phase + 1); // phase 1: host entropy stage of every shard (sample counts, offsets inside the shard)
	size_t total = 0;
	int rc = LW_OK;
	for (auto &s : sh->shards) {
		if (s->rc != LW_OK)
			rc = s->rc;
		s->base = total;
		total += s->out_elems;
	}
	if (rc == LW_OK && total > cap_elems)
		rc = LW_ERR_CAPACITY; // (the host halves of the streams' states have advanced: the call cannot be repeated as is)
	if (rc != LW_OK)
		return rc;
	sh->all_workers(base_phase + 2); // phase 2: H2D, kernels, D2H of every shard, each on its own device and thread
	for (auto &s : sh->shards) {
		if (s->rc != LW_OK)
			rc = s->rc;
		if (s->pk.empty())
			continue;
		const lw_packet_result *r = lw_batch_results(s->batch);
		for (size_t k = 0; k < s->idx.size(); k++) {
			results[s->idx[k]] = r[k];
			results[s->idx[k]].out_offset += s->base;
		}
	}
	return rc;
}

} // extern "C"
