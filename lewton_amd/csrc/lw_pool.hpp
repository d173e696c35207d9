// Persistent worker threads of the host entropy stage (product code).
//
// lw_batch_entropy runs a fraction of a millisecond per 4096-packet batch on a many-core host; creating and joining
// 32-64 threads for every batch cost a fifth of that.  The pool's threads are created on demand and live until the
// process exits.
//
// A parallel region is a function that pulls work from a shared counter until none is left (lw_batch_entropy's worker), so
// the caller only ever waits for helpers that have actually JOINED the region: a helper the host scheduler has not woken
// yet (the GPU box is a shared 2-socket machine) costs the region nothing -- the others take its share -- where a fixed
// fork/join would put the slowest wake-up on every batch's critical path.  Several regions may be open at once (one per
// calling thread: the rings of independent callers, the shards of lw_sharder_decode); idle helpers join whichever open
// region still wants helpers.  Helpers spin for a short while between regions (batches arrive back to back), then sleep;
// a region wakes only as many sleepers as it wants helpers.
#pragma once

#include <functional>

namespace lw {

class EntropyPool {
public:
	static constexpr unsigned MAX_THREADS = 1024;
	EntropyPool();
	// runs fn() on the calling thread and on up to n - 1 helpers, concurrently; returns when the caller's fn() has returned
	// and every helper that entered fn() has left it.  fn must be a work-sharing loop: it may be entered by fewer than n
	// threads (even by the caller alone) and must then still do all the work.
	void run(unsigned n, const std::function<void()> &fn);
	unsigned threads_created() const;

private:
	struct Impl;
	Impl *p_;
};

EntropyPool &entropy_pool();

// Threads the host entropy stage uses when the caller does not say (n_threads <= 0): the CPUs this process may actually run
// on -- hardware threads, cut down to the affinity mask and to the container's CFS quota (cgroup cpu.max; the GPU boxes
// give a container 16 CPUs' worth of a 256-thread host, and 32 runnable threads there are throttled for tens of
// milliseconds at a time).  LW_HOST_THREADS overrides.
unsigned default_host_threads();

} // namespace lw
