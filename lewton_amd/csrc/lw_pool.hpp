// Persistent worker threads of the host entropy stage (product code).
//
// lw_batch_entropy runs a fraction of a millisecond per 4096-packet batch on a many-core host; creating and joining
// 32-64 threads for every batch cost a fifth of that.  The pool grows on demand and its threads live until the process
// exits.  A thread that took part in the last parallel region spins on the region counter for a short while (batches
// arrive back to back) before it sleeps; a thread the last region did NOT need (the pool once served a wider region)
// goes into a deep sleep on its own condition variable and is only woken when a region needs its id again -- a caller
// that asks for 32 threads is not taxed with waking 200 idle ones per batch.
#pragma once

#include <functional>

namespace lw {

class EntropyPool {
public:
	static constexpr unsigned MAX_THREADS = 1024;
	// runs fn() on `n` threads in total (the caller is one of them) and returns when all have finished;
	// parallel regions of different callers are serialised
	void run(unsigned n, const std::function<void()> &fn);
	unsigned threads_created() const;

private:
	struct Impl;
	Impl *impl();
	Impl *p_ = nullptr;
};

EntropyPool &entropy_pool();

} // namespace lw
