// Persistent worker threads of the host entropy stage: see lw_pool.hpp.  Product code.
#include "lw_pool.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

namespace lw {

namespace {
inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
	__builtin_ia32_pause();
#else
	std::this_thread::yield();
#endif
}
} // namespace

struct EntropyPool::Impl {
	std::mutex mu, serial;
	std::condition_variable cv, cv_deep;
	std::vector<std::thread> threads;
	const std::function<void()> *fn = nullptr;
	// each on a cache line of its own: the helpers spin on `state` while finished helpers count `pending` down (on one
	// line every count-down would invalidate every spinner's copy, across both sockets of a 2 x 64-core host)
	alignas(128) std::atomic<size_t> pending{0};
	alignas(128) std::atomic<uint64_t> state{0}; // (region number << 16) | helpers of that region
	alignas(128) std::atomic<unsigned> sleepers{0};
	alignas(128) char pad_[8] = {0};
	unsigned awake_upto = 0; // threads with id >= awake_upto may be in deep sleep (written under `serial`)
	// how long a helper spins for the next region before it sleeps (LW_POOL_SPIN_US overrides).  Longer spins were measured
	// on the GPU box (2 x EPYC 9575F shared with other jobs, load average 20+): 3 ms instead of 100 us made every
	// configuration slower -- the spinning helpers take the cores the box's other work needs, and then their own.
	long spin_us = 100;
	Impl()
	{
		if (const char *e = getenv("LW_POOL_SPIN_US"))
			spin_us = std::max(0L, atol(e));
	}

	void loop(size_t id, uint64_t seen)
	{
		bool deep = false;
		for (;;) {
			uint64_t st = 0;
			if (deep) {
				// not needed by the last region: sleep until a region needs this id (regions in between are not looked at)
				std::unique_lock<std::mutex> g(mu);
				cv_deep.wait(g, [&]() {
					st = state.load();
					return (st >> 16) != seen && id < (st & 0xffffu);
				});
			} else {
				// wait for the next region: spin for spin_us (batches arrive back to back), then sleep
				bool got = false;
				const auto t0 = std::chrono::steady_clock::now();
				for (unsigned spins = 0;; spins++) {
					st = state.load(std::memory_order_acquire);
					if ((st >> 16) != seen) {
						got = true;
						break;
					}
					if ((spins & 63u) == 63u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us))
						break;
					cpu_relax();
				}
				if (!got) {
					std::unique_lock<std::mutex> g(mu);
					sleepers.fetch_add(1);
					cv.wait(g, [&]() { return ((st = state.load()) >> 16) != seen; });
					sleepers.fetch_sub(1);
				}
			}
			seen = st >> 16;
			// helpers of region `seen` are the threads with id < (st & 0xffff): run() keeps fn unchanged until all of them
			// have counted down; any other thread must not look at fn (the region may be over already)
			if (id < (st & 0xffffu)) {
				(*fn)();
				pending.fetch_sub(1, std::memory_order_acq_rel);
				deep = false;
			} else {
				deep = true;
			}
		}
	}
};

EntropyPool::Impl *EntropyPool::impl()
{
	if (!p_)
		p_ = new Impl(); // (entropy_pool() constructs the pool once, under the static-initialisation lock)
	return p_;
}

unsigned EntropyPool::threads_created() const
{
	return p_ ? (unsigned)p_->threads.size() : 0;
}

void EntropyPool::run(unsigned n, const std::function<void()> &fn)
{
	n = std::min(n, MAX_THREADS);
	if (n <= 1) {
		fn();
		return;
	}
	Impl &I = *impl();
	std::unique_lock<std::mutex> serial(I.serial); // one parallel region at a time
	const uint64_t epoch = (I.state.load() >> 16) + 1;
	const unsigned helpers = n - 1;
	{
		std::unique_lock<std::mutex> g(I.mu);
		while (I.threads.size() < helpers)
			I.threads.emplace_back([&I, id = I.threads.size(), epoch]() { I.loop(id, epoch - 1); });
	}
	I.fn = &fn;
	I.pending.store(helpers);
	I.state.store((epoch << 16) | helpers); // region number and its helper count in one word (seq_cst, see loop())
	if (I.sleepers.load() > 0) {
		{ std::unique_lock<std::mutex> g(I.mu); } // a worker between its predicate check and its wait holds mu
		I.cv.notify_all();
	}
	if (helpers > I.awake_upto) { // some of the threads this region needs may be in deep sleep
		{ std::unique_lock<std::mutex> g(I.mu); }
		I.cv_deep.notify_all();
	}
	I.awake_upto = helpers; // threads beyond this region's helpers go (or stay) deep
	fn();
	for (unsigned spins = 0; I.pending.load(std::memory_order_acquire) != 0; spins++) {
		if (spins < 4096)
			cpu_relax();
		else
			std::this_thread::yield();
	}
}

EntropyPool &entropy_pool()
{
	static EntropyPool *p = new EntropyPool(); // never destroyed: its threads are detached from process teardown
	(void)p->threads_created();
	return *p;
}

} // namespace lw
