// Persistent worker threads of the host entropy stage: see lw_pool.hpp.  Product code.
#include "lw_pool.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

#include <sched.h>

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

// One open parallel region.  The slots are persistent (helpers may look at a slot at any time); a slot is reused for
// another region only after its previous region is over, and `ticket` tells the two apart.
struct alignas(128) Region {
	std::atomic<uint64_t> ticket{0}; // (generation << 1) | open
	std::atomic<unsigned> active{0}; // helpers inside fn (or between their join and the re-check of the ticket)
	std::atomic<unsigned> joined{0}; // helpers that have asked to join this generation
	std::atomic<unsigned> want{0};   // helpers this generation takes (written before the ticket is published)
	const std::function<void()> *fn = nullptr;
	bool busy = false;               // slot taken by a caller (under Impl::mu)
};

} // namespace

struct EntropyPool::Impl {
	static constexpr unsigned SLOTS = 32;
	Region slot[SLOTS];
	// bumped whenever a region opens: idle helpers look at the slots again
	alignas(128) std::atomic<uint64_t> epoch{0};
	alignas(128) std::atomic<unsigned> sleepers{0};
	alignas(128) std::atomic<unsigned> spinners{0}; // helpers awake and without work
	alignas(128) char pad_[8] = {0};
	std::mutex mu;
	std::condition_variable cv;
	std::vector<std::thread> threads; // under mu
	unsigned demand = 0;              // helpers wanted by the open regions together (under mu)
	// how long a helper without work spins before it sleeps.  Longer spins were measured on the GPU box (2 x EPYC 9575F shared
	// with other jobs, load average 20+): 3 ms instead of 100 us made every configuration slower -- the spinning helpers
	// take the cores the box's other work needs, and then their own.
	static constexpr long spin_us = 100;

	// joins every open region that still wants helpers; true if this thread did any work
	bool serve()
	{
		bool worked = false;
		for (Region &r : slot) {
			const uint64_t t = r.ticket.load(std::memory_order_acquire);
			if (!(t & 1u) || r.joined.load(std::memory_order_relaxed) >= r.want.load(std::memory_order_relaxed))
				continue;
			// Announce first, then look again: the caller closes the ticket and THEN waits for active == 0 (both sequentially
			// consistent), so either this thread sees the ticket changed and backs off, or the caller sees it active and
			// waits -- fn and the region's captures stay alive while a helper is inside.
			r.active.fetch_add(1, std::memory_order_seq_cst);
			if (r.ticket.load(std::memory_order_seq_cst) == t && r.joined.fetch_add(1, std::memory_order_acq_rel) < r.want.load(std::memory_order_relaxed)) {
				(*r.fn)();
				worked = true;
			}
			r.active.fetch_sub(1, std::memory_order_seq_cst);
		}
		return worked;
	}

	void loop()
	{
		for (;;) {
			const uint64_t e = epoch.load(std::memory_order_seq_cst); // before the scan: a region opened after it bumps the epoch
			if (serve())
				continue;
			// nothing to do: spin for spin_us waiting for the next region, then sleep
			spinners.fetch_add(1, std::memory_order_relaxed);
			bool got = false;
			const auto t0 = std::chrono::steady_clock::now();
			for (unsigned spins = 0;; spins++) {
				if (epoch.load(std::memory_order_acquire) != e) {
					got = true;
					break;
				}
				if ((spins & 63u) == 63u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us))
					break;
				cpu_relax();
			}
			spinners.fetch_sub(1, std::memory_order_relaxed);
			if (!got) {
				std::unique_lock<std::mutex> g(mu);
				sleepers.fetch_add(1, std::memory_order_seq_cst);
				cv.wait(g, [&]() { return epoch.load(std::memory_order_seq_cst) != e; });
				sleepers.fetch_sub(1, std::memory_order_seq_cst);
			}
		}
	}
};

EntropyPool::EntropyPool() : p_(new Impl()) {}

unsigned EntropyPool::threads_created() const
{
	std::unique_lock<std::mutex> g(p_->mu);
	return (unsigned)p_->threads.size();
}

void EntropyPool::run(unsigned n, const std::function<void()> &fn)
{
	n = std::min(n, MAX_THREADS);
	if (n <= 1) {
		fn();
		return;
	}
	Impl &I = *p_;
	const unsigned helpers = n - 1;
	Region *r = nullptr;
	{
		std::unique_lock<std::mutex> g(I.mu);
		for (Region &s : I.slot)
			if (!s.busy) {
				r = &s;
				break;
			}
		if (r) {
			r->busy = true;
			I.demand += helpers;
			while (I.threads.size() < std::min(I.demand, MAX_THREADS))
				I.threads.emplace_back([&I]() { I.loop(); });
		}
	}
	if (!r) { // more concurrent callers than slots: this one works alone
		fn();
		return;
	}
	const uint64_t gen = (r->ticket.load(std::memory_order_relaxed) >> 1) + 1;
	r->fn = &fn;
	r->want.store(helpers, std::memory_order_relaxed);
	r->joined.store(0, std::memory_order_relaxed);
	r->ticket.store((gen << 1) | 1u, std::memory_order_seq_cst); // open
	I.epoch.fetch_add(1, std::memory_order_seq_cst);
	// wake sleepers for the helpers the spinning threads cannot supply
	const unsigned awake = I.spinners.load(std::memory_order_relaxed);
	if (helpers > awake && I.sleepers.load(std::memory_order_seq_cst) > 0) {
		unsigned need = helpers - awake;
		std::unique_lock<std::mutex> g(I.mu); // (a helper between its predicate check and its wait holds mu)
		if (need >= I.sleepers.load(std::memory_order_relaxed))
			I.cv.notify_all();
		else
			while (need--)
				I.cv.notify_one();
	}
	fn();
	r->ticket.store(gen << 1, std::memory_order_seq_cst); // closed: nobody joins any more
	for (unsigned spins = 0; r->active.load(std::memory_order_seq_cst) != 0; spins++) {
		if (spins < 4096)
			cpu_relax();
		else
			std::this_thread::yield();
	}
	std::unique_lock<std::mutex> g(I.mu);
	r->busy = false;
	I.demand -= helpers;
}

namespace {

// quota / period of the container's CPU controller, in CPUs; 0 = unlimited or unknown
double cgroup_cpu_limit()
{
	if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) { // cgroup v2: "<quota|max> <period>"
		char quota[32] = {0};
		long period = 0;
		const int got = fscanf(f, "%31s %ld", quota, &period);
		fclose(f);
		if (got == 2 && period > 0 && quota[0] != 'm')
			return atof(quota) / (double)period;
		return 0.0;
	}
	long quota = -1, period = 0; // cgroup v1
	if (FILE *f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
		if (fscanf(f, "%ld", &quota) != 1)
			quota = -1;
		fclose(f);
	}
	if (FILE *f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
		if (fscanf(f, "%ld", &period) != 1)
			period = 0;
		fclose(f);
	}
	return quota > 0 && period > 0 ? (double)quota / (double)period : 0.0;
}

} // namespace

unsigned default_host_threads()
{
	static const unsigned n = []() {
		if (const char *e = getenv("LW_HOST_THREADS"))
			if (atoi(e) > 0)
				return (unsigned)std::min(atoi(e), (int)EntropyPool::MAX_THREADS);
		unsigned t = std::max(1u, std::thread::hardware_concurrency());
		cpu_set_t set;
		CPU_ZERO(&set);
		if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0)
			t = std::min(t, (unsigned)CPU_COUNT(&set));
		const double lim = cgroup_cpu_limit();
		if (lim > 0.0)
			t = std::min(t, std::max(1u, (unsigned)(lim + 0.5)));
		return t;
	}();
	return n;
}

EntropyPool &entropy_pool()
{
	static EntropyPool *p = new EntropyPool(); // never destroyed: its threads are detached from process teardown
	return *p;
}

} // namespace lw
