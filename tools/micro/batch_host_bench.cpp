// Host side of the batch path without a GPU: the product's lw_runtime.cpp / lw_entropy.cpp / lw_headers.cpp / lw_fast.cpp
// linked against tests/san/hip_standins.inc (device memory = malloc, copies = memcpy, kernel launchers = no-ops), to time
// lw_batch_entropy -- prologue pass, threaded entropy decode into the staging slab, planning pass -- on this machine's
// cores.  A profiling tool only (tools/batch_host_bench.py builds and runs it); nothing here is shipped or tested against.
//   usage: batch_host_bench case.bin [packets 4096] [streams 256] [reps 20] [0] [threads...]
// With LW_HOST_BENCH_CHECK=1 in the environment it does not time anything: it runs the batch on one thread and on every
// listed thread count and compares statuses, sample counts, output offsets and the staged residue vectors (against each
// other and against lw_entropy_decode_host packet by packet); tests/test_host_batch.py runs that under ThreadSanitizer
// (LW_HOST_BENCH_FILE_ORDER=1: packets in file order; LW_HOST_BENCH_DUMP=1: print status / sample count / offset per packet).
#include "../../include/lewton_amd.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../tests/san/hip_standins.inc"

static bool rd(FILE *f, uint32_t &v) { return fread(&v, 4, 1, f) == 1; }
static bool rdv(FILE *f, std::vector<uint8_t> &b)
{
	uint32_t n;
	if (!rd(f, n))
		return false;
	b.resize(n);
	return n == 0 || fread(b.data(), 1, n, f) == n;
}

int main(int argc, char **argv)
{
	if (argc < 2)
		return 2;
	FILE *f = fopen(argv[1], "rb");
	if (!f)
		return 2;
	const size_t NP = argc > 2 ? atoi(argv[2]) : 4096, S = argc > 3 ? atoi(argv[3]) : 256;
	const int reps = argc > 4 ? atoi(argv[4]) : 20;
	const int sym = 0; // (argv[5]: former record-format switch, kept as a placeholder so that the thread list stays at argv[6..])
	uint32_t nc, npk;
	std::vector<uint8_t> idp, stp;
	if (!rd(f, nc) || !rdv(f, idp) || !rdv(f, stp) || !rd(f, npk))
		return 2;
	std::vector<std::vector<uint8_t>> pool(npk);
	for (auto &p : pool)
		if (!rdv(f, p))
			return 2;
	int err = 0;
	lw_ident *id = lw_read_header_ident(idp.data(), idp.size(), &err);
	lw_ident_info info;
	lw_ident_get_info(id, &info);
	lw_setup *st = lw_read_header_setup(stp.data(), stp.size(), info.audio_channels, info.blocksize_0, info.blocksize_1, &err);
	lw_decoder *dec = lw_decoder_create(id, st, 0, &err);
	if (!dec) {
		printf("decoder: %d\n", err);
		return 1;
	}
	lw_batch *b = lw_batch_create(dec, NP, LW_FMT_I16_PLANAR, &err);
	std::vector<lw_pwr *> pwr(S);
	for (auto &p : pwr)
		p = lw_pwr_new(dec);
	std::vector<lw_packet> pk(NP);
	unsigned x = 12345;
	const bool file_order = getenv("LW_HOST_BENCH_FILE_ORDER") != nullptr; // packet k = k-th packet of the file (tests)
	for (size_t k = 0; k < NP; k++) {
		x = x * 1664525u + 1013904223u;
		const auto &p = pool[file_order ? k % npk : (x >> 8) % npk];
		pk[k] = {p.data(), p.size(), pwr[k / (NP / S)]};
	}
	std::vector<int> threads;
	for (int a = 6; a < argc; a++)
		threads.push_back(atoi(argv[a]));
	if (threads.empty())
		threads = {1, 2, 4, 8};
	if (getenv("LW_HOST_BENCH_CHECK")) {
		const size_t ch = info.audio_channels, cap = ch * ((size_t)1 << info.blocksize_1) / 2;
		struct Snap {
			std::vector<lw_packet_result> res;
			std::vector<float> vec;
		};
		auto snap = [&](int nt, Snap &o) {
			for (auto *p : pwr)
				lw_pwr_reset(p);
			if (int rc = lw_batch_entropy(b, pk.data(), NP, nt)) {
				printf("lw_batch_entropy(%d threads): %d\n", nt, rc);
				exit(1);
			}
			o.res.assign(lw_batch_results(b), lw_batch_results(b) + NP);
			o.vec.assign(NP * cap, 0.0f);
			if (!sym)
				for (size_t k = 0; k < NP; k++)
					if (o.res[k].status == LW_OK && lw_batch_tap(b, k, LW_TAP_RESIDUE_PRE_INVERSE, &o.vec[k * cap], cap)) {
						printf("tap %zu failed\n", k);
						exit(1);
					}
		};
		Snap ref;
		snap(1, ref);
		// SURVEY 8(d) bytes of the batch, and the window state crossing HBM at its launch boundary (every stream was reset: nothing
		// comes in, every stream whose last packet decoded leaves its right part behind)
		printf("bytes %llu %llu\n", (unsigned long long)lw_batch_algorithmic_bytes(b), (unsigned long long)lw_batch_state_bytes(b));
		if (getenv("LW_HOST_BENCH_DUMP"))
			for (size_t k = 0; k < NP; k++)
				printf("R %d %u %llu\n", ref.res[k].status, ref.res[k].n_samples, (unsigned long long)ref.res[k].out_offset);
		size_t ok = 0;
		std::vector<uint16_t> fo(ch * lw_setup_floor_stride(st));
		std::vector<float> one(cap), curve(cap);
		for (size_t k = 0; k < NP && !sym; k++) { // the batch path against the one-packet host hook
			uint8_t bs = 0, mode = 0, flags = 0;
			uint64_t bits = 0;
			const int rc = lw_entropy_decode_host(id, st, pk[k].data, pk[k].len, fo.data(), one.data(), cap, &bs, &mode, &flags, &bits,
					curve.data());
			// (a packet can decode here and still fail in the batch: window / state checks of audio.rs:1107-1111)
			if (ref.res[k].status == LW_OK) {
				ok++;
				if (rc != LW_OK || memcmp(one.data(), &ref.vec[k * cap], sizeof(float) * ch * ((size_t)1 << bs) / 2)) {
					printf("packet %zu: batch and single-packet host stage differ\n", k);
					return 1;
				}
			}
		}
		for (int nt : threads) {
			for (int rep = 0; rep < std::max(1, reps); rep++) {
				Snap got;
				snap(nt, got);
				for (size_t k = 0; k < NP; k++)
					if (got.res[k].status != ref.res[k].status || got.res[k].n_samples != ref.res[k].n_samples ||
							got.res[k].out_offset != ref.res[k].out_offset) {
						printf("packet %zu: result differs on %d threads\n", k, nt);
						return 1;
					}
				if (got.vec != ref.vec) {
					printf("staged residue differs on %d threads\n", nt);
					return 1;
				}
			}
		}
		{ // two callers, each with its own batch and streams, share the decoder and the worker pool
			lw_batch *b2 = lw_batch_create(dec, NP, LW_FMT_I16_PLANAR, &err);
			std::vector<lw_pwr *> pwr2(S);
			for (auto &p : pwr2)
				p = lw_pwr_new(dec);
			std::vector<lw_packet> pk2 = pk;
			for (size_t k = 0; k < NP; k++)
				pk2[k].pwr = pwr2[k / (NP / S)];
			std::vector<lw_packet_result> r1, r2;
			bool bad = false;
			auto drive = [&](lw_batch *bb, std::vector<lw_packet> &pp, std::vector<lw_pwr *> &pw, std::vector<lw_packet_result> &out) {
				for (int rep = 0; rep < 4; rep++) {
					for (auto *p : pw)
						lw_pwr_reset(p);
					if (lw_batch_entropy(bb, pp.data(), NP, threads.empty() ? 4 : threads.back()))
						bad = true;
					out.assign(lw_batch_results(bb), lw_batch_results(bb) + NP);
				}
			};
			std::thread t1([&]() { drive(b, pk, pwr, r1); }), t2([&]() { drive(b2, pk2, pwr2, r2); });
			t1.join();
			t2.join();
			for (size_t k = 0; k < NP && !bad; k++)
				bad = r1[k].status != ref.res[k].status || r2[k].status != ref.res[k].status ||
						r1[k].n_samples != ref.res[k].n_samples || r2[k].n_samples != ref.res[k].n_samples;
			if (bad) {
				printf("two concurrent callers: results differ\n");
				return 1;
			}
			lw_batch_destroy(b2);
		}
		{ // entropy stage on the device: the host side (prologues, packet pool, planning) must decide the same statuses, sample
		  // counts and offsets as the host stage -- for eligible streams nothing after the prologue can fail
			lw_batch *b3 = lw_batch_create(dec, NP, LW_FMT_I16_PLANAR, &err);
			const int rc = lw_batch_set_entropy_on_device(b3, 1);
			if (rc == LW_OK) {
				for (auto *p : pwr)
					lw_pwr_reset(p);
				if (lw_batch_entropy(b3, pk.data(), NP, 2) || lw_batch_upload(b3, nullptr)) {
					printf("device-entropy batch failed on the host side\n");
					return 1;
				}
				const lw_packet_result *r3 = lw_batch_results(b3);
				for (size_t k = 0; k < NP; k++)
					if (r3[k].status != ref.res[k].status || r3[k].n_samples != ref.res[k].n_samples ||
							r3[k].out_offset != ref.res[k].out_offset) {
						printf("packet %zu: device-entropy mode plans differently (%d/%u vs %d/%u)\n", k, r3[k].status, r3[k].n_samples,
								ref.res[k].status, ref.res[k].n_samples);
						return 1;
					}
				printf("device-entropy mode: host side agrees\n");
			} else if (rc != LW_ERR_UNSUPPORTED) {
				printf("lw_batch_set_entropy_on_device: %d\n", rc);
				return 1;
			}
			lw_batch_destroy(b3);
		}
		printf("check ok: %zu packets (%zu decodable vs single-packet hook)\n", NP, ok);
		return 0;
	}
	// this also wakes the machine's other cores up (VMs hand out idle vCPUs lazily: the first second of a multi-threaded
	// run can be serialised on one core)
	for (auto t0 = std::chrono::steady_clock::now(); std::chrono::steady_clock::now() - t0 < std::chrono::seconds(2);)
		lw_batch_entropy(b, pk.data(), NP, 0);
	for (int nt : threads) {
		double best = 1e9;
		for (int r = 0; r < reps; r++) {
			auto t0 = std::chrono::steady_clock::now();
			int rc = lw_batch_entropy(b, pk.data(), NP, nt);
			double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
			if (rc) {
				printf("rc %d\n", rc);
				return 1;
			}
			best = std::min(best, dt);
		}
		printf("threads %2d: %.3f ms per batch of %zu  -> %.3f M packets/s  (%.2f us per packet-thread)\n", nt, best * 1e3, NP,
				NP / best * 1e-6, best * nt / NP * 1e6);
	}
	return 0;
}
