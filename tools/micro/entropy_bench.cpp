// Single-thread throughput of the host entropy stage (lw::entropy_decode) on a packet file written by
// tools/entropy_bench.py (the format of tests/san/host_fuzz.cpp).  No GPU, no library: built from the product sources.
//   usage: entropy_bench case.bin [reps]
#include "../../lewton_amd/csrc/lw_entropy.hpp"
#include <cstdlib>
#include <chrono>
#include <cstdio>
static bool rd(FILE *f, uint32_t &v) { return fread(&v, 4, 1, f) == 1; }
static bool rdv(FILE *f, std::vector<uint8_t> &b) { uint32_t n; if (!rd(f, n)) return false; b.resize(n); return n == 0 || fread(b.data(), 1, n, f) == n; }
int main(int argc, char **argv)
{
	FILE *f = fopen(argv[1], "rb");
	int reps = argc > 2 ? atoi(argv[2]) : 20;
		uint32_t nc, npk; rd(f, nc);
	std::vector<uint8_t> idp, stp; rdv(f, idp); rdv(f, stp); rd(f, npk);
	std::vector<std::vector<uint8_t>> pk(npk);
	for (auto &p : pk) rdv(f, p);
	int err = 0;
	auto id = lw::read_header_ident(idp.data(), idp.size(), err);
	auto st = lw::read_header_setup(stp.data(), stp.size(), id->channels, id->bs0, id->bs1, err);
	unsigned fstride = 2;
	for (const auto &fl : st->floors) if (fl.type == 1) fstride = std::max<unsigned>(fstride, (unsigned)fl.f1.x_list.size());
	size_t ch = id->channels, half = ((size_t)1 << id->bs1) / 2;
	std::vector<uint16_t> fo(ch * fstride); std::vector<float> res(ch * half);
	lw::EntropyScratch scr;
	double best = 1e9; uint64_t bits = 0;
	for (int r = 0; r < reps; r++) {
		auto t0 = std::chrono::steady_clock::now();
		for (auto &p : pk) {
			lw::Prologue pr; uint64_t b = 0;
			int rc = lw::entropy_decode(*id, *st, p.data(), p.size(), pr, fo.data(), fstride, res.data(), scr, &b, nullptr);
			if (rc) { printf("rc %d\n", rc); return 1; }
			bits += b;
		}
		double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		best = std::min(best, dt);
	}
	printf("%.2f us/packet  (%.0f packets/s/thread) bits/pk %.0f\n", best / npk * 1e6, npk / best, (double)bits / reps / npk);
}
