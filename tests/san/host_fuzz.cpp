// Sanitizer harness for the host stage (the Rust reference is memory-safe by construction, the C++
// host code is not): built by tests/test_fuzz_host.py with g++ -fsanitize=address,undefined from the PRODUCT sources
// (lw_headers.cpp, lw_entropy.cpp) and fed mutated header and audio packets.  Any out-of-bounds access, overflow or
// uninitialised-index use aborts the process; the test only checks the exit status.
//
// Input file: u32 n_cases, then per case: u32 len_ident, ident, u32 len_setup, setup, u32 n_packets, {u32 len, bytes}...
#include "../../lewton_amd/csrc/lw_entropy.hpp"
#include "../../lewton_amd/csrc/lw_host.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

static bool rd(FILE *f, uint32_t &v) { return fread(&v, 4, 1, f) == 1; }
static bool rdv(FILE *f, std::vector<uint8_t> &b)
{
	uint32_t n;
	if (!rd(f, n) || n > (64u << 20))
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
	uint32_t n_cases = 0;
	if (!rd(f, n_cases))
		return 2;
	size_t parsed = 0, decoded = 0;
	for (uint32_t c = 0; c < n_cases; c++) {
		std::vector<uint8_t> idp, stp;
		uint32_t n_pk = 0;
		if (!rdv(f, idp) || !rdv(f, stp) || !rd(f, n_pk))
			return 2;
		int err = 0;
		auto id = lw::read_header_ident(idp.data(), idp.size(), err);
		std::unique_ptr<lw::Setup> st;
		if (id)
			st = lw::read_header_setup(stp.data(), stp.size(), id->channels, id->bs0, id->bs1, err);
		if (st)
			parsed++;
		for (uint32_t k = 0; k < n_pk; k++) {
			std::vector<uint8_t> pk;
			if (!rdv(f, pk))
				return 2;
			if (!st)
				continue;
			const size_t ch = id->channels, half = ((size_t)1 << id->bs1) / 2;
			unsigned fstride = 2;
			bool any0 = false;
			for (const auto &fl : st->floors) {
				if (fl.type == 1)
					fstride = std::max<unsigned>(fstride, (unsigned)fl.f1.x_list.size());
				else
					any0 = true;
			}
			fstride = (fstride + 1) & ~1u;
			// exact-size heap buffers: the sanitizer sees every byte written past them
			std::vector<uint16_t> floor(ch * fstride);
			std::vector<float> res(ch * half), curve(any0 ? ch * half : 0);
			lw::EntropyScratch scr;
			lw::Prologue p;
			size_t cnt = 0;
			(void)lw::decoded_sample_count(*id, *st, pk.data(), pk.size(), cnt);
			int rc = lw::entropy_decode(*id, *st, pk.data(), pk.size(), p, floor.data(), fstride, res.data(), scr, nullptr,
					any0 ? curve.data() : nullptr);
			if (rc == 0)
				decoded++;
		}
	}
	fclose(f);
	printf("cases %u, setups parsed %zu, packets decoded %zu\n", n_cases, parsed, decoded);
	return 0;
}
