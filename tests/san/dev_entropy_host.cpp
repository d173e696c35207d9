// The DEVICE entropy stage's algorithm (lewton_amd/csrc/lw_dev_entropy.h, the function the HIP kernel runs per lane)
// executed on the host, packet by packet, against the host entropy stage (lw::entropy_decode): floor records and residue
// vectors must be bit-identical on intact, truncated and bit-flipped packets.  Test harness (tests/test_dev_entropy_host.py).
//   usage: dev_entropy_host case.bin [mutations per packet] [seed]
//   case.bin: u32 n_cases(ignored) | u32 len ident | u32 len setup | u32 n_packets | (u32 len, bytes)*
#include "../../lewton_amd/csrc/lw_dev_entropy.hpp"
#include "../../lewton_amd/csrc/lw_entropy.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static bool rd(FILE *f, uint32_t &v) { return fread(&v, 4, 1, f) == 1; }
static bool rdv(FILE *f, std::vector<uint8_t> &v)
{
	uint32_t n;
	if (!rd(f, n))
		return false;
	v.resize(n);
	return n == 0 || fread(v.data(), 1, n, f) == n;
}

int main(int argc, char **argv)
{
	if (argc < 2)
		return 2;
	FILE *f = fopen(argv[1], "rb");
	if (!f)
		return 2;
	const int muts = argc > 2 ? atoi(argv[2]) : 4;
	unsigned x = argc > 3 ? (unsigned)atoi(argv[3]) : 1u;
	uint32_t nc, npk;
	std::vector<uint8_t> idp, stp;
	if (!rd(f, nc) || !rdv(f, idp) || !rdv(f, stp) || !rd(f, npk))
		return 2;
	std::vector<std::vector<uint8_t>> pool(npk);
	for (auto &p : pool)
		if (!rdv(f, p))
			return 2;
	int err = 0;
	auto id = lw::read_header_ident(idp.data(), idp.size(), err);
	if (!id)
		return 3;
	auto st = lw::read_header_setup(stp.data(), stp.size(), id->channels, id->bs0, id->bs1, err);
	if (!st)
		return 3;
	const size_t ch = id->channels;
	unsigned fstride = 0;
	for (const auto &fl : st->floors)
		if (fl.type == 1)
			fstride = std::max<unsigned>(fstride, (unsigned)fl.f1.x_list.size());
	fstride = (fstride + 1) & ~1u;
	lw::DevEntropyImage img;
	const char *why = "";
	if (!lw::dev_entropy_build(*id, *st, fstride, img, &why)) {
		printf("not eligible: %s\n", why);
		return 0;
	}
	const LwEntTables T = lw::dev_entropy_view(img, img.blob.data());
	const size_t n1 = (size_t)1 << id->bs1;
	std::vector<uint16_t> fa(ch * fstride), fb(ch * fstride);
	std::vector<float> ra(ch * n1 / 2), rb(ch * n1 / 2);
	std::vector<uint8_t> ws(T.ws_bytes + 16);
	std::vector<uint32_t> words;
	lw::EntropyScratch scr;
	size_t checked = 0, decoded = 0;
	auto rnd = [&]() {
		x = x * 1664525u + 1013904223u;
		return x >> 8;
	};
	for (size_t k = 0; k < pool.size(); k++) {
		for (int m = 0; m <= muts; m++) {
			std::vector<uint8_t> p = pool[k];
			if (m > 0 && !p.empty()) {
				const unsigned kind = rnd() % 3;
				if (kind == 0) { // cut short
					p.resize(rnd() % p.size());
				} else if (kind == 1) { // flip bits
					for (int q = 0; q < 1 + (int)(rnd() % 4); q++)
						p[rnd() % p.size()] ^= (uint8_t)(1u << (rnd() % 8));
				} else { // cut and flip
					p.resize(1 + rnd() % p.size());
					p[rnd() % p.size()] ^= (uint8_t)(1u << (rnd() % 8));
				}
			}
			lw::Prologue pr;
			std::fill(fa.begin(), fa.end(), 0);
			const int rc = lw::entropy_decode(*id, *st, p.data(), p.size(), pr, fa.data(), fstride, ra.data(), scr);
			lw::BitReader br(p.data(), p.size());
			lw::Prologue pr2;
			const int rcp = lw::read_prologue(*id, *st, br, pr2);
			checked++;
			if (rcp != lw::OK) {
				if (rc != rcp) {
					printf("packet %zu mutation %d: prologue status %d but entropy status %d\n", k, m, rcp, rc);
					return 1;
				}
				continue;
			}
			if (rc != lw::OK) {
				printf("packet %zu mutation %d: eligible stream, prologue ok, host entropy stage says %d\n", k, m, rc);
				return 1;
			}
			decoded++;
			const size_t half = pr.n / 2;
			words.assign((p.size() + 3) / 4 + 4, 0u);
			if (!p.empty())
				std::memcpy(words.data(), p.data(), p.size());
			std::fill(fb.begin(), fb.end(), 0);
			std::fill(rb.begin(), rb.begin() + ch * half, 0.0f);
			uint8_t *w = ws.data() + ((16 - ((uintptr_t)ws.data() & 15)) & 15);
			lw_ent_decode_packet(T, words.data(), (uint32_t)p.size(), (uint32_t)br.pos, pr2.mode, pr2.n, fb.data(), rb.data(), (uint32_t *)w,
					(uint16_t *)(w + LW_ENT_POSTS_BYTES), T.general != 0);
			for (size_t c = 0; c < ch; c++) {
				const uint16_t *a = fa.data() + c * fstride, *b = fb.data() + c * fstride;
				const lw::Mapping &map = st->mappings[st->modes[pr.mode].mapping];
				const size_t F = st->floors[map.submap_floor[map.mux[c]]].f1.x_list.size();
				const size_t cmp = a[0] == LW_FLOOR_UNUSED ? 1 : F;
				if (memcmp(a, b, cmp * 2)) {
					printf("packet %zu mutation %d channel %zu: floor records differ\n", k, m, c);
					return 1;
				}
			}
			if (memcmp(ra.data(), rb.data(), ch * half * sizeof(float))) {
				size_t at = 0;
				while (at < ch * half && !memcmp(&ra[at], &rb[at], 4))
					at++;
				printf("packet %zu mutation %d: residue vectors differ at element %zu (%g vs %g)\n", k, m, at, ra[at], rb[at]);
				return 1;
			}
		}
	}
	printf("device entropy algorithm == host entropy stage: %zu cases, %zu decoded past the prologue\n", checked, decoded);
	return 0;
}
