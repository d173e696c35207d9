// The one-process multi-device path (lw_sharder_*, csrc/lw_shard.cpp) without a GPU: the PRODUCT sources over stand-ins for
// the HIP runtime (hip_standins.inc: device memory = calloc, kernels = no-ops -- sample values are zero, everything the host
// decides is real), built with ThreadSanitizer.  G logical shards with three calls in flight, both collect forms in turn,
// against the same packets through a one-shard sharder run call by call: statuses, sample counts and block offsets of every
// packet must agree (the offsets after mapping the shard-by-shard layout), and a stream's packets must stay on one shard.
// Test harness (tests/test_host_shard.py).
//   usage: shard_host case.bin shards streams per_call calls [device_entropy]
//   case.bin: u32 n_cases(ignored) | u32 len ident | u32 len setup | u32 n_packets | (u32 len, bytes)*
#include "../../include/lewton_amd.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hip_standins.inc"

static bool rd(FILE *f, uint32_t &v) { return fread(&v, 4, 1, f) == 1; }
static bool rdv(FILE *f, std::vector<uint8_t> &v)
{
	uint32_t n;
	if (!rd(f, n))
		return false;
	v.resize(n);
	return n == 0 || fread(v.data(), 1, n, f) == n;
}

struct Side {
	lw_sharder *sh = nullptr;
	std::vector<lw_shard_stream *> st;
};

#include <atomic>
extern std::atomic<int> lw_standin_fail_launch; // hip_standins.inc
#define CHECK(c, ...)                  \
	do {                               \
		if (!(c)) {                    \
			printf(__VA_ARGS__);       \
			printf("\n");              \
			return 1;                  \
		}                              \
	} while (0)

int main(int argc, char **argv)
{
	if (argc < 6)
		return 2;
	FILE *f = fopen(argv[1], "rb");
	if (!f)
		return 2;
	const size_t G = (size_t)atoi(argv[2]), S = (size_t)atoi(argv[3]), per = (size_t)atoi(argv[4]), calls = (size_t)atoi(argv[5]);
	const bool dev_entropy = argc > 6 && atoi(argv[6]) != 0;
	const bool fail_mode = argc > 7 && atoi(argv[7]) != 0;
	uint32_t nc, npk;
	std::vector<uint8_t> idp, stp;
	if (!rd(f, nc) || !rdv(f, idp) || !rdv(f, stp) || !rd(f, npk))
		return 2;
	std::vector<std::vector<uint8_t>> pool(npk);
	for (auto &p : pool)
		if (!rdv(f, p))
			return 2;
	fclose(f);
	int err = 0;
	lw_ident *id = lw_read_header_ident(idp.data(), idp.size(), &err);
	CHECK(id, "ident header: %d", err);
	lw_ident_info info;
	lw_ident_get_info(id, &info);
	lw_setup *setup = lw_read_header_setup(stp.data(), stp.size(), info.audio_channels, info.blocksize_0, info.blocksize_1, &err);
	CHECK(setup, "setup header: %d", err);
	const size_t n_call = S * per, ch = info.audio_channels, n1 = (size_t)1 << info.blocksize_1;
	std::vector<int> devs(G, 0), one(1, 0);
	{ // with the measurement hook every logical shard gets its own CUs: [32 j / G, 32 (j + 1) / G) of each of the 8 XCDs.  FIRST in
	  // this process: once a ring of an unshared tenant has made the device's copier stream, CU shares are refused (lewton_amd.h)
		lw_debug_sharder_share_cus(1);
		lw_sharder *C = lw_sharder_create(id, setup, devs.data(), G, n_call, LW_FMT_I16_PLANAR, &err);
		lw_debug_sharder_share_cus(0);
		CHECK(C, "sharder(G, CU shares): %d", err);
		int sum = 0;
		for (size_t g = 0; g < G; g++) {
			const int want = G == 1 || G > 32 ? 256 : 8 * (int)((32 * (g + 1) + G - 1) / G - (32 * g + G - 1) / G);
			CHECK(lw_sharder_shard_cus(C, g) == want, "shard %zu of %zu plans for %d CUs, expected %d", g, G,
			      lw_sharder_shard_cus(C, g), want);
			sum += want;
		}
		CHECK(G > 32 || sum == 256, "CU shares add up to %d", sum);
		lw_sharder_destroy(C);
	}
	Side A, B; // A: G shards, pipelined; B: one shard, call by call
	A.sh = lw_sharder_create(id, setup, devs.data(), G, n_call, LW_FMT_I16_PLANAR, &err);
	CHECK(A.sh, "sharder(G): %d", err);
	B.sh = lw_sharder_create(id, setup, one.data(), 1, n_call, LW_FMT_I16_PLANAR, &err);
	CHECK(B.sh, "sharder(1): %d", err);
	// logical shards share the device's CUs unless the measurement hook gives each its own
	for (size_t g = 0; g < G; g++)
		CHECK(lw_sharder_shard_cus(A.sh, g) == 256, "shard %zu of %zu plans for %d CUs", g, G, lw_sharder_shard_cus(A.sh, g));
	CHECK(lw_sharder_shard_cus(B.sh, 0) == 256 && lw_sharder_shard_cus(B.sh, 1) == 0, "CUs of the lone shard");
	if (dev_entropy) {
		CHECK(lw_sharder_set_entropy_on_device(A.sh, 1) == LW_OK && lw_sharder_set_entropy_on_device(B.sh, 1) == LW_OK,
		      "stream not eligible for the device entropy stage");
	}
	for (size_t s = 0; s < S; s++) {
		A.st.push_back(lw_sharder_stream_open(A.sh, 1000 + 7 * s));
		B.st.push_back(lw_sharder_stream_open(B.sh, 1000 + 7 * s));
		CHECK(A.st.back() && B.st.back(), "stream_open");
		CHECK(lw_sharder_shard_of(A.sh, 1000 + 7 * s) == (1000 + 7 * s) % G, "shard_of");
	}
	// every stream walks through the pool from its own start (the pool holds damaged packets too)
	auto packet_of = [&](size_t call, size_t s, size_t k) -> const std::vector<uint8_t> & { return pool[(s * 5 + call * per + k) % pool.size()]; };
	const size_t cap = n_call * ch * n1;
	std::vector<int16_t> outA(cap), outB(cap);
	std::vector<std::vector<lw_packet_result>> resA(calls, std::vector<lw_packet_result>(n_call)), resB(calls, std::vector<lw_packet_result>(n_call));
	std::vector<size_t> elemsA(calls), elemsB(calls);
	std::vector<lw_shard_packet> pk(n_call);
	if (fail_mode) {
		// ---- a launch failure on ONE shard while older calls are in flight on it: the shard's ring is started
		// over, the older calls' parts on that shard come back LW_ERR_DEVICE (never another call's data), the other shards'
		// parts are intact, and the shard works again afterwards
		CHECK(G >= 2 && calls >= 4, "fail mode wants >= 2 shards and >= 4 calls");
		auto fill = [&](size_t c) {
			for (size_t s = 0; s < S; s++)
				for (size_t k = 0; k < per; k++) {
					const auto &p = packet_of(c, s, k);
					pk[s * per + k] = lw_shard_packet{A.st[s], p.data(), p.size()};
				}
		};
		for (size_t c = 0; c < 2; c++) {
			fill(c);
			CHECK(lw_sharder_submit(A.sh, pk.data(), n_call, 2, &elemsA[c]) == LW_OK, "submit %zu", c);
		}
		lw_standin_fail_launch.store(1); // the next launch of whichever shard gets there first
		fill(2);
		CHECK(lw_sharder_submit(A.sh, pk.data(), n_call, 2, &elemsA[2]) == LW_ERR_DEVICE, "the submit with the failing launch has to say so");
		CHECK(lw_sharder_in_flight(A.sh) == 3, "the failed call stays queued so that the other shards' slots can be freed");
		size_t dead_shard = G;
		for (size_t c = 0; c < 3; c++) {
			const int rc = lw_sharder_collect(A.sh, outA.data(), cap, resA[c].data(), n_call);
			CHECK(rc == LW_ERR_DEVICE, "collect %zu after the failure: %d", c, rc);
			for (size_t i = 0; i < n_call; i++) {
				const size_t g = lw_sharder_shard_of(A.sh, 1000 + 7 * (i / per));
				if (resA[c][i].status == LW_ERR_DEVICE) {
					CHECK(dead_shard == G || dead_shard == g, "packets of two shards failed: %zu and %zu", dead_shard, g);
					dead_shard = g;
				}
			}
			CHECK(dead_shard != G, "no packet of call %zu carries the device error", c);
			for (size_t i = 0; i < n_call; i++) // every packet of the failed shard, and only those
				if (lw_sharder_shard_of(A.sh, 1000 + 7 * (i / per)) == dead_shard)
					CHECK(resA[c][i].status == LW_ERR_DEVICE, "call %zu packet %zu of the failed shard: status %d", c, i, resA[c][i].status);
				else
					CHECK(resA[c][i].status != LW_ERR_DEVICE, "call %zu packet %zu of a healthy shard: device error", c, i);
		}
		CHECK(lw_sharder_in_flight(A.sh) == 0, "every call is consumed by its collect");
		for (size_t c = 3; c < calls; c++) { // the shard works again (the streams' states on it are whatever the failed calls left)
			fill(c);
			CHECK(lw_sharder_submit(A.sh, pk.data(), n_call, 2, &elemsA[c]) == LW_OK, "submit after the failure");
			CHECK(lw_sharder_collect(A.sh, outA.data(), cap, resA[c].data(), n_call) == LW_OK, "collect after the failure");
			for (size_t i = 0; i < n_call; i++)
				CHECK(resA[c][i].status != LW_ERR_DEVICE, "call %zu packet %zu after the restart: device error", c, i);
		}
		for (auto *s : A.st)
			lw_sharder_stream_close(s);
		for (auto *s : B.st)
			lw_sharder_stream_close(s);
		lw_sharder_destroy(A.sh);
		lw_sharder_destroy(B.sh);
		lw_setup_free(setup);
		lw_ident_free(id);
		printf("sharder failure ok: shard %zu of %zu restarted, older calls reported, %zu calls\n", dead_shard, G, calls);
		return 0;
	}
	// ---- B: call by call
	for (size_t c = 0; c < calls; c++) {
		for (size_t s = 0; s < S; s++)
			for (size_t k = 0; k < per; k++) {
				const auto &p = packet_of(c, s, k);
				pk[s * per + k] = lw_shard_packet{B.st[s], p.data(), p.size()};
			}
		int rc = lw_sharder_submit(B.sh, pk.data(), n_call, 2, &elemsB[c]);
		CHECK(rc == LW_OK, "one-shard submit: %d", rc);
		rc = lw_sharder_collect(B.sh, outB.data(), cap, resB[c].data(), n_call);
		CHECK(rc == LW_OK, "one-shard collect: %d", rc);
	}
	// ---- A: three calls in flight, the two collect forms in turn
	size_t submitted = 0, collected = 0;
	std::vector<const void *> pcm(G);
	std::vector<size_t> pel(G);
	auto take = [&]() -> int {
		const size_t c = collected++;
		if (c % 2 == 0) {
			if (elemsA[c] > 0)
				CHECK(lw_sharder_collect(A.sh, outA.data(), elemsA[c] - 1, resA[c].data(), n_call) == LW_ERR_CAPACITY,
				      "a collect into too small a buffer has to fail without consuming the call");
			const int rc = lw_sharder_collect(A.sh, outA.data(), cap, resA[c].data(), n_call);
			CHECK(rc == LW_OK, "collect: %d", rc);
			return 0;
		}
		int rc = lw_sharder_collect_pinned(A.sh, resA[c].data(), n_call, pcm.data(), pel.data());
		CHECK(rc == LW_OK, "collect_pinned: %d", rc);
		CHECK(lw_sharder_collect_pinned(A.sh, resA[c].data(), n_call, pcm.data(), pel.data()) != LW_OK, "a held call cannot be collected twice");
		// offsets are relative to the owning shard's block: make them absolute in shard order for the comparison below
		std::vector<size_t> base(G, 0);
		for (size_t g = 1; g < G; g++)
			base[g] = base[g - 1] + pel[g - 1];
		for (size_t i = 0; i < n_call; i++) {
			const size_t s = i / per, g = lw_sharder_shard_of(A.sh, 1000 + 7 * s);
			CHECK(resA[c][i].status != LW_OK || resA[c][i].out_offset + (size_t)resA[c][i].n_samples * ch <= pel[g], "a block outside its shard's buffer");
			resA[c][i].out_offset += base[g];
		}
		rc = lw_sharder_release(A.sh);
		CHECK(rc == LW_OK, "release: %d", rc);
		return 0;
	};
	for (size_t c = 0; c < calls; c++) {
		if (lw_sharder_in_flight(A.sh) == 3) {
			CHECK(lw_sharder_submit(A.sh, pk.data(), n_call, 2, &elemsA[c]) == LW_ERR_CAPACITY, "a fourth call in flight has to be refused");
			if (take())
				return 1;
		}
		for (size_t s = 0; s < S; s++)
			for (size_t k = 0; k < per; k++) {
				const auto &p = packet_of(c, s, k);
				pk[s * per + k] = lw_shard_packet{A.st[s], p.data(), p.size()};
			}
		const int rc = lw_sharder_submit(A.sh, pk.data(), n_call, 2, &elemsA[c]);
		CHECK(rc == LW_OK, "submit: %d", rc);
		submitted++;
	}
	while (lw_sharder_in_flight(A.sh))
		if (take())
			return 1;
	CHECK(lw_sharder_collect(A.sh, outA.data(), cap, resA[0].data(), n_call) == LW_ERR_CAPACITY, "nothing in flight");
	// ---- the two sides agree packet by packet; blocks are laid out shard by shard, inside a shard in call order
	size_t ok_packets = 0, failed_packets = 0;
	for (size_t c = 0; c < calls; c++) {
		CHECK(elemsA[c] == elemsB[c], "call %zu: %zu elements against %zu", c, elemsA[c], elemsB[c]);
		std::vector<size_t> at(G, 0), base(G, 0);
		for (size_t i = 0; i < n_call; i++)
			if (resB[c][i].status == LW_OK)
				at[lw_sharder_shard_of(A.sh, 1000 + 7 * (i / per))] += (size_t)resB[c][i].n_samples * ch;
		for (size_t g = 1; g < G; g++)
			base[g] = base[g - 1] + at[g - 1];
		std::fill(at.begin(), at.end(), 0);
		for (size_t i = 0; i < n_call; i++) {
			const lw_packet_result &a = resA[c][i], &b = resB[c][i];
			CHECK(a.status == b.status && a.n_samples == b.n_samples, "call %zu packet %zu: status %d / %u samples against %d / %u", c, i,
			      a.status, a.n_samples, b.status, b.n_samples);
			if (a.status != LW_OK) {
				failed_packets++;
				continue;
			}
			ok_packets++;
			const size_t g = lw_sharder_shard_of(A.sh, 1000 + 7 * (i / per));
			CHECK(a.out_offset == base[g] + at[g], "call %zu packet %zu: block at %llu, expected %zu", c, i, (unsigned long long)a.out_offset,
			      base[g] + at[g]);
			at[g] += (size_t)a.n_samples * ch;
		}
	}
	for (auto *s : A.st)
		lw_sharder_stream_close(s);
	for (auto *s : B.st)
		lw_sharder_stream_close(s);
	lw_sharder_destroy(A.sh);
	lw_sharder_destroy(B.sh);
	lw_setup_free(setup);
	lw_ident_free(id);
	printf("sharder ok: %zu shards, %zu calls of %zu packets, %zu decoded + %zu rejected alike\n", G, calls, n_call, ok_packets, failed_packets);
	return 0;
}
