// Host logic of the stream layer (lw_ogg.cpp: OggStreamReader semantics of inside_ogg.rs:66-313 -- header bootstrap,
// chained streams, final-granule truncation, skip_samples_linear, seek_absgp_pg, the look-ahead queue) WITHOUT a GPU: the
// product sources linked against hip_standins.inc, so every sample VALUE is zero but every count, status, serial and
// granule position is what the product decides on the host.  tests/test_host_ogg.py compares the trace printed here
// with the oracle's OggStreamReader (oracle/pyogg.py) and runs mutated files through it under ASan/UBSan.
//   usage: ogg_stream_host file.ogg seq | ahead K | skip N | seek G | mix K singles skip goal | hop N
//   environment: LW_OSH_DEVICE_ENTROPY=1, LW_OSH_READ_AHEAD=K (lw_ogg_stream_set_read_ahead)
// trace lines:  P <n_samples> <status> <serial> <link> <absgp|->      (one per decoded packet)
//               E <code>                                               (terminating error), "EOF" at a clean end
#include "../../include/lewton_amd.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "hip_standins.inc"

static void show(lw_ogg_stream *s, size_t n, int status)
{
	uint64_t gp = 0;
	const int has = lw_ogg_stream_last_absgp(s, &gp);
	if (has)
		printf("P %zu %d %u %u %llu\n", n, status, lw_ogg_stream_serial(s), lw_ogg_stream_link_index(s), (unsigned long long)gp);
	else
		printf("P %zu %d %u %u -\n", n, status, lw_ogg_stream_serial(s), lw_ogg_stream_link_index(s));
}

static size_t cap_for(lw_ogg_stream *s)
{
	lw_ident_info info;
	lw_ident_get_info(lw_ogg_stream_ident(s), &info);
	return (size_t)info.audio_channels << info.blocksize_1;
}

int main(int argc, char **argv)
{
	if (argc < 3)
		return 2;
	FILE *f = fopen(argv[1], "rb");
	if (!f)
		return 2;
	std::vector<uint8_t> data;
	uint8_t buf[65536];
	for (size_t k; (k = fread(buf, 1, sizeof buf, f)) > 0;)
		data.insert(data.end(), buf, buf + k);
	fclose(f);
	const std::string mode = argv[2];
	const unsigned long long arg = argc > 3 ? strtoull(argv[3], nullptr, 10) : 0;
	int err = 0;
	lw_ogg_stream *s = lw_ogg_stream_open(lw_ogg_reader_open_memory(data.data(), data.size(), 0), 0, &err);
	if (!s) {
		printf("E %d\n", err);
		return 0;
	}
	if (getenv("LW_OSH_DEVICE_ENTROPY")) // look-ahead batches in device-entropy mode: the host side copies packets, plans, rolls back
		lw_ogg_stream_set_entropy_on_device(s, 1);
	if (getenv("LW_OSH_READ_AHEAD")) // the packet-by-packet calls served from batches decoded ahead: every trace must stay what it is
		lw_ogg_stream_set_read_ahead(s, (size_t)atoi(getenv("LW_OSH_READ_AHEAD")), 2);
	std::vector<int16_t> out(cap_for(s));
	auto drain = [&]() {
		for (;;) {
			size_t n = 0;
			const int rc = lw_ogg_stream_read_dec_packet(s, LW_FMT_I16_PLANAR, out.data(), out.size(), &n);
			if (rc == LW_ERR_CAPACITY) { // chain boundary: the new link needs a bigger block
				out.resize(cap_for(s));
				continue;
			}
			if (rc == LW_OGG_EOF) {
				printf("EOF\n");
				return;
			}
			if (rc >= LW_AUDIO_END_OF_PACKET && rc <= LW_AUDIO_BUFFER_NOT_ADDRESSABLE && getenv("LW_OSH_GO_ON")) {
				show(s, 0, rc); // BadAudio(code): the packet is consumed, the caller may go on (LW_OSH_GO_ON: the trace does)
				continue;
			}
			if (rc != LW_OK) {
				printf("E %d\n", rc);
				return;
			}
			show(s, n, 0);
		}
	};
	if (mode == "seq") {
		drain();
	} else if (mode == "ahead") {
		const size_t K = (size_t)std::max<unsigned long long>(1, arg);
		const int n_threads = getenv("LW_OSH_THREADS") ? atoi(getenv("LW_OSH_THREADS")) : 2; // (profiling runs: more entropy threads)
		std::vector<uint32_t> ns(K);
		std::vector<int32_t> st(K);
		for (;;) {
			size_t np = 0;
			out.resize(std::max(out.size(), cap_for(s) * K));
			const int rc = lw_ogg_stream_read_dec_packets(s, LW_FMT_I16_PLANAR, K, n_threads, out.data(), out.size(), ns.data(), st.data(), &np);
			if (rc == LW_OGG_EOF) {
				printf("EOF\n");
				break;
			}
			if (rc != LW_OK) {
				printf("E %d\n", rc);
				break;
			}
			printf("B %zu\n", np);
			for (size_t i = 0; i < np; i++)
				printf("Q %u %d\n", ns[i], st[i]);
			if (np == 0) { // in front of a chain boundary: cross it with the single-packet call
				size_t n = 0;
				int r1;
				while ((r1 = lw_ogg_stream_read_dec_packet(s, LW_FMT_I16_PLANAR, out.data(), out.size(), &n)) == LW_ERR_CAPACITY)
					out.resize(cap_for(s) * K);
				if (r1 == LW_OGG_EOF) {
					printf("EOF\n");
					break;
				}
				if (r1 != LW_OK) {
					printf("E %d\n", r1);
					break;
				}
				show(s, n, 0);
			} else {
				show(s, 0, 0); // position after the batch
			}
		}
	} else if (mode == "skip") {
		size_t left = (size_t)arg, n = 0;
		int got = 0, rc;
		while ((rc = lw_ogg_stream_skip_samples_linear(s, left, LW_FMT_I16_PLANAR, out.data(), out.size(), &n, &left, &got)) ==
				LW_ERR_CAPACITY)
			out.resize(cap_for(s));
		if (rc != LW_OK) {
			printf("E %d\n", rc);
		} else {
			printf("S %d %zu %zu\n", got, got ? n : 0, left);
			if (got)
				show(s, n, 0);
			drain();
		}
	} else if (mode == "mix") {
		// the roll-back paths of the look-ahead pipeline: one batched call of K packets, then `singles` packet-by-packet
		// calls, a batched call, a skip of `skip` samples, a batched call, optionally a seek, then drain packet by packet.
		//   argv: K singles skip seek_goal(or -1)
		const size_t K = (size_t)std::max<unsigned long long>(1, arg);
		const size_t singles = argc > 4 ? strtoull(argv[4], nullptr, 10) : 2;
		const size_t skip = argc > 5 ? strtoull(argv[5], nullptr, 10) : 0;
		const long long goal = argc > 6 ? atoll(argv[6]) : -1;
		std::vector<uint32_t> ns(K);
		std::vector<int32_t> st(K);
		bool stop = false;
		auto batch = [&]() {
			size_t np = 0;
			out.resize(std::max(out.size(), cap_for(s) * K));
			const int rc = lw_ogg_stream_read_dec_packets(s, LW_FMT_I16_PLANAR, K, 2, out.data(), out.size(), ns.data(), st.data(), &np);
			if (rc == LW_OGG_EOF) {
				printf("EOF\n");
				stop = true;
				return;
			}
			if (rc != LW_OK) {
				printf("E %d\n", rc);
				stop = true;
				return;
			}
			for (size_t i = 0; i < np; i++)
				printf("Q %u %d\n", ns[i], st[i]);
			show(s, 0, 0);
		};
		batch();
		for (size_t i = 0; i < singles && !stop; i++) {
			size_t n = 0;
			int r1;
			while ((r1 = lw_ogg_stream_read_dec_packet(s, LW_FMT_I16_PLANAR, out.data(), out.size(), &n)) == LW_ERR_CAPACITY)
				out.resize(cap_for(s) * K);
			if (r1 == LW_OGG_EOF) {
				printf("EOF\n");
				stop = true;
			} else if (r1 != LW_OK) {
				printf("E %d\n", r1);
				stop = true;
			} else {
				show(s, n, 0);
			}
		}
		if (!stop)
			batch();
		if (!stop && skip) {
			size_t left = skip, n = 0;
			int got = 0, rc;
			while ((rc = lw_ogg_stream_skip_samples_linear(s, left, LW_FMT_I16_PLANAR, out.data(), out.size(), &n, &left, &got)) ==
					LW_ERR_CAPACITY)
				out.resize(cap_for(s) * K);
			if (rc != LW_OK) {
				printf("E %d\n", rc);
				stop = true;
			} else {
				printf("S %d %zu %zu\n", got, got ? n : 0, left);
				if (got)
					show(s, n, 0);
			}
		}
		if (!stop)
			batch();
		if (!stop && goal >= 0) {
			const int rc = lw_ogg_stream_seek_absgp_pg(s, (uint64_t)goal);
			printf("K %d\n", rc);
			if (rc == LW_OK)
				batch();
		}
		if (!stop)
			drain();
	} else if (mode == "hop") {
		// `arg` packet-by-packet calls, a skip of argv[4] samples, argv[5] more calls, a seek to argv[6] (or -1), then drain: with the
		// read-ahead on, every one of these lands in the middle of a served batch
		const size_t skip = argc > 4 ? strtoull(argv[4], nullptr, 10) : 0;
		const size_t more = argc > 5 ? strtoull(argv[5], nullptr, 10) : 0;
		const long long goal = argc > 6 ? atoll(argv[6]) : -1;
		bool stop = false;
		auto singles = [&](size_t cnt) {
			for (size_t i = 0; i < cnt && !stop; i++) {
				size_t n = 0;
				int r1;
				while ((r1 = lw_ogg_stream_read_dec_packet(s, LW_FMT_I16_PLANAR, out.data(), out.size(), &n)) == LW_ERR_CAPACITY)
					out.resize(cap_for(s));
				if (r1 == LW_OGG_EOF) {
					printf("EOF\n");
					stop = true;
				} else if (r1 != LW_OK) {
					printf("E %d\n", r1);
					stop = true;
				} else {
					show(s, n, 0);
				}
			}
		};
		singles((size_t)arg);
		if (!stop && skip) {
			size_t left = skip, n = 0;
			int got = 0, rc;
			while ((rc = lw_ogg_stream_skip_samples_linear(s, left, LW_FMT_I16_PLANAR, out.data(), out.size(), &n, &left, &got)) ==
					LW_ERR_CAPACITY)
				out.resize(cap_for(s));
			if (rc != LW_OK) {
				printf("E %d\n", rc);
				stop = true;
			} else {
				printf("S %d %zu %zu\n", got, got ? n : 0, left);
				if (got)
					show(s, n, 0);
			}
		}
		singles(more);
		if (!stop && goal >= 0) {
			const int rc = lw_ogg_stream_seek_absgp_pg(s, (uint64_t)goal);
			printf("K %d\n", rc);
			stop = rc != LW_OK;
		}
		if (!stop)
			drain();
	} else if (mode == "seek") {
		const int rc = lw_ogg_stream_seek_absgp_pg(s, arg);
		printf("K %d\n", rc);
		if (rc == LW_OK)
			drain();
	}
	lw_ogg_stream_close(s);
	return 0;
}
