// Host entropy stage of the audio-packet path (product code): everything of
// `read_audio_packet_generic` that is bit-serial (src/audio.rs:921-986) plus the floor-1 amplitude
// unwrapping (audio.rs:391-435, header-only neighbour search hoisted to setup time).
#pragma once

#include "lw_host.hpp"
#include "lw_records.h"

#include <vector>

namespace lw {

struct Prologue {
	uint8_t mode = 0;
	bool blockflag = false, prev_flag = false, next_flag = false;
	uint8_t bs = 0;
	uint32_t n = 0;
};

// audio.rs:921-938.  Returns OK or an AudioReadError code; leaves `r` after the window flags.
int read_prologue(const Ident &id, const Setup &s, BitReader &r, Prologue &p);

// audio.rs:874-909
int decoded_sample_count(const Ident &id, const Setup &s, const uint8_t *pkt, size_t len, size_t &count);

struct EntropyScratch {
	std::vector<uint32_t> cls;
	std::vector<float> interleaved;
	std::vector<float> sub;
};

// "Tier B" residue hand-over: instead of adding the VQ vectors on the host, the entropy stage
// records which vector goes where; the additions (audio.rs:595, :610-612) and the type-2 de-interleave (:748-754) run
// in k_residue_vq.  One 64-bit op per decoded codeword:
//   bits  0-23 start coordinate in the submap's vector space ([sub_ch][n/2], or the interleaved type-2 vector)
//   bits 24-31 codebook number            bits 32-55 codebook entry
//   bits 56-59 submap                     bits 60-62 cascade pass
struct SymbolSink {
	std::vector<uint64_t> ops; // decode order; sorted by pass (stable) with sort_by_pass()
	uint32_t pass_off[9];
	uint32_t submap = 0;
	uint32_t last_pass = 0;
	bool in_order = true; // passes never decreased so far (always true for one submap): no sort needed
	void clear()
	{
		ops.clear();
		submap = 0;
		last_pass = 0;
		in_order = true;
	}
	void push(uint32_t coord, uint32_t book, uint32_t entry, uint32_t pass)
	{
		in_order &= pass >= last_pass;
		last_pass = pass;
		ops.push_back((uint64_t)(coord & 0xffffffu) | ((uint64_t)(book & 0xffu) << 24) | ((uint64_t)(entry & 0xffffffu) << 32) |
				((uint64_t)(submap & 0xfu) << 56) | ((uint64_t)(pass & 7u) << 60));
	}
	void sort_by_pass(std::vector<uint64_t> &tmp);
};

// Streams whose residue books all satisfy `dims | partition_size` can use the sink: then no codeword reaches into
// the next partition and the additions of one cascade pass touch disjoint bins (order-free within a pass).
bool symbols_supported(const Ident &id, const Setup &s, const char **why);

// Decodes floors and residues of one packet.
//   floor_out   [ch][fstride] u16 records (see lw_records.h)
//   residue_out [ch][n/2] f32, pre-decoupling
//   fcurve_out  [ch][n/2] f32: floor-0 channels get their curve here (audio.rs:109-212; host libm); may be nullptr
//               when the setup has no floor of type 0
// Returns OK or an AudioReadError code (on error the outputs are unspecified).
int entropy_decode(const Ident &id, const Setup &s, const uint8_t *pkt, size_t len, Prologue &p, uint16_t *floor_out,
		unsigned fstride, float *residue_out, EntropyScratch &scr, uint64_t *bits_consumed = nullptr,
		float *fcurve_out = nullptr, SymbolSink *sink = nullptr);

} // namespace lw
