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

// Decodes floors and residues of one packet.
//   floor_out   [ch][fstride] u16 records (see lw_records.h)
//   residue_out [ch][n/2] f32, pre-decoupling
//   fcurve_out  [ch][n/2] f32: floor-0 channels get their curve here (audio.rs:109-212; host libm); may be nullptr
//               when the setup has no floor of type 0
// Returns OK or an AudioReadError code (on error the outputs are unspecified).
int entropy_decode(const Ident &id, const Setup &s, const uint8_t *pkt, size_t len, Prologue &p, uint16_t *floor_out,
		unsigned fstride, float *residue_out, EntropyScratch &scr, uint64_t *bits_consumed = nullptr,
		float *fcurve_out = nullptr);

} // namespace lw
