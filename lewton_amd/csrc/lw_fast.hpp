// Host side of the specialised long-block (n = 2048) synthesis kernel: eligibility, the LDS table image
// and the per-launch work plan.  Product code.
#pragma once

#include "lw_host.hpp"

#include <cstdint>
#include <vector>

#define LW_FAST_BS 11          // the kernel is specialised for blocksize_1 = 11 (n = 2048)
#define LW_FAST_MAX_FLOORS 2   // distinct floor-1 configurations staged in LDS
#define LW_FAST_WAVES 16       // waves (packet-units) per workgroup

// Byte offsets inside the LDS image (all 16-byte aligned).  Index conventions: lane = 0..63,
// p = pair index (u[2p], u[2p+1]) of the n/2-point butterfly array, m' = 2*lane + c.
struct LwFastImage {
	uint32_t apair;  // float2[512]        A as pairs: step 1 (imdct.rs:337-371) reads [m] and [511-m]
	uint32_t tw_s2;  // float2[4][64]      step 2 twiddle of lower pair p = 64x+lane: A[n/2-4-4p ..]
	uint32_t tw_l0;  // float2[2][64]      stage l=0: A[8r ..],  r = 127 - 64b - lane
	uint32_t tw_l1;  // float2[64]         stage l=1: A[16r ..], r = 63 - lane
	uint32_t tw_l2;  // float2[4][8]       stage l=2: A[32r ..], r = 31 - (8*yy + lo3)
	uint32_t tw_l3;  // float2[2][8]       stage l=3: A[64r ..], r = 15 - (8*b + lo3)
	uint32_t tw_l4;  // float2[8]          stage l=4: A[128r ..], r = 7 - lo3
	uint32_t a2;     // float              A[n/8] (imdct.rs:237-238)
	uint32_t c4;     // float4[2][64]      C[4m' .. 4m'+3]
	uint32_t b_lo;   // float4[2][64]      B[4m' .. 4m'+3]
	uint32_t b_hi;   // float4[2][64]      B[4(255-m') .. +3]
	uint32_t win;    // float[2][64][8]    window slope pairs (s[q], s[n/2-1-q]) for q = 511-2m', 510-2m', 1+2m', 2m'
	uint32_t inv_db; // float[256]
	uint32_t xsf;    // float[LW_FAST_MAX_FLOORS][64]   ascending post x of each staged floor (padded with +inf)
	uint32_t sid16;  // u16[LW_FAST_MAX_FLOORS][4][64][4] 16 * (static interval index) of bin 4(64x+lane)+j
	uint32_t total;  // bytes
};

struct LwFastUnit {
	int8_t ch_a, ch_b;  // channels handled by one wave; ch_b = -1 for a single channel
	uint8_t coupled;    // (ch_a = magnitude, ch_b = angle) form a coupling step
	uint8_t floor_a, floor_b; // staged floor slot (0 / 1) of each channel
};

struct LwFastPlan {
	bool eligible = false;
	const char *why_not = "";
	std::vector<uint8_t> image;
	LwFastImage off{};
	uint8_t long_mode_mask[32] = {0};        // bit m set: mode m is a long mode covered by the plan
	std::vector<LwFastUnit> units;           // same for every covered mode
	uint32_t n_staged_floors = 0;
	uint8_t staged_floor_F[LW_FAST_MAX_FLOORS] = {0};
};

struct LwFastItem {
	uint32_t pkt;  // index into the batch's records
	uint32_t halo; // halo slot that holds (main pass) / receives (pre-pass) the predecessor's right half, or 0xFFFFFFFF
};

struct LwFastLaunch {
	LwFastImage off;
	const uint8_t *d_image;
	const LwFastItem *d_items;
	uint32_t n_items;
	const LwFastItem *d_halo_items;
	uint32_t n_halo_items;
	uint32_t n_units;
	LwFastUnit units[LW_FAST_WAVES];
	float *d_halo;
};

namespace lw {
// Decide whether the stream shape is covered by the specialised kernel and build its LDS image.
void build_fast_plan(const Ident &id, const Setup &s, LwFastPlan &plan);
}
