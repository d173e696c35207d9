// Host side of the specialised long-block (n = 2048) synthesis kernel: eligibility, the LDS table image
// and the per-launch work plan.  Product code.
#pragma once

#include "lw_host.hpp"

#include <cstdint>
#include <vector>

#define LW_FAST_BS 11          // the kernel is specialised for blocksize_1 = 11 (n = 2048)
#define LW_FAST_MAX_FLOORS 2   // distinct floor-1 configurations staged in LDS
#define LW_FAST_WAVES 16       // waves (packet-units) per workgroup and round: 4 per SIMD, <= 128 VGPRs each
#define LW_FAST_MAX_ROUNDS 16  // rounds per workgroup (chunk = rounds * packets per round consecutive items)

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

// The layout is fixed at compile time (LWI_*): the kernel folds these offsets into its ds_read instructions instead
// of holding sixteen of them in scalar registers; build_fast_plan() checks that the image it writes agrees.
enum : uint32_t {
	LWI_APAIR = 0,      // 4096
	LWI_TW_S2 = 4096,   // 2048
	LWI_TW_L0 = 6144,   // 1024
	LWI_TW_L1 = 7168,   // 512
	LWI_TW_L2 = 7680,   // 256
	LWI_TW_L3 = 7936,   // 128
	LWI_TW_L4 = 8064,   // 64
	LWI_A2 = 8128,      // 16
	LWI_C4 = 8144,      // 2048
	LWI_B_LO = 10192,   // 2048
	LWI_B_HI = 12240,   // 2048
	LWI_WIN = 14288,    // 4096
	LWI_INV_DB = 18384, // 1024
	LWI_XSF = 19408,    // LW_FAST_MAX_FLOORS * 256
	LWI_SID16 = 19920,  // LW_FAST_MAX_FLOORS * 2048
	LWI_TOTAL = 24576   // 24016 padded to a multiple of 1024
};

struct LwFastUnit {
	int8_t ch_a, ch_b;  // channels handled by one wave; ch_b = -1 for a single channel
	uint8_t coupled;    // (ch_a = magnitude, ch_b = angle) form a coupling step
	uint8_t floor_a, floor_b; // staged floor slot (0 / 1) of each channel
	uint8_t F_a, F_b;   // floor-1 post count of each channel (<= 64)
	uint8_t slot;       // kernel argument copy only: packet slot of the wave inside a round (0xFF = idle wave)
};
static_assert(sizeof(LwFastUnit) == 8, "LwFastUnit is loaded as one 8-byte scalar");

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

// where the previous packet's un-windowed right half comes from
#define LW_SRC_NONE 0u   // no previous window: 0 samples out (audio.rs:1140-1152)
#define LW_SRC_LDS 1u    // the previous item of the work list, same workgroup (hand-over buffer in LDS)
#define LW_SRC_STATE 2u  // state slot src_arg (parity in flags)
#define LW_SRC_HALO 3u   // halo slot src_arg, filled by the RIGHT_ONLY pre-pass
#define LW_SRC_TD 4u     // time-domain block at float offset src_arg of B.td (generic-kernel predecessor)

#define LW_IF_NEXT_LDS 1u // the next item of the list takes this packet's right half through LDS
#define LW_IF_TDONLY 32u  // = LW_RF_TDONLY: no overlap-add here, the whole time-domain block goes to td

// One work item of the specialised kernel = one packet; everything the kernel needs, in one 32-byte scalar load.
struct LwFastItem {
	uint32_t res_off;   // float offset of the packet's [ch][1024] residue block
	uint32_t floor_off; // u16 offset of its [ch][fstride] floor block
	uint32_t out_off;   // element offset of its output block
	uint32_t src_arg;   // see LW_SRC_*
	int32_t state_out;  // state slot that receives the raw right half, or -1
	uint32_t halo_out;  // RIGHT_ONLY pre-pass: halo slot to fill
	uint8_t src_kind;   // LW_SRC_*
	uint8_t mode;
	uint8_t flags;      // LW_RF_PARITY_IN / LW_RF_PARITY_OUT / LW_RF_WRITE_TD / LW_IF_NEXT_LDS
	uint8_t pad;
	uint32_t pkt;       // index into the batch's records (host bookkeeping)
};
static_assert(sizeof(LwFastItem) == 32, "LwFastItem must stay 32 bytes");

struct LwFastLaunch {
	LwFastImage off;
	const uint8_t *d_image;
	const LwFastItem *d_items;
	uint32_t n_items;
	const LwFastItem *d_halo_items;
	uint32_t n_halo_items;
	uint32_t n_units;
	uint32_t per_round; // packets per workgroup and round
	uint32_t rounds;    // rounds per workgroup
	uint32_t dense;     // item k == packet k with uniform block sizes
	uint32_t late_from; // first wave of a workgroup that issues its first HBM loads late
	uint32_t has_tdonly; // some item is LW_IF_TDONLY
	LwFastUnit units[LW_FAST_WAVES];
	float *d_halo;
};

namespace lw {
// Decide whether the stream shape is covered by the specialised kernel and build its LDS image.
void build_fast_plan(const Ident &id, const Setup &s, LwFastPlan &plan);
}
