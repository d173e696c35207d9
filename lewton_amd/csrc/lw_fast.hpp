// Host side of the specialised long-block (n = 2048) synthesis kernel: eligibility, the LDS table image
// and the per-launch work plan.  Product code.
#pragma once

#include "lw_host.hpp"
#include "lw_records.h"

#include <cstdint>
#include <vector>

#define LW_FAST_BS 11          // the kernel is specialised for blocksize_1 = 11 (n = 2048)
#define LW_FAST_MAX_FLOORS 2   // distinct floor-1 configurations staged in LDS
#define LW_FLOOR_EXACT_ADX 4096u // longest line (in bins) the kernels' closed form of render_line is proven exact for
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

// unit.coupled
#define LW_UNIT_SPLIT_MAG 2u // half of a coupled pair: this wave finishes ch_a = the magnitude channel; ch_b = its partner, whose raw
#define LW_UNIT_SPLIT_ANG 3u // residues are loaded for the inverse coupling only (3: ch_a is the angle channel)

struct LwFastUnit {
	int8_t ch_a, ch_b;  // channels handled by one wave; ch_b = -1 for a single channel
	uint8_t coupled;    // 1: (ch_a = magnitude, ch_b = angle) form a coupling step; LW_UNIT_SPLIT_*: see above
	uint8_t floor_a, floor_b; // staged floor slot (0 / 1) of each channel
	uint8_t F_a, F_b;   // floor-1 post count of each channel (<= 64)
	uint8_t slot;       // kernel argument copy only: packet slot of the wave inside a round (0xFF = idle wave)
};
static_assert(sizeof(LwFastUnit) == 8, "LwFastUnit is loaded as one 8-byte scalar");

// Canonicalising pre-pass (k_prep, lw_kernels.hip; round 6).  The wave-pipeline and block kernels take a stream shape in which every
// covered mode has the same coupling list of disjoint channel pairs and the same floor per channel, at most LW_FAST_MAX_FLOORS distinct
// floor-1 configurations of a bounded post count.  Every other legal shape (header.rs:985-1058: any coupling list; :1060-1080: modes
// with their own mappings; :771-918: floor 0, 65 posts, many floor configurations) is brought to that form per packet by k_prep,
// which writes a second residue buffer -- ALL coupling steps applied (audio.rs:990-1002), and, for the channels marked
// LW_PREP_PREMUL, already multiplied with their floor curve (audio.rs:1035-1037) -- and a second floor buffer in which those
// channels carry the UNIT floor (two posts at inverse-dB index 255 = 1.0: the kernel's own multiply is then exact), a synthetic
// configuration staged next to the native ones.  The kernels run unchanged on units of uncoupled channel pairs.
// (LW_PREP_*: lw_records.h)
struct LwPrepPlan {
	bool on = false;              // the class's packets go through k_prep
	bool premul = false;          // some channel of some covered mode is LW_PREP_PREMUL: the kernels read the second floor buffer
	int unit_slot = -1;           // staged slot of the unit floor, or -1
	std::vector<uint8_t> action;  // [n_modes][ch] LW_PREP_*; rows of modes outside the class are LW_PREP_NONE
	const char *why = "";         // what made the pre-pass necessary (census)
};

// Coupling steps inside the waves (k_long<..., PRE>; round 6).  A coupling list that is not disjoint pairs -- libvorbis' own 5.1
// mapping: (0,2), (3,4), (0,1), (0,3) -- splits into a PREFIX of disjoint pairs, which are the last steps the decoder applies
// (audio.rs:990-1002 walks the list backwards) and stay the units' own steps, and the steps behind it, which come first.  Those a
// unit evaluates itself for its own channels: it loads up to two more channels of the packet (t0, t1: the other readers of these
// lines are waves of the same workgroup, they meet in L2) and runs up to three steps on the registers r0 = channel a, r1 = channel
// b, t0, t1 -- each step (m, a) -> (m', a') in place, results nobody asked for simply stay unused -- before its own step.  One
// dword per unit: t0 | t1 << 8 (channel, 0xFF none) | op0 << 16 | op1 << 20 | op2 << 24 | number of ops << 28, op = m_reg << 2 | a_reg.
#define LW_PRE_NO_CH 0xFFu
#define LW_PRE_MAX_OPS 3u

struct LwFastPlan {
	bool eligible = false;
	const char *why_not = "";
	LwPrepPlan prep;
	std::vector<uint32_t> pre;               // per unit: its program (see above); empty: the stream needs none
	std::vector<uint8_t> image;
	LwFastImage off{};
	uint8_t long_mode_mask[32] = {0};        // bit m set: mode m is a long mode covered by the plan
	std::vector<LwFastUnit> units;           // same for every covered mode
	std::vector<LwFastUnit> units_split;     // one channel per wave (sparse launches: the launch lasts as long as one wave's chain)
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
#define LW_IF_EDGE_L 2u   // (edge mode: short blocks in k_short) short slope on the left: samples 128.. and the raw left edge from here
#define LW_IF_EDGE_R 4u   // short slope on the right: samples up to 1472, the raw right edge / 128-sample state from here
#define LW_IF_SILENT 8u   // no previous window: the packet yields no samples (audio.rs:1140-1152), also none past a short slope
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
	uint32_t split;      // the units are LW_UNIT_SPLIT_* halves (sparse launch): SPLIT instantiation
	uint32_t edge_mode;  // the stream's short blocks run through k_short: EDGE instantiation, d_edge valid
	float *d_edge;       // [packet][side][ch][edge_n]
	uint32_t edge_n;     // values per raw edge = blocksize_0 / 4 (LW_EDGE_VALUES next to k_long)
	const uint16_t *d_sid12; // k_long12: static interval table of the staged floors (HBM)
	LwFastUnit units[LW_FAST_WAVES];
	float *d_halo;
	uint32_t pre_on;              // k_long<..., PRE>: the units evaluate coupling steps themselves, pre[u] = unit u's program (LwFastPlan::pre)
	uint32_t pre[LW_FAST_WAVES];
};

// ---------------------------------------------------------------------------------------------
// Block kernel k_short<L> (n = 32 L = 256 / 512 / 1024): one wave = 64 / L blocks ("slots") of one unit, L lanes per block,
// the transform in registers with the layouts of k_long cut down to 8 L complex pairs per block (b = log2 L):
//   B': lane = p[b-1:0], reg = p[b+2:b]   step 1, step 2, stages l = 0, 1
//   C': (L >= 16) register bits 1, 0 exchanged with lane bits 4, 3 (L = 16: register bit 0 with lane bit 3): stages l = 2 (, 3)
//   D': register = p[2:0] after the 8 x 8 register <-> lane-bits-2..0 transpose: fused last three stages
//   E': lane l of a block handles m' = 2 l + c (c = 0, 1): bit-reverse gather, step 7, step 8, window / overlap-add
// It serves the short blocks (blocksize_0 = 8, 9, 10) of a stream and, where k_long (n = 2048) does not apply, its long blocks
// (blocksize_1 = 9, 10) whose window slopes are both long.
// ---------------------------------------------------------------------------------------------
#define LW_BLK_MIN_BS 8         // block sizes the kernel is instantiated for: 2^8 .. 2^10
#define LW_BLK_MAX_BS 10
#define LW_BLK_MAX_SLOTS 8      // blocks per wave: 64 / L
// k_big<BS> (lw_kernels_big.hip) takes the same slot descriptors for the long blocks of 4096 / 8192 points: one workgroup of
// L = n / 32 = 128 / 256 threads per task, one slot per pass, no table image
#define LW_BIG_MIN_BS 12
#define LW_BIG_MAX_BS 13
#define LW_BIG_MAX_PASSES 64u   // consecutive blocks of a stream per workgroup (one of them a recomputed predecessor inside a stream)
#define LW_BLK_MAX_POSTS(L) ((L) >= 16 ? 64 : 32) // floor-1 posts per channel of a block: 4 per lane (L = 8, 16), 2 per lane (L = 32)

// byte offsets inside the LDS image of k_short<L> (compile-time layout; build_blk_plan writes the same)
template <int L>
struct LwBlkLayout {
	static constexpr uint32_t APAIR = 0;                  // float2[8L]       A as pairs: step 1 reads [m] and [8L - 1 - m]
	static constexpr uint32_t TW_S2 = APAIR + 64u * L;    // float2[4][L]     step 2 twiddle of lower pair p = L x + l: A[n/2 - 4 - 4p ..]
	static constexpr uint32_t TW_L0 = TW_S2 + 32u * L;    // float2[2][L]     stage l = 0: A[8r ..], r = 2L - 1 - L b - l
	static constexpr uint32_t TW_L1 = TW_L0 + 16u * L;    // float2[L]        stage l = 1: A[16r ..], r = L - 1 - l
	static constexpr uint32_t TW_C2 = TW_L1 + 8u * L;     // float2[2][8]     stage l = 2 (L >= 16): A[32r ..]; L = 32: r = 15 - 8 b - lo3, L = 16: r = 7 - lo3 ([0] only)
	static constexpr uint32_t TW_C3 = TW_C2 + 128u;       // float2[8]        stage l = 3 (L = 32): A[64r ..], r = 7 - lo3
	static constexpr uint32_t A2 = TW_C3 + 64u;           // float            A[n/8]
	static constexpr uint32_t C4 = A2 + 16u;              // float4[2][L]     C[4m' .. 4m'+3]
	static constexpr uint32_t B_LO = C4 + 32u * L;        // float4[2][L]     B[4m' .. 4m'+3]
	static constexpr uint32_t B_HI = B_LO + 32u * L;      // float4[2][L]     B[4(4L - 1 - m') .. +3]
	static constexpr uint32_t WIN = B_HI + 32u * L;       // float[2][L][8]   window slope pairs (s[q], s[16L - 1 - q]) for q = 8L-1-2m', 8L-2-2m', 1+2m', 2m'
	static constexpr uint32_t INV_DB = WIN + 64u * L;     // float[256]
	static constexpr uint32_t XSF = INV_DB + 1024u;       // float[LW_FAST_MAX_FLOORS][64]  ascending post x of each staged floor (padded with +inf)
	static constexpr uint32_t SID16 = XSF + 512u;         // u16[LW_FAST_MAX_FLOORS][4][L][4] 16 * (static interval index) of bin 4(L x + l) + j
	static constexpr uint32_t END = SID16 + 64u * L;
	static constexpr uint32_t TOTAL = (END + 1023u) & ~1023u; // whole 1 KB rows: 64 lanes x 16 bytes per staging step
};

// ---------------------------------------------------------------------------------------------
// k_long12 (lw_long12.inc): k_long's design for blocksize_1 = 12 (n = 4096): one wave per CHANNEL of a packet, 16 complex pairs per
// lane.  After step 1 and step 2 (pair bit 9) the 1024 pairs of a channel are two independent 512-pair problems (p9 = 0 / 1) that go
// through k_long's register layouts B -> C -> D with one more stage in front (l = 0 .. 2 in layout B, l = 3 .. 5 in layout C, the fused
// last three in layout D) and meet again in the bit-reverse gather (8 KB of LDS per wave); a lane then finishes m' = 128 h + 2 lane + c2
// for (h, c2) = k4 = 2 h + c2 in 0 .. 3.  LDS image (byte offsets, all 16-byte aligned; the step-1 twiddles come from the A table in
// HBM / L2 and the static interval indices of the floor from `sid12`, both read once per wave):
// ---------------------------------------------------------------------------------------------
struct LwL12Layout {
	static constexpr uint32_t TW_S2 = 0;        // float2[8][64]   step 2 twiddle of lower pair p = 64 x + lane: A[n/2 - 4 - 4p ..]
	static constexpr uint32_t TW_L0 = 4096;     // float2[4][64]   stage l = 0: A[8 r ..],   r = 255 - (64 y + lane)
	static constexpr uint32_t TW_L1 = 6144;     // float2[2][64]   stage l = 1: A[16 r ..],  r = 127 - (64 b + lane)
	static constexpr uint32_t TW_L2 = 7168;     // float2[64]      stage l = 2: A[32 r ..],  r = 63 - lane
	static constexpr uint32_t TW_L3 = 7680;     // float2[4][8]    stage l = 3: A[64 r ..],  r = 31 - (8 yy + lo3)
	static constexpr uint32_t TW_L4 = 7936;     // float2[2][8]    stage l = 4: A[128 r ..], r = 15 - (8 b + lo3)
	static constexpr uint32_t TW_L5 = 8064;     // float2[8]       stage l = 5: A[256 r ..], r = 7 - lo3
	static constexpr uint32_t A2 = 8128;        // float           A[n/8]
	static constexpr uint32_t C4 = 8144;        // float4[4][64]   C[4m' .. 4m'+3], m' = 128 h + 2 lane + c2 at [2 h + c2][lane]
	static constexpr uint32_t B_LO = 12240;     // float4[4][64]   B[4m' .. 4m'+3]
	static constexpr uint32_t B_HI = 16336;     // float4[4][64]   B[4(511 - m') .. +3]
	static constexpr uint32_t WIN = 20432;      // float[4][64][8] window slope pairs (s[q], s[2047 - q]) for q = 1023-2m', 1022-2m', 1+2m', 2m'
	static constexpr uint32_t INV_DB = 28624;   // float[256]
	static constexpr uint32_t XSF = 29648;      // float[LW_FAST_MAX_FLOORS][64]  ascending post x of each staged floor (padded with +inf)
	static constexpr uint32_t END = 30160;
	static constexpr uint32_t TOTAL = 30208;    // whole 16-byte rows
	// sid12 (HBM): u16[LW_FAST_MAX_FLOORS][8][64][4]  16 * (static interval index) of bin 4 (64 x + lane) + j
	static constexpr uint32_t SID_BYTES = LW_FAST_MAX_FLOORS * 8u * 64u * 4u * 2u;
};

struct LwShortPlan {
	bool eligible = false;
	const char *why_not = "";
	LwPrepPlan prep;
	uint32_t lanes = 0;            // L: lanes per block = 2^(bs - 5)
	uint32_t passes = 1;           // most passes of 64 / L slots a wave may work through (more slots per recomputed predecessor where
	                               // a wave holds few blocks: L = 32 -> 3, L = 16 -> 2); the planner picks per batch
	uint32_t bs = 0;               // log2 of the block size
	std::vector<uint8_t> image;
	uint8_t short_mode_mask[32] = {0}; // modes covered (of the plan's blockflag)
	std::vector<LwFastUnit> units;
	uint32_t n_staged_floors = 0;
	uint8_t staged_floor_F[LW_FAST_MAX_FLOORS] = {0};
	uint32_t fl_of[LW_FAST_MAX_FLOORS] = {0}; // floor index (header order) of each staged floor slot (k_big reads the posts' x from T.floor_x)
	// blocksize 12 (k_long12): its LDS image (LwL12Layout; goes where the other block sizes' image does), the static interval table
	// kept in HBM, and the units with every channel pair split over two waves (LW_UNIT_SPLIT_*)
	std::vector<uint8_t> sid12;
	std::vector<LwFastUnit> units_split;
};

// what a slot of k_short is
#define LW_SS_IDLE 0u      // nothing
#define LW_SS_BLOCK 1u     // a short block: transform, overlap-add with its predecessor's right part, samples out
#define LW_SS_HALO 2u      // a short block whose right part the NEXT slot needs (its own samples are another slot's business)
#define LW_SS_EDGE 3u      // no transform: the stored right part `prev` only feeds the long successor's short-slope overlap
// where a slot's previous right part (64 values pb(0..63) per channel; the other 64 are their mirror image) comes from
#define LW_SP_NONE 0u      // no previous window: no samples (audio.rs:1140-1152)
#define LW_SP_LANE 1u      // the previous slot of the same wave
#define LW_SP_STATE 2u     // state pool: prev_arg = slot, parity in flags
#define LW_SP_EDGE 3u      // edge buffer entry prev_arg (right side) written by k_long for a long block with a short right slope
#define LW_SP_TD 4u        // float offset prev_arg in B.td (channel 0), channel stride prev_stride: a generic-kernel predecessor
#define LW_SF_WRITE_TD 2u     // also store the raw right part into this packet's td block (generic successor)

// One slot = one L-lane group of a k_short wave; 48 bytes (three 16-byte loads by every lane of the group).
struct LwShortSlot {
	uint32_t res_off;     // float offset of the packet's [ch][128] residue block
	uint32_t floor_off;   // u16 offset of its floor block
	uint32_t out_off;     // element offset of its output block
	uint32_t prev_arg;    // see LW_SP_*
	int32_t state_out;    // state slot that receives the raw right part (128 floats per channel), or -1
	uint32_t next_edge;   // packet index of the long successor whose left edge is overlapped here (edge buffer entry), or 0xFFFFFFFF
	uint32_t next_out;    // that successor's out_off
	uint32_t next_m;      // samples per channel that successor yields (576 or 1024): channel stride of its planar output block
	uint16_t prev_stride; // LW_SP_TD: channel stride of the source block
	uint8_t kind;         // LW_SS_*
	uint8_t prev_kind;    // LW_SP_*
	uint32_t flags;       // LW_RF_PARITY_IN / LW_RF_PARITY_OUT (state pool) | LW_SF_*
	uint32_t pkt;         // batch index (host bookkeeping)
	uint32_t pad;
};
static_assert(sizeof(LwShortSlot) == 48, "LwShortSlot is 48 bytes");

// floats per packet in the edge buffer: [side: 0 = left edge pa(448..511), 1 = right edge pb(448..511)][ch][64] next to k_long
// (256-point short blocks).  In general a raw edge holds blocksize_0 / 4 = 8 L values, L = lanes per short block: the top of the
// long block's pa / pb (k_long10 next to 256- / 512-point short blocks: 64 / 128 values)
#define LW_EDGE_VALUES 64u

// k_mix (a mixed short / long batch in ONE launch, lw_kernels_long.hip): the last LW_MIX_SHORT_WAVES waves of every workgroup run
// k_short's work while the first LW_FAST_WAVES - 6 run k_long's (a sparse launch leaves them idle anyway)
#define LW_MIX_SHORT_WAVES 4u
#define LW_MIX_LONG_WAVES (LW_FAST_WAVES - 6u)

struct LwShortLaunch {
	const uint8_t *d_image;
	const LwShortSlot *d_slots; // [n_tasks][passes][64 / lanes]
	uint32_t lanes;             // L
	uint32_t passes;            // passes of 64 / L slots a wave works through
	uint32_t n_tasks;
	uint32_t n_units;
	LwFastUnit units[LW_FAST_WAVES];
	float *d_edge;
	uint32_t fl_of[LW_FAST_MAX_FLOORS]; // k_big: see LwShortPlan
};

static inline uint32_t lw_blk_inv_db_offset(uint32_t lanes)
{
	return lanes == 8 ? LwBlkLayout<8>::INV_DB : lanes == 16 ? LwBlkLayout<16>::INV_DB : lanes == 128 ? LwL12Layout::INV_DB : LwBlkLayout<32>::INV_DB;
}

namespace lw {
// blocksize_0 = blocksize_1 and some mode has the block flag: all modes are planned (and routed, lw_batch.cpp) as the long class
bool lw_unified_classes(const Ident &id, const Setup &s);
// Decide whether the stream shape is covered by the specialised kernel and build its LDS image.
void build_fast_plan(const Ident &id, const Setup &s, LwFastPlan &plan);
// The same for k_short<L>: the blocks of the modes with `blockflag` (0: the short blocks; 1: the long blocks of a stream
// k_long does not cover).
void build_blk_plan(const Ident &id, const Setup &s, bool blockflag, const LwFastPlan &fast, LwShortPlan &plan);
}
