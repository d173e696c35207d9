// GPU-stage input records: what the host entropy stage hands to the HIP kernels.
// Plain-old-data shared by host (lw_entropy.cpp, lw_device.hip host code) and device code.
//
// Per batch, in HBM (structure of arrays):
//   recs    [n_packets]                 LwPacketRec, 32 B each
//   floor   [n_packets][ch][fstride]    u16 per floor-1 post in ascending-x order:
//                                       bits 0-7 = final_y * multiplier (<= 255), bit 15 = post is active
//                                       (step2 flag); entry 0 == 0xFFFF marks an unused floor (audio.rs:66-70);
//                                       entry 0 == 0xFFFE: floor 0 -- the curve (audio.rs:160-212, transcendental,
//                                       evaluated on the host) is given explicitly in `fcurve`
//   fcurve  [same layout as residue]    f32 floor curve of floor-0 channels (allocated only for setups with a floor 0)
//   residue [sum over packets ch*n/2]   f32, per packet [ch][n/2], BEFORE inverse coupling
//                                       (the vectors of audio.rs:957-986, type-2 already de-interleaved)
// Algorithmic bytes per packet: ch*(n/2)*4 + ch*F*2 + 16 in, ch*m*2 out.
#pragma once

#include <stdint.h>

#define LW_MAX_POSTS 65       // header.rs:873
#define LW_FLOOR_UNUSED 0xFFFFu
#define LW_FLOOR_EXPLICIT 0xFFFEu
#define LW_POST_ACTIVE 0x8000u

// per (mode, channel) action of the canonicalising pre-pass k_prep (LwPrepPlan, lw_fast.hpp)
#define LW_PREP_NONE 0u    // the packet's mode is not one of a pre-passed block class
#define LW_PREP_COPY 1u    // channel: residue after inverse coupling; its floor record as it is (evaluated by the kernel)
#define LW_PREP_PREMUL 2u  // channel: residue after inverse coupling x floor curve; unit floor record

// rec.flags
#define LW_RF_LONG 1u          // mode blockflag
#define LW_RF_SLOPE_BS1 2u     // overlap window slope comes from blocksize_1 (left_n_use_bs1, audio.rs:1058-1064)
#define LW_RF_SKIP 4u          // packet failed in the entropy stage: no device work, no output
#define LW_RF_FAST 8u          // handled by the specialised long-block kernel
#define LW_RF_WRITE_TD 16u     // fast packet must also store its raw right half into its td block (generic successor)
#define LW_RF_TDONLY 32u       // (with LW_RF_FAST) long block whose window shape or stored state is not the (1,1) / full-half case:
                               // the specialised kernel does floor, decoupling and IMDCT and writes the whole time-domain
                               // block; window / overlap-add / state are done by k_ola_generic

struct LwPacketRec {
	uint32_t res_off;   // float offset of this packet's [ch][n/2] residue block
	uint32_t floor_off; // u16 offset of this packet's [ch][fstride] floor block
	uint32_t out_off;   // element offset of this packet's output block
	int32_t prev;       // >= 0: batch index of the previous packet of the same stream; -1: no state
	                    // (0 samples out); <= -2: state slot -(prev + 2), read buffer parity in `flags` bit 7
	int32_t state_out;  // state slot that receives cur[rs..re) (written to the parity NOT being read), or -1
	uint16_t ls, rs;    // left_win_start, right_win_start (audio.rs:1058-1073)
	uint16_t re, plen;  // right_win_end; per-channel length of the stored right part to overlap (0 = none)
	uint8_t bs;         // log2(n)
	uint8_t mode;       // mode number
	uint8_t flags;      // LW_RF_*; bit 6: state_out parity, bit 7: state-in parity
	uint8_t xflags;     // LW_XF_* (host planning only)
};                      // samples per channel = (prev == -1) ? 0 : rs - ls

#ifdef __cplusplus
static_assert(sizeof(LwPacketRec) == 32, "LwPacketRec must stay 32 bytes");
#endif

// rec.xflags: long blocks of streams whose short blocks run through k_short (values of LW_IF_EDGE_* in lw_fast.hpp)
#define LW_XF_EDGE_L 2u        // short slope on the left
#define LW_XF_EDGE_R 4u        // short slope on the right

#define LW_RF_PARITY_OUT 64u
#define LW_RF_PARITY_IN 128u

// One task of the generic overlap-add kernel, packed by the host's planning pass: everything the workgroup needs in ONE load
// (the list index -> packet record -> predecessor's record chain cost the kernel three dependent round trips before its
// first sample).  32 bytes.
struct LwOlaDesc {
	uint32_t cur_off;   // this packet's time-domain block [ch][n] in B.td (float offset)
	uint32_t prev_off;  // previous right part: float offset of channel 0 in B.td (kind 1) or in the state pool (kind 2)
	uint32_t out_off;   // first output element
	int32_t state_out;  // state slot the raw right part goes to, or -1
	uint16_t n, ls, rs, re, plen, prev_stride;
	uint8_t prev_kind;  // 0: no previous window (no samples), 1: predecessor's block in B.td, 2: state pool
	uint8_t flags;      // LW_RF_SLOPE_BS1 | LW_RF_PARITY_OUT
	uint16_t pad;
};

// One task of the short-block transform kernel (a packet's channel), packed by the host's planning pass: the packet record
// and what the kernel used to look up behind it (floor number and post count of the channel, its coupling partner), so that
// one load replaces the chain list index -> record -> mode tables -> floor tables.  40 bytes.
struct LwGenTask {
	LwPacketRec rec;
	uint8_t c;        // channel
	uint8_t fl;       // its floor (mode_floor)
	uint8_t F;        // posts of that floor (floor_F)
	uint8_t role;     // 1 magnitude / 2 angle of the one coupling step the channel is in (pair_coupling streams), 0 none
	int8_t partner;   // the other channel of that step, or -1
	uint8_t pad[3];
};

