// Internals shared by the translation units of the runtime (lw_runtime.cpp: headers, device context, PreviousWindowRight;
// lw_batch.cpp: batches -- host entropy stage, work plan, upload, launches; lw_packet.cpp: the drop-in single-packet call).
// Product code, not part of the C ABI.
#pragma once

#include "../../include/lewton_amd.h"

#include "lw_entropy.hpp"
#include "lw_fast.hpp"
#include "lw_host.hpp"
#include "lw_dev_entropy.h"
#include "lw_kernels.hpp"

#include <memory>
#include <mutex>
#include <string>
#include <vector>

// thread-local text of the last HIP error (lw_last_device_error)
bool lw_hip_ok(hipError_t e, const char *what);
void lw_set_device_error(const std::string &msg);
#define HIP_TRY(expr)                        \
	do {                                     \
		if (!lw_hip_ok((expr), #expr))       \
			return LW_ERR_DEVICE;            \
	} while (0)

#define LW_XCDS 8u // accelerator dies of an MI355X, each with 32 CUs and its own L2; CU mask bit i of a queue = CU i / 8 of XCD i % 8

struct lw_ident {
	std::shared_ptr<lw::Ident> p;
};
struct lw_setup {
	std::shared_ptr<lw::Setup> p;
};
struct lw_comment {
	std::unique_ptr<lw::Comment> p;
};

struct lw_decoder {
	std::shared_ptr<lw::Ident> id;
	std::shared_ptr<lw::Setup> setup;
	int device = 0;
	int n_cus = 256;                // compute units this decoder's launches are planned for (its share of the device)
	int n_cus_device = 256;
	bool is_gfx950 = false;         // the device the CU-mask layout of lw_decoder_set_cu_share was measured on
	std::vector<uint32_t> cu_mask;  // lw_decoder_set_cu_share: the mask of the streams made for this decoder (empty = all CUs)
	bool shares_device = false;     // lw_decoder_set_shared_device: other decoders' rings run on this GPU as well
	LwDevTables T{};
	void *d_blob = nullptr; // one allocation holding every table
	bool any_coupling = false;
	bool any_floor0 = false; // some floor is of type 0: batches carry explicit floor curves
	// entropy stage on the device (lw_dev_entropy.h): the flattened setup image in HBM, or why the stream is not eligible
	bool dev_entropy_ok = false;
	std::string dev_entropy_why;
	void *d_ent_blob = nullptr;
	LwEntTables E{};
	std::vector<uint8_t> h_mode_floor, h_floor_F, h_mode_role; // host copies of the device tables of the same names
	std::vector<int8_t> h_mode_partner;
	uint32_t max_posts = 2;
	std::vector<uint64_t> mode_floor_bytes; // per mode: bytes of floor input over all channels (SURVEY 8(d) accounting)
	// PreviousWindowRight pool: [slots][2][ch][n1/2] floats
	std::mutex mu;
	float *d_state = nullptr;
	size_t state_cap = 0;
	std::vector<int> free_slots;
	LwFastPlan fast;               // specialised long-block kernel: eligibility, units, LDS image
	uint8_t *d_fast_image = nullptr;
	LwFastUnit *d_fast_units = nullptr;
	LwShortPlan blkp[2];           // block kernel k_short<L>: [0] the short blocks, [1] the long blocks where k_long does not apply
	uint8_t *d_blk_image[2] = {nullptr, nullptr};
	uint16_t *d_l12_sid = nullptr; // k_long12: the static interval table of its floors (LwL12Layout::SID_BYTES, HBM)
	// canonicalising pre-pass k_prep (LwPrepPlan): which block classes' packets go through it, the merged [n_modes][ch] action table
	bool prep_cls[2] = {false, false};
	bool prep_floors = false;      // some channel is LW_PREP_PREMUL: k_prep also writes the second floor buffer, the kernels read it
	std::vector<uint8_t> h_prep_action, h_prep_mode;
	uint8_t *d_prep_action = nullptr;
	lw_batch *one = nullptr; // internal batch for lw_read_audio_packet
	void *one_out = nullptr; // pinned host output for the single-packet path
	size_t one_out_bytes = 0;
};

hipError_t lw_decoder_stream_create(lw_decoder *d, hipStream_t *s);
// the kinds of tenant stream this process has made on a device (process-wide latch; see "Known hazard" in include/lewton_amd.h)
#define LW_TENANT_COPIER_STREAM 1  // the copier thread's own copy stream (a tenant's ring without a CU share)
#define LW_TENANT_MASKED_STREAMS 2 // CU-masked streams (a ring of a decoder with a CU share)
int lw_tenant_streams(int device);
void lw_tenant_streams_note(int device, int kind);

struct lw_pwr {
	lw_decoder *dec = nullptr;
	int slot = -1;
	bool present = false;
	uint32_t len = 0;   // per-channel length
	uint8_t parity = 0; // which of the two buffers holds the valid state
};

struct lw_batch {
	lw_decoder *dec = nullptr;
	size_t max_packets = 0;
	int fmt = 0;
	uint8_t *h_slab = nullptr, *d_slab = nullptr; // all host->device buffers below are slices of these
	size_t slab_bytes = 0;
	LwPacketRec *h_recs = nullptr;
	uint16_t *h_floor = nullptr;
	float *h_res = nullptr;
	float *h_fcurve = nullptr, *d_fcurve = nullptr; // explicit floor curves (floor 0), layout of the residues
	// packets of the generic kernels, by size class (block size <= / > 2^9): dense launch grids instead of 8192
	// workgroups that mostly find out they have nothing to do
	uint32_t *h_gen = nullptr, *d_gen = nullptr; // [4][max_packets]: small blocks, large blocks, k_ola_generic's packets, k_prep's packets
	uint32_t n_gen_small = 0, n_gen_large = 0, n_gen_ola = 0;
	LwGenTask *h_tasks = nullptr, *d_tasks = nullptr; // [max_packets * ch] tasks of the short-block transform kernel
	LwOlaDesc *h_ola = nullptr, *d_ola = nullptr; // [max_packets] descriptors of k_ola_generic's tasks (order of the third list)
	bool has_tdonly = false; // the specialised kernel's work list contains LW_RF_TDONLY packets
	// k_short: the short blocks of streams it covers, eight slots per task (lw_fast.hpp)
	LwShortSlot *h_slots[2] = {nullptr, nullptr}, *d_slots[2] = {nullptr, nullptr}; // per block class
	size_t max_tasks[2] = {0, 0}, n_tasks[2] = {0, 0}; // (max_tasks: capacity in SLOTS)
	uint32_t blk_passes[2] = {1, 1}; // passes per wave of this batch (lw_fast.hpp)
	bool edge_mode = false;  // short blocks in k_short, long blocks with short slopes in k_long<EDGE>
	float *d_edge = nullptr; // [max_packets][2][ch][64], and behind it the flags of k_mix: [max_packets][2][ch] dwords
	int mix_mode = -1;       // lw_debug_batch_set_mix: -1 = k_mix where it applies, 0 = never (two launches)
	// device error word: one dword of pinned host memory the kernels can write (k_mix: a wave whose producer never signalled);
	// lw_batch_device_status reads it once the launches have completed
	uint32_t *h_err = nullptr, *d_err = nullptr;
	void *last_stream = nullptr; // the HIP stream of the most recent launch (lw_batch_device_status clears the edge flags on it)
	uint32_t mix_break_spin = 0; // lw_debug_batch_break_mix: != 0 = the long blocks' waves of k_mix never signal, give up after this many polls
	std::vector<uint32_t> blk_idx[2], blk_slot[2];
	std::vector<int32_t> succ; // per packet: the next packet of the same stream in this batch, or -1
	// entropy stage on the device: the packets themselves go up (word-aligned, zero padded) with one descriptor each
	bool dev_entropy = false;
	LwEntPacket *h_pk = nullptr, *d_pk = nullptr; // [max_packets]
	uint32_t *h_pool = nullptr, *d_pool = nullptr;
	size_t pool_cap_words = 0, pool_words = 0;
	bool ent_done = false;   // k_entropy has run for the records uploaded last (lw_batch_device_entropy ahead of lw_batch_synth)
	LwPacketRec *d_recs = nullptr;
	uint16_t *d_floor = nullptr;
	float *d_res = nullptr;
	float *d_decoupled = nullptr, *d_td = nullptr, *d_tap = nullptr;
	uint16_t *d_floor_alt = nullptr; // k_prep's floor records (layout of d_floor)
	uint32_t n_prep = 0;             // packets of k_prep: the fourth list of h_gen / d_gen
	void *d_out = nullptr;
	size_t d_out_elems = 0;
	LwFastItem *h_items = nullptr, *d_items = nullptr;           // [max_packets] main pass
	LwFastItem *h_halo_items = nullptr, *d_halo_items = nullptr; // [max_packets] halo pre-pass
	float *d_halo = nullptr;
	size_t halo_cap = 0, n_items = 0, n_halo_items = 0;
	std::vector<uint32_t> fast_idx, fast_slot, fast_order;
	uint32_t fast_per_round = 1, fast_rounds = 1, fast_dense = 0, fast_late_from = 1;
	bool fast_split = false; // the specialised kernel runs one channel per wave (sparse launch)
	// blocksize_1 = 10: the long blocks with two long slopes (block class 1, 32 lanes per block) run through k_long10 -- k_long's
	// work list, units and launch shape on the block kernel's table image -- instead of k_short<32>
	bool use_l10 = false;
	int l10_cls = 1;         // the block class k_long10 serves: 1 = the long blocks; 0 = a stream whose ONLY mode is a 1024-point mode
	                         // without the block flag (blocksize_0 = blocksize_1 = 10: libvorbis at 16 / 22 kHz, lowest quality)
	bool use_l12 = false;    // blocksize_1 = 12: the same for k_long12 (one wave per channel) instead of k_big<12>; l10_mode 0 switches it off too
	int l10_mode = -1;       // lw_debug_batch_set_long10: -1 = k_long10 where it applies (long blocks next to short ones in its EDGE
	                         // form when the short blocks run through k_short), 1 = k_long10 without the EDGE form (those blocks through
	                         // the generic kernels), 0 = never (k_short<32>)
	int forced_rounds = 0; // lw_debug_batch_set_rounds: rounds per workgroup of the specialised kernel (0 = the planner decides)
	int halo_mode = -1;    // lw_debug_batch_set_halo: -1 = predecessors of chunk starts recomputed inside the launch where that pays, 0 = by the pre-pass
	size_t n_inline_halo = 0; // such items in the work list of the batch planned last
	size_t n = 0, res_floats = 0, out_elems = 0;
	uint32_t max_n = 0;
	bool has_generic = false, has_fast = false, force_generic = false;
	std::vector<lw_packet_result> results;
	uint64_t alg_bytes = 0;
	uint64_t state_bytes = 0; // window state crossing HBM at the launch boundary (lw_batch_state_bytes)
	std::string last_kernels;
	std::vector<lw::Prologue> prologues;
	std::vector<int> status;
	std::vector<int32_t> slot_last; // per state slot: last ok packet index in this batch (-1 none)
	std::vector<uint32_t> slot_seen; // per state slot: epoch of the batch that last touched it
	uint32_t epoch = 0;
	std::vector<lw_pwr *> touched;
};


inline size_t lw_elem_size(int fmt)
{
	return fmt == LW_FMT_F32_PLANAR ? 4 : 2;
}
int lw_decoder_set_device(const lw_decoder *d);
// grow the state pool to at least `slots` (caller holds d->mu)
int lw_grow_state(lw_decoder *d, size_t slots);
