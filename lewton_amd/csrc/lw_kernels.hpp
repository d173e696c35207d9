// Device-side tables and kernel launchers of the audio-packet synthesis path (gfx950).
#pragma once

#include "lw_records.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <type_traits>

// hipFuncSetAttribute is per DEVICE, and launchers are entered from several host threads (one decoder per GPU in one
// process; the sharder's workers; the Ogg staging thread next to the caller): `run(set)` calls
// `set` exactly once per (kernel family, current device) -- under the mutex, so that a second thread on the same device
// cannot launch before the attributes are in place -- and marks the device done only when it succeeded.
struct LwPerDeviceOnce {
	std::mutex mu;
	uint64_t done[4] = {0, 0, 0, 0}; // one bit per device ordinal
	template <class Set> hipError_t run(Set &&set)
	{
		int dev = 0;
		if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 256)
			return set(); // unknown ordinal: set the attributes every time (cheap, idempotent)
		std::lock_guard<std::mutex> g(mu);
		if (done[dev >> 6] & (1ull << (dev & 63)))
			return hipSuccess;
		const hipError_t e = set();
		if (e == hipSuccess)
			done[dev >> 6] |= 1ull << (dev & 63); // (a failure leaves the bit clear: the next launch tries again)
		return e;
	}
};

// Kernel launch that RETURNS the launch status (hipLaunchKernelGGL drops it; hipGetLastError would also pick up an earlier
// hipErrorNotReady of a polled event): the arguments are passed by address, in the kernel's parameter order -- and hipLaunchKernel
// reads sizeof(parameter) bytes behind each address, so every argument must BE the parameter's type (a bool or a size_t handed in
// for an int parameter would be read as garbage without a diagnostic): checked against the kernel's signature at compile time.
template <class... P, class... A>
static inline hipError_t lw_launch_k(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t st, A &...a)
{
	static_assert(sizeof...(P) == sizeof...(A), "lw_launch_k: one argument per kernel parameter");
	static_assert((std::is_same<typename std::remove_cv<A>::type, typename std::remove_cv<P>::type>::value && ...),
			"lw_launch_k: every argument must have exactly its kernel parameter's type");
	void *args[] = {(void *)&a...};
	return hipLaunchKernel((const void *)kernel, grid, block, args, lds, st);
}

// CachedBlocksizeDerived (header_cached.rs:19-110) in HBM; computed on the host, never on the device.
struct LwDevBs {
	const float *A, *B, *C, *window;
	const uint32_t *bitrev;
	uint32_t bs, n;
};

#define LW_XSTRIDE 66 // u16 entries per floor in floor_x (LW_MAX_POSTS + 1)

struct LwDevTables {
	LwDevBs bs[2];
	const float *inv_db;          // FLOOR1_INVERSE_DB_TABLE, audio.rs:437-501
	const uint16_t *floor_x;      // [n_floors][LW_XSTRIDE] ascending x of each floor-1 config (header.rs:887-889)
	const uint8_t *floor_F;       // [n_floors] post count
	const uint8_t *mode_floor;    // [n_modes][ch] floor index of each channel (mapping_mux -> submap floor)
	const uint16_t *couple_off;   // [n_modes + 1] offsets (in steps) into `couple`
	const uint8_t *couple;        // (magnitude, angle) channel pairs in header order (audio.rs:990-1002 walks them reversed)
	const uint8_t *sid;           // fast path: [n_floors][2][n_max/2] static interval index per bin (may be null)
	const int8_t *mode_partner;   // [n_modes][ch] the other channel of the ONE coupling step this channel is in, or -1
	const uint8_t *mode_role;     // [n_modes][ch] 1 = magnitude, 2 = angle of that step
	uint32_t pair_coupling;       // 1: in every mode a channel takes part in at most one coupling step (the two tables are valid)
	uint32_t ch, fstride, n_modes, n_floors;
	uint32_t state_stride;        // floats per (slot, parity): ch * (n1 / 2)
	uint32_t state_chan_stride;   // n1 / 2
};

// generic kernels: blocks up to 2^LW_SMALL_BS points are transformed by ONE wave each (four per workgroup), larger ones by
// a 256-thread workgroup (host: lw_runtime.cpp builds one task list per class)
#define LW_SMALL_BS 9

enum LwOutFmt { LW_OUT_I16_PLANAR = 0, LW_OUT_I16_INTERLEAVED = 1, LW_OUT_F32_PLANAR = 2 };

struct LwBatchDev {
	const LwPacketRec *recs;
	const uint16_t *floors;
	const float *residue;
	const float *fcurve; // explicit floor curves (floor 0), layout of residue; nullptr when the setup has none
	float *decoupled; // scratch [same layout as residue]
	float *td;        // scratch: per packet [ch][n] time-domain blocks at float offset 2 * res_off
	float *state;     // state pool [slots][2][ch][n1/2]
	uint32_t n_packets;
	// packets the generic kernels work on, by block-size class; nullptr = every packet (filters inside the kernels)
	const uint32_t *gen_small, *gen_large;
	uint32_t n_gen_small, n_gen_large;
	const uint32_t *gen_ola; // packets of k_ola_generic: the two lists above plus the LW_RF_TDONLY packets
	uint32_t n_gen_ola;
	const LwOlaDesc *ola;    // one descriptor per entry of gen_ola (null: k_ola_generic reads the records)
	const LwGenTask *gen_tasks; // [n_gen_small * ch] tasks of k_imdct_generic<64> (null: it reads lists and records)
};

// Generic path (any block size 64..8192, any window shape, any channel count / coupling list), two phases so
// that the specialised kernel can run in between (it reads td blocks of generic predecessors and writes td
// right halves for generic successors).
// (the launchers that opt a kernel into a large dynamic LDS segment return the result of that call)
hipError_t lw_launch_generic_imdct(const LwDevTables &T, const LwBatchDev &B, float *tap_spec, hipStream_t st, uint32_t max_n,
		bool any_coupling, bool include_fast);
// Canonicalising pre-pass of the specialised kernels (k_prep, lw_kernels.hip; LwPrepPlan in lw_fast.hpp): for the n_list packets of
// d_list, B.decoupled = residues after every coupling step (x floor curve for the channels d_action marks), d_floors_out (may be
// null) = their floor records with the unit floor for those channels.
hipError_t lw_launch_prep(const LwDevTables &T, const LwBatchDev &B, const uint32_t *d_list, uint32_t n_list, const uint8_t *d_action,
		uint16_t *d_floors_out, hipStream_t st);
// entropy stage on the device (lw_kernels_entropy.hip): floor records and residue vectors of n packets from their raw bytes
struct LwEntTables;
struct LwEntPacket;
hipError_t lw_launch_entropy(const LwEntTables &T, const LwEntPacket *d_pk, const LwPacketRec *d_recs, const uint32_t *d_pool,
		uint16_t *d_floor, float *d_res, uint32_t n, hipStream_t st);
void lw_launch_generic_ola(const LwDevTables &T, const LwBatchDev &B, void *out, int fmt, hipStream_t st, bool include_fast);

// Specialised long-block path (lw_kernels_long.hip): optional halo pre-pass + main pass.
struct LwFastLaunch;
hipError_t lw_launch_long(const LwDevTables &T, const LwBatchDev &B, const LwFastLaunch &L, void *out, int fmt, hipStream_t st);
// The same design for blocksize_1 = 10 (k_long10, lw_long10.inc): L.d_image = the block kernel's image for 32 lanes per block
// (LwBlkLayout<32>), the work list and the halo pre-pass as for k_long with 512-value residue vectors.
hipError_t lw_launch_long10(const LwDevTables &T, const LwBatchDev &B, const LwFastLaunch &L, void *out, int fmt, hipStream_t st);
// ... and for blocksize_1 = 12 (k_long12, lw_long12.inc): one wave per channel (L.units = the split units), L.d_image = LwL12Layout
hipError_t lw_launch_long12(const LwDevTables &T, const LwBatchDev &B, const LwFastLaunch &L, void *out, int fmt, hipStream_t st);
// k_mix10: k_long10<EDGE> and k_short<8 / 16> in one launch (arguments as lw_launch_mix)
struct LwShortLaunch;
bool lw_mix10_applicable(const LwFastLaunch &LL, const LwShortLaunch &LS, int n_cus);
hipError_t lw_launch_mix10(const LwDevTables &T, const LwBatchDev &B, const LwFastLaunch &LL, const LwShortLaunch &LS, uint32_t *d_flags,
		uint32_t *d_err, uint32_t spin, bool drop_flags, void *out, int fmt, hipStream_t st);
// Short blocks of such streams (k_short, same translation unit); runs after lw_launch_long (it reads the edge buffer).
struct LwShortLaunch;
hipError_t lw_launch_short(const LwDevTables &T, const LwBatchDev &B, const LwShortLaunch &L, void *out, int fmt, hipStream_t st);
// Long blocks of 4096 / 8192 points (k_big, lw_kernels_big.hip): the same slot descriptors, L.lanes = 128 / 256
hipError_t lw_launch_big(const LwDevTables &T, const LwBatchDev &B, const LwShortLaunch &L, void *out, int fmt, hipStream_t st);
// Both in ONE launch (k_mix, same translation unit) where lw_mix_applicable says so; d_flags: [packets][2][ch] dwords, zero.
bool lw_mix_applicable(const LwFastLaunch &LL, const LwShortLaunch &LS, int n_cus);
// d_err: the batch's device error word (device-visible host memory): a short block's wave that does not see its edge flags within
// `spin` polls (0 = the default, about a second) raises it and stores nothing.  drop_flags: test hook, no producer signals.
hipError_t lw_launch_mix(const LwDevTables &T, const LwBatchDev &B, const LwFastLaunch &LL, const LwShortLaunch &LS, uint32_t *d_flags,
		uint32_t *d_err, uint32_t spin, bool drop_flags, void *out, int fmt, hipStream_t st);
