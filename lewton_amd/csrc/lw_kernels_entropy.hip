// Entropy stage on the device (product code, gfx950): one WAVE per packet.  The wave runs lw_ent_decode_packet
// (lw_dev_entropy.h -- the same source the CPU suite holds against the host entropy stage bit for bit) with wave-uniform
// control: floor-1 decode + amplitude unwrap into the floor record, residue Huffman / VQ decode accumulated into the packet's
// residue vectors, which live in the wave's LDS until the end and then leave in one coalesced write -- exactly the records
// the host stage would have staged, so the synthesis kernels behind it do not know the difference.
//
// Why a wave and not a lane per packet (first version of this kernel, 2.7 ms per 4096 packets whatever was tuned): on
// gfx9 stores and loads share one counter, so with the accumulators in HBM every codeword's table look-up also waited for
// the previous codeword's stores to be acknowledged; 64 packets per wave made every lane sit through every other lane's
// branch; and per-lane tables meant flat addressing.  Here the accumulators are LDS (ds_ instructions, their own
// counter), the control flow is uniform, one codeword's vector is added by dims lanes at once.
//
// Bound: latency -- a packet is a serial chain of ~700 codewords, each a dependent table look-up (L1 / L2-resident
// tables: the image of a typical setup is a few hundred KB; ~70-130 ns per dependent access on this chip,
// tools/micro/chase.hip).  Up to 16 packets per CU in flight (9 KB of LDS each).  Algorithmic bytes per packet: the packet
// itself in (~0.5 KB), floor records + residue vectors out (8.3 KB for a stereo long block).
#include "lw_dev_entropy.h"
#include "lw_kernels.hpp"

#include <hip/hip_runtime.h>

template <bool GENERAL> // GENERAL: mappings with several submaps (vectors of a submap = a subset of the channels)
__global__ void __launch_bounds__(64) k_entropy(LwEntTables T, const LwEntPacket *pk, const LwPacketRec *recs, const uint32_t *pool,
		uint16_t *floors, float *residue, uint32_t n)
{
	extern __shared__ __attribute__((aligned(16))) float smem[]; // [T.res_floats accumulators][64 dump slots][posts][digits]
	const uint32_t i = blockIdx.x; // wave-uniform: the decode state stays in SGPRs
	const LwPacketRec rec = recs[i];
	if (rec.flags & LW_RF_SKIP)
		return;
	const LwEntPacket p = pk[i];
	const uint32_t blk = 1u << rec.bs, res_n = T.ch * (blk >> 1);
	for (uint32_t k = threadIdx.x; k < res_n; k += 64)
		smem[k] = 0.0f;
	smem[T.res_floats + threadIdx.x] = 0.0f;
	__syncthreads();
	LwEntAcc acc = (LwEntAcc)smem;
	LwEntPosts posts = (LwEntPosts)(smem + T.res_floats + LW_ENT_DUMP_FLOATS);
	LwEntDigits digits = (LwEntDigits)(smem + T.res_floats + LW_ENT_DUMP_FLOATS + LW_ENT_POSTS_BYTES / 4u);
	lw_ent_decode_packet(T, (const LW_K uint32_t *)(pool + p.word_off), p.len, p.start_bit, rec.mode, blk, floors + rec.floor_off, acc, posts, digits, GENERAL);
	__syncthreads();
	// residue blocks start at multiples of ch * n0 / 2 floats: 16-byte aligned
	float *out = (float *)__builtin_assume_aligned(residue + rec.res_off, 16);
	for (uint32_t k = threadIdx.x * 4; k < res_n; k += 256) {
		float4 v;
		v.x = smem[k];
		v.y = smem[k + 1];
		v.z = smem[k + 2];
		v.w = smem[k + 3];
		*(float4 *)(out + k) = v;
	}
}

hipError_t lw_launch_entropy(const LwEntTables &T, const LwEntPacket *d_pk, const LwPacketRec *d_recs, const uint32_t *d_pool,
		uint16_t *d_floor, float *d_res, uint32_t n, hipStream_t st)
{
	if (n == 0)
		return hipSuccess;
	const size_t lds = ((size_t)T.res_floats + LW_ENT_DUMP_FLOATS) * 4 + T.ws_bytes;
	static LwPerDeviceOnce once;
	{
		const hipError_t e = once.run([] {
			hipError_t r = hipFuncSetAttribute((const void *)k_entropy<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LW_ENT_MAX_LDS);
			if (r == hipSuccess)
				r = hipFuncSetAttribute((const void *)k_entropy<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LW_ENT_MAX_LDS);
			return r;
		});
		if (e != hipSuccess)
			return e; // the caller reports this launch; the next one tries again
	}
	if (T.general)
		return lw_launch_k(k_entropy<true>, dim3(n), dim3(64), lds, st, T, d_pk, d_recs, d_pool, d_floor, d_res, n);
	return lw_launch_k(k_entropy<false>, dim3(n), dim3(64), lds, st, T, d_pk, d_recs, d_pool, d_floor, d_res, n);
}
