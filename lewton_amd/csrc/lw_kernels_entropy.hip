// Entropy stage on the device (product code, gfx950): one LANE per packet, 64 packets per wave.  Every lane runs
// lw_ent_decode_packet (lw_dev_entropy.h -- the same source the CPU suite holds against the host entropy stage bit for bit)
// on its packet: floor-1 decode + amplitude unwrap into the floor record, residue Huffman / VQ decode accumulated straight
// into the packet's residue vectors [ch][n/2] in HBM (zeroed by a memset in front of this kernel), i.e. exactly the records
// the host stage would have staged -- the synthesis kernels behind it do not know the difference.
//
// Bound: latency.  A packet is a serial chain of ~700 codewords, each a dependent table look-up (L2-resident tables: the
// image of a typical setup is a few hundred KB) plus the read-modify-write of 1-8 residue elements that nothing later in
// the chain waits for.  Lanes of a wave diverge only in trip counts (a partition takes psize / dims codewords), not in
// the loop structure; packets of similar size finish together.  Algorithmic bytes per packet: the packet itself in
// (~0.5 KB), floor records + residue vectors out (8.3 KB for a stereo long block).
#include "lw_dev_entropy.h"
#include "lw_kernels.hpp"

#include <hip/hip_runtime.h>

__global__ void __launch_bounds__(64) k_entropy(LwEntTables T, const LwEntPacket *pk, const LwPacketRec *recs, const uint32_t *pool,
		uint16_t *floors, float *residue, uint8_t *ws, uint32_t n)
{
	const uint32_t i = blockIdx.x * 64u + threadIdx.x;
	if (i >= n)
		return;
	const LwPacketRec rec = recs[i];
	if (rec.flags & LW_RF_SKIP)
		return;
	const LwEntPacket p = pk[i];
	lw_ent_decode_packet(T, pool + p.word_off, p.len, p.start_bit, rec.mode, 1u << rec.bs, floors + rec.floor_off,
			residue + rec.res_off, ws + (size_t)i * T.ws_bytes);
}

void lw_launch_entropy(const LwEntTables &T, const LwEntPacket *d_pk, const LwPacketRec *d_recs, const uint32_t *d_pool,
		uint16_t *d_floor, float *d_res, uint8_t *d_ws, uint32_t n, hipStream_t st)
{
	if (n == 0)
		return;
	hipLaunchKernelGGL(k_entropy, dim3((n + 63) / 64), dim3(64), 0, st, T, d_pk, d_recs, d_pool, d_floor, d_res, d_ws, n);
}
