// Specialised long-block synthesis kernel for gfx950 (MI355X): n = 2048, window flags (1,1).
//
// One wave64 handles one "unit" (a coupled channel pair, or a single channel) of one packet from the
// entropy-decoded record to PCM:
//     residue load (coalesced float4) -> inverse coupling -> floor-1 curve -> floor x residue
//     -> IMDCT entirely in registers with three wave-private LDS transposes -> window/overlap-add
//     -> i16/f32 store (coalesced).
// 16 waves form a workgroup (one per CU, 150 KB LDS: 22 KB re-ordered twiddle/window tables staged once
// + 8 KB transpose scratch per wave).  Consecutive packets of a stream sit in consecutive waves; a packet's
// un-windowed right half is handed to its successor through LDS (one s_barrier per workgroup); across
// workgroup boundaries it comes from a halo buffer filled by a RIGHT_ONLY pre-pass of this same kernel, at
// run starts from the stream's state slot.
//
// Register layouts of the 512 complex pairs p (u[2p], u[2p+1]) of imdct.rs's butterfly array, 8 per lane:
//   B: lane = p[5:0], reg = p[8:6]   step 2 and stages l = 0,1   (pair bits 8,7,6 are lane-local)
//   C: lane = (p[8:6], p[2:0]), reg = p[5:3]   stages l = 2,3,4
//   D: lane = p[8:3], reg = p[2:0]   fused last three stages (imdct.rs:234-288), lane-local
//   E: lane handles m' = 2*lane + c: bit-reverse gather (imdct.rs:490-528), step 7, step 8, overlap-add
// The numpy model tests/fast_model.py is the executable specification of these layouts; it is checked
// bit-for-bit against the oracle on the CPU.
//
// Arithmetic contract: identical to lw_kernels.hip -- same f32 operations on the same operands as the
// reference, compiled with -ffp-contract=off.  The floor curve uses trunc((t*dy +- 0.5) * (1/adx)), proven
// equal to the reference's integer render_line for every reachable segment (tests/test_fast_model.py).
#include "lw_fast.hpp"
#include "lw_kernels.hpp"

#define LW_NONE 0xFFFFFFFFu
#define LW_WG (64 * LW_FAST_WAVES)
#define LW_SCR_FLOATS 2048 // per wave: [2 channels][1024 floats]

struct LwFastArgs {
	LwFastImage off;
	const uint8_t *image;      // LDS image in HBM
	const LwFastItem *items;   // (packet index, halo slot) in stream-sorted order
	uint32_t n_items;
	uint32_t n_units;
	LwFastUnit units[LW_FAST_WAVES];
	float *halo;               // [slots][ch][512]
	void *out;
};

typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void lds_fence()
{
	asm volatile("" ::: "memory"); // wave-private LDS traffic is in order in hardware; stop compiler motion only
}

struct Pair {
	float e0, e1;
};

// imdct.rs:36-41 / :94-99 / :161-166 on (hi, lo) pairs
__device__ __forceinline__ void bfly(Pair &H, Pair &L, float2_t t)
{
	const float k00 = H.e1 - L.e1;
	const float k01 = H.e0 - L.e0;
	H.e1 = H.e1 + L.e1;
	H.e0 = H.e0 + L.e0;
	L.e1 = k00 * t.x - k01 * t.y;
	L.e0 = k01 * t.x + k00 * t.y;
}

__device__ __forceinline__ float2_t lds2(const char *base, uint32_t byte_off)
{
	return *reinterpret_cast<const float2_t *>(base + byte_off);
}

__device__ __forceinline__ float4_t lds4(const char *base, uint32_t byte_off)
{
	return *reinterpret_cast<const float4_t *>(base + byte_off);
}

// samples.rs:92-103
__device__ __forceinline__ int to_i16s(float x)
{
	const float t = x * 32768.0f;
	if (t > 32767.0f)
		return 32767;
	if (t < -32768.0f)
		return -32768;
	return (int)t;
}

__device__ __forceinline__ uint32_t pack2(float a, float b)
{
	return ((uint32_t)to_i16s(a) & 0xffffu) | ((uint32_t)to_i16s(b) << 16);
}

template <int FMT, bool RIGHT_ONLY>
__global__ void __launch_bounds__(LW_WG) k_long(LwDevTables T, LwBatchDev B, LwFastArgs F)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

	// ---- stage the table image (all waves), then everything below is wave-private until the hand-over
	{
		const uint4 *src = reinterpret_cast<const uint4 *>(F.image);
		uint4 *dst = reinterpret_cast<uint4 *>(smem);
		for (uint32_t i = threadIdx.x; i < F.off.total / 16; i += LW_WG)
			dst[i] = src[i];
	}
	__syncthreads();
	const char *img = smem;
	float *scr = reinterpret_cast<float *>(smem + F.off.total) + wave * LW_SCR_FLOATS;
	char *scrb = reinterpret_cast<char *>(scr);

	const uint32_t g = blockIdx.x * LW_FAST_WAVES + wave;
	const uint32_t item = g / F.n_units, uidx = g - item * F.n_units;
	const bool valid = item < F.n_items;

	// per-channel results that live across the barrier
	float pbv[2][2][4]; // [channel][c][k] un-windowed right half at q_k(c)
	float pav[2][2][4];
	LwPacketRec rec{};
	LwFastUnit un{};
	int nch = 0;
	uint32_t pkt = 0;

	if (valid) {
		pkt = F.items[item].pkt;
		rec = B.recs[pkt];
		un = F.units[uidx];
		nch = un.ch_b >= 0 ? 2 : 1;
		const int chn[2] = {un.ch_a, un.ch_b >= 0 ? un.ch_b : un.ch_a};
		const int fslot[2] = {un.floor_a, un.floor_b};

		// ---- residue loads: lane holds float4 groups m = 64x + lane (coalesced 1 KB per instruction)
		float4_t r[2][4];
#pragma unroll
		for (int c = 0; c < 2; c++) {
			if (c < nch) {
				const float4_t *src = reinterpret_cast<const float4_t *>(B.residue + rec.res_off + (uint32_t)chn[c] * 1024u);
#pragma unroll
				for (int x = 0; x < 4; x++)
					r[c][x] = src[64 * x + lane];
			}
		}
		// ---- floor segment tables (one 16-byte entry per static interval), built by lanes = posts
		bool unused[2] = {false, false};
#pragma unroll
		for (int c = 0; c < 2; c++) {
			if (c < nch) {
				const uint32_t Fp = T.floor_F[T.mode_floor[rec.mode * T.ch + chn[c]]];
				const uint16_t *frec = B.floors + rec.floor_off + (uint32_t)chn[c] * T.fstride;
				const uint32_t e = lane < Fp ? frec[lane] : 0u;
				unused[c] = __builtin_amdgcn_readfirstlane(e) == LW_FLOOR_UNUSED;
				const unsigned long long M = __ballot((e & LW_POST_ACTIVE) != 0);
				const unsigned long long lowmask = (2ull << lane) - 1ull;
				const unsigned long long below = M & lowmask, above = M & ~lowmask;
				const int lo = below ? 63 - __builtin_clzll(below) : 0;
				const int hi = above ? __builtin_ctzll(above) : -1;
				const int y = (int)(e & 0xffu);
				const float xs = *reinterpret_cast<const float *>(img + F.off.xsf + 4u * (64u * fslot[c] + lane));
				const int ylo = __builtin_amdgcn_ds_bpermute(lo << 2, y);
				const int yhi = __builtin_amdgcn_ds_bpermute((hi < 0 ? lo : hi) << 2, y);
				const float xlo = __int_as_float(__builtin_amdgcn_ds_bpermute(lo << 2, __float_as_int(xs)));
				const float xhi = __int_as_float(__builtin_amdgcn_ds_bpermute((hi < 0 ? lo : hi) << 2, __float_as_int(xs)));
				float4_t ent;
				ent.x = xlo;
				ent.y = hi < 0 ? 0.0f : (float)(yhi - ylo);
				ent.z = hi < 0 ? 1.0f : __builtin_amdgcn_rcpf(xhi - xlo);
				ent.w = __int_as_float(ylo << 2);
				if (lane < Fp)
					*reinterpret_cast<float4_t *>(scrb + 4096 * c + 16 * lane) = ent;
			}
		}
		lds_fence();
		// ---- inverse coupling (audio.rs:762-777, :990-1002) on the raw residues
		if (nch == 2 && un.coupled) {
#pragma unroll
			for (int x = 0; x < 4; x++) {
#pragma unroll
				for (int j = 0; j < 4; j++) {
					const float m = r[0][x][j], a = r[1][x][j];
					const float s = m + a, d = m - a;
					const bool mp = m > 0.0f, ap = a > 0.0f;
					r[0][x][j] = mp ? (ap ? m : s) : (ap ? m : d);
					r[1][x][j] = mp ? (ap ? d : m) : (ap ? s : m);
				}
			}
		}
		// ---- floor value per bin; spectrum = floor * residue in place (audio.rs:1035-1037)
#pragma unroll
		for (int c = 0; c < 2; c++) {
			if (c < nch) {
				if (unused[c]) {
#pragma unroll
					for (int x = 0; x < 4; x++)
#pragma unroll
						for (int j = 0; j < 4; j++)
							r[c][x][j] = 0.0f * r[c][x][j]; // zero floor (audio.rs:1021-1024)
				} else {
#pragma unroll
					for (int x = 0; x < 4; x++) {
						const uint2_t sw = *reinterpret_cast<const uint2_t *>(
								img + F.off.sid16 + 8u * ((fslot[c] * 4 + x) * 64u + lane));
						const uint32_t s16[4] = {sw.x & 0xffffu, sw.x >> 16, sw.y & 0xffffu, sw.y >> 16};
						const float kf = (float)(256 * x + 4 * (int)lane);
#pragma unroll
						for (int j = 0; j < 4; j++) {
							const float4_t ent = lds4(scrb + 4096 * c, s16[j]);
							const float tf = (kf + (float)j) - ent.x;
							const float z = __builtin_fmaf(tf, ent.y, __builtin_copysignf(0.5f, ent.y)); // exact: |t*dy| < 2^18
							const int q = (int)(z * ent.z);
							const uint32_t idx = (uint32_t)((q << 2) + __float_as_int(ent.w));
							r[c][x][j] = *reinterpret_cast<const float *>(img + F.off.inv_db + idx) * r[c][x][j];
						}
					}
				}
			}
		}
		lds_fence();

		// ---- IMDCT step 1 (imdct.rs:337-371) in the load layout, exchange with the mirror lane -> layout B
		Pair P[2][8];
		const uint32_t mirror = (63u - lane) << 2;
#pragma unroll
		for (int x = 0; x < 4; x++) {
			const uint32_t m = 64u * x + lane;
			const float2_t au = lds2(img + F.off.apair, 8u * m);
			const float2_t al = lds2(img + F.off.apair, 8u * (511u - m));
#pragma unroll
			for (int c = 0; c < 2; c++) {
				if (c < nch) {
					const float X0 = r[c][x][0], X1 = r[c][x][1], X2 = r[c][x][2], X3 = r[c][x][3];
					const float u1 = X0 * au.x - X2 * au.y;
					const float u0 = X0 * au.y + X2 * au.x;
					P[c][x].e1 = (-X3) * al.x - (-X1) * al.y;
					P[c][x].e0 = (-X3) * al.y + (-X1) * al.x;
					P[c][7 - x].e0 = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(u0)));
					P[c][7 - x].e1 = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(u1)));
				}
			}
		}
		// ---- step 2 (imdct.rs:385-430) and stages l = 0, 1 (imdct.rs:445-452)
#pragma unroll
		for (int x = 0; x < 4; x++) {
			const float2_t t = lds2(img + F.off.tw_s2, 8u * (64u * x + lane));
#pragma unroll
			for (int c = 0; c < 2; c++)
				if (c < nch)
					bfly(P[c][x + 4], P[c][x], t);
		}
#pragma unroll
		for (int b = 0; b < 2; b++) {
			const float2_t t = lds2(img + F.off.tw_l0, 8u * (64u * b + lane));
#pragma unroll
			for (int c = 0; c < 2; c++)
				if (c < nch) {
					bfly(P[c][2 + b], P[c][b], t);
					bfly(P[c][6 + b], P[c][4 + b], t);
				}
		}
		{
			const float2_t t = lds2(img + F.off.tw_l1, 8u * lane);
#pragma unroll
			for (int c = 0; c < 2; c++)
				if (c < nch) {
#pragma unroll
					for (int x = 1; x < 8; x += 2)
						bfly(P[c][x], P[c][x - 1], t);
				}
		}
		// ---- T2: layout B -> C.  slot(p) = p ^ ((p>>6 & 3) << 3)
		const uint32_t X3b = lane >> 3, lo3 = lane & 7u;
#pragma unroll
		for (int c = 0; c < 2; c++)
			if (c < nch) {
#pragma unroll
				for (int x = 0; x < 8; x++) {
					const uint32_t slot = 64u * x + (lane ^ ((x & 3u) << 3));
					*reinterpret_cast<float2_t *>(scrb + 4096 * c + 8u * slot) = float2_t{P[c][x].e0, P[c][x].e1};
				}
			}
		lds_fence();
		Pair Q[2][8];
#pragma unroll
		for (int c = 0; c < 2; c++)
			if (c < nch) {
#pragma unroll
				for (int y = 0; y < 8; y++) {
					const uint32_t slot = 64u * X3b + lo3 + 8u * ((uint32_t)y ^ (X3b & 3u));
					const float2_t v = lds2(scrb + 4096 * c, 8u * slot);
					Q[c][y].e0 = v.x;
					Q[c][y].e1 = v.y;
				}
			}
		lds_fence();
		// ---- stages l = 2, 3, 4 (imdct.rs:454-477)
#pragma unroll
		for (int yy = 0; yy < 4; yy++) {
			const float2_t t = lds2(img + F.off.tw_l2, 8u * (8u * yy + lo3));
#pragma unroll
			for (int c = 0; c < 2; c++)
				if (c < nch)
					bfly(Q[c][4 + yy], Q[c][yy], t);
		}
#pragma unroll
		for (int b = 0; b < 2; b++) {
			const float2_t t = lds2(img + F.off.tw_l3, 8u * (8u * b + lo3));
#pragma unroll
			for (int c = 0; c < 2; c++)
				if (c < nch) {
					bfly(Q[c][2 + b], Q[c][b], t);
					bfly(Q[c][6 + b], Q[c][4 + b], t);
				}
		}
		{
			const float2_t t = lds2(img + F.off.tw_l4, 8u * lo3);
#pragma unroll
			for (int c = 0; c < 2; c++)
				if (c < nch) {
#pragma unroll
					for (int y = 1; y < 8; y += 2)
						bfly(Q[c][y], Q[c][y - 1], t);
				}
		}
		// ---- T3: layout C -> D.  slot(p) = 8 nu + (z ^ (nu>>2 & 7)), nu = p>>3, z = p&7
#pragma unroll
		for (int c = 0; c < 2; c++)
			if (c < nch) {
#pragma unroll
				for (int y = 0; y < 8; y++) {
					const uint32_t nu = 8u * X3b + y;
					const uint32_t slot = 8u * nu + (lo3 ^ ((nu >> 2) & 7u));
					*reinterpret_cast<float2_t *>(scrb + 4096 * c + 8u * slot) = float2_t{Q[c][y].e0, Q[c][y].e1};
				}
			}
		lds_fence();
		float z[2][16];
#pragma unroll
		for (int c = 0; c < 2; c++)
			if (c < nch) {
#pragma unroll
				for (int zz = 0; zz < 8; zz++) {
					const uint32_t slot = 8u * lane + ((uint32_t)zz ^ ((lane >> 2) & 7u));
					const float2_t v = lds2(scrb + 4096 * c, 8u * slot);
					z[c][2 * zz] = v.x;
					z[c][2 * zz + 1] = v.y;
				}
			}
		lds_fence();
		// ---- fused last three stages (imdct.rs:234-288), lane-local on z[0..16) = u[16 lane ..]
		const float a2 = *reinterpret_cast<const float *>(img + F.off.a2);
#pragma unroll
		for (int c = 0; c < 2; c++)
			if (c < nch) {
				float *zc = z[c];
				float k00, k11;
				k00 = zc[15] - zc[7];
				k11 = zc[14] - zc[6];
				zc[15] = zc[15] + zc[7];
				zc[14] = zc[14] + zc[6];
				zc[7] = k00;
				zc[6] = k11;
				k00 = zc[13] - zc[5];
				k11 = zc[12] - zc[4];
				zc[13] = zc[13] + zc[5];
				zc[12] = zc[12] + zc[4];
				zc[5] = (k00 + k11) * a2;
				zc[4] = (k11 - k00) * a2;
				k00 = zc[3] - zc[11];
				k11 = zc[10] - zc[2];
				zc[11] = zc[11] + zc[3];
				zc[10] = zc[10] + zc[2];
				zc[3] = k11;
				zc[2] = k00;
				k00 = zc[1] - zc[9];
				k11 = zc[8] - zc[0];
				zc[9] = zc[9] + zc[1];
				zc[8] = zc[8] + zc[0];
				zc[1] = (k00 + k11) * a2;
				zc[0] = (k00 - k11) * a2;
#pragma unroll
				for (int b = 8; b >= 0; b -= 8) { // imdct.rs:202-232
					float *w = zc + b;
					const float i00 = w[7] - w[3], y0 = w[7] + w[3], y2 = w[5] + w[1], k22 = w[5] - w[1];
					const float k33 = w[4] - w[0], i11 = w[6] - w[2], y1 = w[6] + w[2], y3 = w[4] + w[0];
					w[7] = y0 + y2;
					w[5] = y0 - y2;
					w[3] = i00 + k33;
					w[1] = i00 - k33;
					w[6] = y1 + y3;
					w[4] = y1 - y3;
					w[2] = i11 - k22;
					w[0] = i11 + k22;
				}
			}
		// ---- T4: layout D -> bit-reverse gather.  slot(p) = (p & ~127) | ((p & 3) << 5) | ((p >> 2) & 31)
		{
			const uint32_t base = 128u * (lane >> 4) + 2u * (lane & 15u);
#pragma unroll
			for (int c = 0; c < 2; c++)
				if (c < nch) {
#pragma unroll
					for (int zz = 0; zz < 8; zz++) {
						const uint32_t slot = base + 32u * (zz & 3) + (zz >> 2);
						*reinterpret_cast<float2_t *>(scrb + 4096 * c + 8u * slot) = float2_t{z[c][2 * zz], z[c][2 * zz + 1]};
					}
				}
		}
		lds_fence();
		// rho = rev6(lane); s0 = slot(2 * rho): pairs 2v, 2v+256, 255-2v, 511-2v
		const uint32_t rho = __builtin_bitreverse32(lane) >> 26;
		const uint32_t s0 = ((rho & 1u) << 6) | (rho >> 1);
#pragma unroll
		for (int c2 = 0; c2 < 2; c2++) { // m' = 2 lane + c2
			const float4_t Cq = lds4(img + F.off.c4, 16u * (64u * c2 + lane));
			const float4_t Bl = lds4(img + F.off.b_lo, 16u * (64u * c2 + lane));
			const float4_t Bh = lds4(img + F.off.b_hi, 16u * (64u * c2 + lane));
			const uint32_t sa = 128u * c2 + s0;         // slot of pair 2v
			const uint32_t sb = 255u - sa;              // slot of pair 255 - 2v
#pragma unroll
			for (int c = 0; c < 2; c++)
				if (c < nch) {
					const char *sc = scrb + 4096 * c;
					const float2_t pq = lds2(sc, 8u * sa), pq256 = lds2(sc, 8u * (sa + 256u));
					const float2_t p255 = lds2(sc, 8u * sb), p511 = lds2(sc, 8u * (sb + 256u));
					const float D0i = p511.y, D1i = p511.x, D2i = p255.y, D3i = p255.x;
					const float E0i = pq256.y, E1i = pq256.x, E2i = pq.y, E3i = pq.x;
					// step 7 (imdct.rs:547-579)
					float a02 = D0i - E2i, a11 = D1i + E3i;
					float b0 = Cq.y * a02 + Cq.x * a11, b1 = Cq.y * a11 - Cq.x * a02;
					float b2 = D0i + E2i, b3 = D1i - E3i;
					const float D0 = b2 + b0, D1 = b3 + b1, E2 = b2 - b0, E3 = b1 - b3;
					a02 = D2i - E0i;
					a11 = D3i + E1i;
					b0 = Cq.w * a02 + Cq.z * a11;
					b1 = Cq.w * a11 - Cq.z * a02;
					b2 = D2i + E0i;
					b3 = D3i - E1i;
					const float D2 = b2 + b0, D3 = b3 + b1, E0 = b2 - b0, E1 = b1 - b3;
					// step 8 (imdct.rs:618-657): q = 511-2m', 510-2m', 1+2m', 2m'
					pav[c][c2][0] = D0 * Bl.y - D1 * Bl.x;
					pbv[c][c2][0] = (-D0) * Bl.x - D1 * Bl.y;
					pav[c][c2][1] = D2 * Bl.w - D3 * Bl.z;
					pbv[c][c2][1] = (-D2) * Bl.z - D3 * Bl.w;
					pav[c][c2][2] = E0 * Bh.y - E1 * Bh.x;
					pbv[c][c2][2] = (-E0) * Bh.x - E1 * Bh.y;
					pav[c][c2][3] = E2 * Bh.w - E3 * Bh.z;
					pbv[c][c2][3] = (-E2) * Bh.z - E3 * Bh.w;
				}
		}
		lds_fence();
		// ---- publish the right half for the successor wave (own scratch, [channel][c2][lane] float4)
#pragma unroll
		for (int c = 0; c < 2; c++)
			if (c < nch) {
#pragma unroll
				for (int c2 = 0; c2 < 2; c2++)
					*reinterpret_cast<float4_t *>(scrb + 4096 * c + 16u * (64u * c2 + lane)) =
						float4_t{pbv[c][c2][0], pbv[c][c2][1], pbv[c][c2][2], pbv[c][c2][3]};
			}
	}
	__syncthreads();
	if (!valid)
		return;

	const int chn[2] = {un.ch_a, un.ch_b >= 0 ? un.ch_b : un.ch_a};
	// q positions: group 0 = [4 lane .. +3], group 1 = [508 - 4 lane .. +3] (ascending); value order per DESIGN.md
	if (RIGHT_ONLY) {
		const uint32_t hs = F.items[item].halo;
#pragma unroll
		for (int c = 0; c < 2; c++)
			if (c < nch) {
				float *dst = F.halo + ((size_t)hs * T.ch + chn[c]) * 512u;
				*reinterpret_cast<float4_t *>(dst + 4u * lane) =
					float4_t{pbv[c][0][3], pbv[c][0][2], pbv[c][1][3], pbv[c][1][2]};
				*reinterpret_cast<float4_t *>(dst + 508u - 4u * lane) =
					float4_t{pbv[c][1][1], pbv[c][1][0], pbv[c][0][1], pbv[c][0][0]};
			}
		return;
	}

	// ---- previous right half: LDS (predecessor wave), state slot, halo buffer, or a generic packet's td block
	if (rec.prev != -1) {
		bool from_lds = false;
		const float *gsrc[2] = {nullptr, nullptr};
		if (rec.prev <= -2) {
			const uint32_t slot = (uint32_t)(-(rec.prev + 2));
			const uint32_t par = (rec.flags & LW_RF_PARITY_IN) ? 1u : 0u;
			const float *st = B.state + ((size_t)slot * 2 + par) * T.state_stride;
			gsrc[0] = st + (uint32_t)chn[0] * T.state_chan_stride;
			gsrc[1] = st + (uint32_t)chn[1] * T.state_chan_stride;
		} else if (wave >= F.n_units && F.items[item - 1].pkt == (uint32_t)rec.prev) {
			from_lds = true;
		} else if (F.items[item].halo != LW_NONE) {
			const float *h = F.halo + (size_t)F.items[item].halo * T.ch * 512u;
			gsrc[0] = h + (uint32_t)chn[0] * 512u;
			gsrc[1] = h + (uint32_t)chn[1] * 512u;
		} else {
			const LwPacketRec pr = B.recs[rec.prev];
			const float *td = B.td + 2u * (size_t)pr.res_off + 1024u;
			gsrc[0] = td + (uint32_t)chn[0] * 2048u;
			gsrc[1] = td + (uint32_t)chn[1] * 2048u;
		}
		const float *pscr = scr - (size_t)F.n_units * LW_SCR_FLOATS; // predecessor wave's scratch
#pragma unroll
		for (int c = 0; c < 2; c++)
			if (c < nch) {
				float pp[2][4]; // previous packet's pb at this lane's q positions
				if (from_lds) {
#pragma unroll
					for (int c2 = 0; c2 < 2; c2++) {
						const float4_t v = *reinterpret_cast<const float4_t *>(
								reinterpret_cast<const char *>(pscr) + 4096 * c + 16u * (64u * c2 + lane));
						pp[c2][0] = v.x;
						pp[c2][1] = v.y;
						pp[c2][2] = v.z;
						pp[c2][3] = v.w;
					}
				} else {
					const float4_t g0 = *reinterpret_cast<const float4_t *>(gsrc[c] + 4u * lane);
					const float4_t g1 = *reinterpret_cast<const float4_t *>(gsrc[c] + 508u - 4u * lane);
					pp[0][3] = g0.x; // q = 4 lane
					pp[0][2] = g0.y; // 4 lane + 1
					pp[1][3] = g0.z; // 4 lane + 2
					pp[1][2] = g0.w; // 4 lane + 3
					pp[1][1] = g1.x; // 508 - 4 lane
					pp[1][0] = g1.y; // 509 - 4 lane
					pp[0][1] = g1.z; // 510 - 4 lane
					pp[0][0] = g1.w; // 511 - 4 lane
				}
				// ---- window + overlap-add (audio.rs:1116-1118): out[q] and out[1023-q] from (pa, pb', s[q], s[1023-q])
				float oq[2][4], om[2][4];
#pragma unroll
				for (int c2 = 0; c2 < 2; c2++) {
					const float4_t w0 = lds4(img + F.off.win, 32u * (64u * c2 + lane));
					const float4_t w1 = lds4(img + F.off.win, 32u * (64u * c2 + lane) + 16u);
					const float sq[4] = {w0.x, w0.z, w1.x, w1.z}, sr[4] = {w0.y, w0.w, w1.y, w1.w};
#pragma unroll
					for (int k = 0; k < 4; k++) {
						oq[c2][k] = (pav[c][c2][k] * sq[k]) + (pp[c2][k] * sr[k]);
						om[c2][k] = ((-pav[c][c2][k]) * sr[k]) + (pp[c2][k] * sq[k]);
					}
				}
				// positions: [4l..4l+3] = oq(0,3) oq(0,2) oq(1,3) oq(1,2); [508-4l..] = oq(1,1) oq(1,0) oq(0,1) oq(0,0)
				//            [512+4l..] = om(0,0) om(0,1) om(1,0) om(1,1); [1020-4l..] = om(1,2) om(1,3) om(0,2) om(0,3)
				const float g0[4] = {oq[0][3], oq[0][2], oq[1][3], oq[1][2]};
				const float g1[4] = {oq[1][1], oq[1][0], oq[0][1], oq[0][0]};
				const float g2[4] = {om[0][0], om[0][1], om[1][0], om[1][1]};
				const float g3[4] = {om[1][2], om[1][3], om[0][2], om[0][3]};
				const uint32_t p0 = 4u * lane, p1 = 508u - 4u * lane, p2 = 512u + 4u * lane, p3 = 1020u - 4u * lane;
				if (FMT == LW_OUT_I16_PLANAR) {
					int16_t *o = reinterpret_cast<int16_t *>(F.out) + rec.out_off + (uint32_t)chn[c] * 1024u;
					*reinterpret_cast<uint2_t *>(o + p0) = uint2_t{pack2(g0[0], g0[1]), pack2(g0[2], g0[3])};
					*reinterpret_cast<uint2_t *>(o + p1) = uint2_t{pack2(g1[0], g1[1]), pack2(g1[2], g1[3])};
					*reinterpret_cast<uint2_t *>(o + p2) = uint2_t{pack2(g2[0], g2[1]), pack2(g2[2], g2[3])};
					*reinterpret_cast<uint2_t *>(o + p3) = uint2_t{pack2(g3[0], g3[1]), pack2(g3[2], g3[3])};
				} else if (FMT == LW_OUT_F32_PLANAR) {
					float *o = reinterpret_cast<float *>(F.out) + rec.out_off + (uint32_t)chn[c] * 1024u;
					*reinterpret_cast<float4_t *>(o + p0) = float4_t{g0[0], g0[1], g0[2], g0[3]};
					*reinterpret_cast<float4_t *>(o + p1) = float4_t{g1[0], g1[1], g1[2], g1[3]};
					*reinterpret_cast<float4_t *>(o + p2) = float4_t{g2[0], g2[1], g2[2], g2[3]};
					*reinterpret_cast<float4_t *>(o + p3) = float4_t{g3[0], g3[1], g3[2], g3[3]};
				} else {
					int16_t *o = reinterpret_cast<int16_t *>(F.out) + rec.out_off + (uint32_t)chn[c];
#pragma unroll
					for (int i = 0; i < 4; i++) {
						o[(p0 + i) * T.ch] = (int16_t)to_i16s(g0[i]);
						o[(p1 + i) * T.ch] = (int16_t)to_i16s(g1[i]);
						o[(p2 + i) * T.ch] = (int16_t)to_i16s(g2[i]);
						o[(p3 + i) * T.ch] = (int16_t)to_i16s(g3[i]);
					}
				}
			}
	}
	// ---- raw right half to the stream's state slot and/or to the td block a generic successor reads
	const bool to_state = rec.state_out >= 0, to_td = (rec.flags & LW_RF_WRITE_TD) != 0;
	if (to_state || to_td) {
#pragma unroll
		for (int c = 0; c < 2; c++)
			if (c < nch) {
				const float4_t lo0 = float4_t{pbv[c][0][3], pbv[c][0][2], pbv[c][1][3], pbv[c][1][2]}; // q = 4l..4l+3
				const float4_t lo1 = float4_t{pbv[c][1][1], pbv[c][1][0], pbv[c][0][1], pbv[c][0][0]}; // q = 508-4l..511-4l
				const float4_t hi0 = float4_t{lo1.w, lo1.z, lo1.y, lo1.x};                               // 1023-q for q = 511-4l..508-4l
				const float4_t hi1 = float4_t{lo0.w, lo0.z, lo0.y, lo0.x};
				for (int t = 0; t < 2; t++) {
					float *dst;
					if (t == 0) {
						if (!to_state)
							continue;
						const uint32_t par = (rec.flags & LW_RF_PARITY_OUT) ? 1u : 0u;
						dst = B.state + ((size_t)rec.state_out * 2 + par) * T.state_stride + (uint32_t)chn[c] * T.state_chan_stride;
					} else {
						if (!to_td)
							continue;
						dst = B.td + 2u * (size_t)rec.res_off + (uint32_t)chn[c] * 2048u + 1024u;
					}
					*reinterpret_cast<float4_t *>(dst + 4u * lane) = lo0;
					*reinterpret_cast<float4_t *>(dst + 508u - 4u * lane) = lo1;
					*reinterpret_cast<float4_t *>(dst + 512u + 4u * lane) = hi0;
					*reinterpret_cast<float4_t *>(dst + 1020u - 4u * lane) = hi1;
				}
			}
	}
}

// ---------------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------------
void lw_launch_long(const LwDevTables &T, const LwBatchDev &B, const LwFastLaunch &L, void *out, int fmt, hipStream_t st)
{
	LwFastArgs F{};
	F.off = L.off;
	F.image = L.d_image;
	F.n_units = L.n_units;
	for (uint32_t i = 0; i < L.n_units && i < LW_FAST_WAVES; i++)
		F.units[i] = L.units[i];
	F.halo = L.d_halo;
	F.out = out;
	const size_t lds = (size_t)L.off.total + (size_t)LW_FAST_WAVES * LW_SCR_FLOATS * sizeof(float);
	const uint32_t per_wg = LW_FAST_WAVES / L.n_units;
	static bool attr_done = false;
	if (!attr_done) {
		(void)hipFuncSetAttribute((const void *)k_long<LW_OUT_I16_PLANAR, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		(void)hipFuncSetAttribute((const void *)k_long<LW_OUT_I16_INTERLEAVED, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		(void)hipFuncSetAttribute((const void *)k_long<LW_OUT_F32_PLANAR, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		(void)hipFuncSetAttribute((const void *)k_long<LW_OUT_I16_PLANAR, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		attr_done = true;
	}
	if (L.n_halo_items) {
		F.items = L.d_halo_items;
		F.n_items = L.n_halo_items;
		const uint32_t grid = (L.n_halo_items + per_wg - 1) / per_wg;
		hipLaunchKernelGGL((k_long<LW_OUT_I16_PLANAR, true>), dim3(grid), dim3(LW_WG), lds, st, T, B, F);
	}
	if (L.n_items) {
		F.items = L.d_items;
		F.n_items = L.n_items;
		const uint32_t grid = (L.n_items + per_wg - 1) / per_wg;
		if (fmt == LW_OUT_I16_PLANAR)
			hipLaunchKernelGGL((k_long<LW_OUT_I16_PLANAR, false>), dim3(grid), dim3(LW_WG), lds, st, T, B, F);
		else if (fmt == LW_OUT_I16_INTERLEAVED)
			hipLaunchKernelGGL((k_long<LW_OUT_I16_INTERLEAVED, false>), dim3(grid), dim3(LW_WG), lds, st, T, B, F);
		else
			hipLaunchKernelGGL((k_long<LW_OUT_F32_PLANAR, false>), dim3(grid), dim3(LW_WG), lds, st, T, B, F);
	}
}
