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
	uint32_t items_per_wg;     // packets per workgroup (<= LW_FAST_WAVES / n_units)
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

// ---------------------------------------------------------------------------------------------
// Packed f32 arithmetic.  Every multiply/add of the transform is issued as v_pk_mul_f32 / v_pk_add_f32 on
// register pairs, with op_sel / op_sel_hi choosing the half of each source that feeds the low / high result
// and neg_lo / neg_hi negating sources (exact).  Written as inline asm so that the operation tree is exactly
// the reference's (no contraction, no re-association) and no register shuffling is left to the vectoriser.
// tests/fast_model.py emulates these very modifier sets and is checked bit-for-bit against the oracle.
// ---------------------------------------------------------------------------------------------
#define LW_PK(name, op, mods)                                                              \
	__device__ __forceinline__ float2_t name(float2_t a, float2_t b)                       \
	{                                                                                      \
		float2_t d;                                                                        \
		asm(op " %0, %1, %2 " mods : "=v"(d) : "v"(a), "v"(b));                            \
		return d;                                                                          \
	}
LW_PK(pk_add, "v_pk_add_f32", "")                                                              // (a.lo+b.lo, a.hi+b.hi)
LW_PK(pk_sub, "v_pk_add_f32", "neg_lo:[0,1] neg_hi:[0,1]")                                     // (a.lo-b.lo, a.hi-b.hi)
LW_PK(pk_add_A2, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")                  // (a.lo-b.hi, a.hi+b.lo)
LW_PK(pk_add_A3, "v_pk_add_f32", "op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[1,0]")     // (a.hi-b.hi, b.lo-a.lo)
LW_PK(pk_add_A4, "v_pk_add_f32", "op_sel:[0,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]")     // (b.lo-a.lo, a.hi-b.hi)
LW_PK(pk_add_A5, "v_pk_add_f32", "op_sel:[1,0] op_sel_hi:[1,0] neg_lo:[0,1]")                  // (a.hi-b.lo, a.hi+b.lo)
LW_PK(pk_add_A6, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")                  // (a.lo+b.hi, a.hi-b.lo)
LW_PK(pk_add_A7, "v_pk_add_f32", "neg_hi:[0,1]")                                               // (a.lo+b.lo, a.hi-b.hi)
LW_PK(pk_add_A8, "v_pk_add_f32", "neg_lo:[0,1]")                                               // (a.lo-b.lo, a.hi+b.hi)
LW_PK(pk_add_A9, "v_pk_add_f32", "neg_lo:[0,1] neg_hi:[1,0]")                                  // (a.lo-b.lo, b.hi-a.hi)
LW_PK(pk_mul, "v_pk_mul_f32", "")                                                              // (a.lo*b.lo, a.hi*b.hi)
LW_PK(pk_mul_M1, "v_pk_mul_f32", "op_sel:[0,0] op_sel_hi:[1,0]")                               // (a.lo*b.lo, a.hi*b.lo)
LW_PK(pk_mul_M2, "v_pk_mul_f32", "op_sel:[1,1] op_sel_hi:[0,1] neg_hi:[0,1]")                  // (a.hi*b.hi, -a.lo*b.hi)
LW_PK(pk_mul_M3, "v_pk_mul_f32", "op_sel:[0,1] op_sel_hi:[0,0]")                               // (a.lo*b.hi, a.lo*b.lo)
LW_PK(pk_mul_M4, "v_pk_mul_f32", "op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]")                  // (a.lo*b.lo, -a.lo*b.hi)
LW_PK(pk_mul_M5, "v_pk_mul_f32", "op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]")     // (-a.hi*b.hi, -a.hi*b.lo)
LW_PK(pk_mul_M6, "v_pk_mul_f32", "op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0]")                  // (-a.hi*b.lo, a.hi*b.hi)
LW_PK(pk_mul_M7, "v_pk_mul_f32", "op_sel:[0,1] op_sel_hi:[1,1]")                               // (a.lo*b.hi, a.hi*b.hi)
LW_PK(pk_mul_M8, "v_pk_mul_f32", "op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[0,1]")                  // (-a.hi*b.lo, a.lo*b.lo)
LW_PK(pk_mul_M9, "v_pk_mul_f32", "op_sel:[1,1] op_sel_hi:[1,0] neg_hi:[1,0]")                  // (a.hi*b.hi, -a.hi*b.lo)
LW_PK(pk_mul_M10, "v_pk_mul_f32", "op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]")    // (-a.lo*b.lo, -a.lo*b.hi)
LW_PK(pk_mul_M12lo, "v_pk_mul_f32", "op_sel:[0,1] op_sel_hi:[0,0]")                            // (a.lo*b.hi, a.lo*b.lo)
LW_PK(pk_mul_M12hi, "v_pk_mul_f32", "op_sel:[1,1] op_sel_hi:[1,0]")                            // (a.hi*b.hi, a.hi*b.lo)

// imdct.rs:36-41 on pairs (e0, e1) = (u[hi-1], u[hi]); t = (t0, t1)
__device__ __forceinline__ void bfly2(float2_t &H, float2_t &L, float2_t t)
{
	const float2_t S = pk_add(H, L);
	const float2_t K = pk_sub(H, L);                  // (k01, k00)
	L = pk_add(pk_mul_M1(K, t), pk_mul_M2(K, t));     // (k01 t0 + k00 t1, k00 t0 - k01 t1)
	H = S;
}

// imdct.rs:548-560 (one half of a step-7 iteration): P = (D1, D0), Q = (E3, E2), C2 = (C0, C1)
__device__ __forceinline__ void step7_block(float2_t P, float2_t Q, float2_t C2, float2_t &Dn, float2_t &En)
{
	const float2_t Aa = pk_add_A7(P, Q);              // (a11, a02)
	const float2_t Bb = pk_add_A8(P, Q);              // (b3, b2)
	const float2_t Bv = pk_add(pk_mul_M7(Aa, C2), pk_mul_M8(Aa, C2)); // (b1, b0)
	Dn = pk_add(Bb, Bv);                              // (b3 + b1, b2 + b0)
	En = pk_add_A9(Bv, Bb);                           // (b1 - b3, b2 - b0)
}

// imdct.rs:619-620: Wv = (w1, w0), Bq = (Bc, Bs) -> (pa, pb)
__device__ __forceinline__ float2_t step8(float2_t Wv, float2_t Bq)
{
	return pk_add(pk_mul_M9(Wv, Bq), pk_mul_M10(Wv, Bq));
}

// ---- phase 1: everything up to the un-windowed halves (pa, pb) of this packet-unit (wave-private)
template <int NCH>
__device__ __forceinline__ void long_phase1(const LwDevTables &T, const LwBatchDev &B, const LwFastArgs &F, const char *img,
		char *scrb, uint32_t lane, const LwPacketRec &rec, const LwFastUnit &un, float4_t (&r)[2][4],
		float2_t (&R)[2][2][4])
{
	const int chn[2] = {un.ch_a, un.ch_b >= 0 ? un.ch_b : un.ch_a};
	const int fslot[2] = {un.floor_a, un.floor_b};
	// ---- floor segment tables (one 16-byte entry per static interval), built by lanes = posts:
	//      {dy, 0.5*sgn(dy) - x0*dy, 1/adx, 4*y0}; y(k) = y0 + trunc((k*dy + c0) * rinv)
	bool unused[2] = {false, false};
#pragma unroll
	for (int c = 0; c < 2; c++) {
		if (c < NCH) {
			const uint32_t Fp = T.floor_F[T.mode_floor[rec.mode * T.ch + chn[c]]];
			const uint16_t *frec = B.floors + rec.floor_off + (uint32_t)chn[c] * T.fstride;
			const uint32_t e = lane < Fp ? frec[lane] : 0u;
			unused[c] = __builtin_amdgcn_readfirstlane(e) == LW_FLOOR_UNUSED;
			const unsigned long long M = __ballot((e & LW_POST_ACTIVE) != 0);
			const unsigned long long lowmask = (2ull << lane) - 1ull;
			const unsigned long long below = M & lowmask, above = M & ~lowmask;
			const int lo = below ? 63 - __builtin_clzll(below) : 0;
			const int hi = above ? __builtin_ctzll(above) : lo;
			const int y = (int)(e & 0xffu);
			const float xs = *reinterpret_cast<const float *>(img + F.off.xsf + 4u * (64u * fslot[c] + lane));
			const int ylo = __builtin_amdgcn_ds_bpermute(lo << 2, y);
			const int yhi = __builtin_amdgcn_ds_bpermute(hi << 2, y);
			const float xlo = __int_as_float(__builtin_amdgcn_ds_bpermute(lo << 2, __float_as_int(xs)));
			const float xhi = __int_as_float(__builtin_amdgcn_ds_bpermute(hi << 2, __float_as_int(xs)));
			const float dy = (float)(yhi - ylo); // 0 when there is no later active post (flat, audio.rs:546-548)
			float4_t ent;
			ent.x = dy;
			ent.y = __builtin_copysignf(0.5f, dy) - xlo * dy; // exact
			ent.z = above ? __builtin_amdgcn_rcpf(xhi - xlo) : 1.0f;
			ent.w = __int_as_float(ylo << 2);
			if (lane < Fp)
				*reinterpret_cast<float4_t *>(scrb + 4096 * c + 16 * lane) = ent;
		}
	}
	lds_fence();
	// ---- inverse coupling (audio.rs:762-777, :990-1002) on the raw residues
	if (NCH == 2 && un.coupled) {
#pragma unroll
		for (int x = 0; x < 4; x++) {
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const float m = r[0][x][j], a = r[1][x][j];
				const float ap = m > 0.0f ? a : -a; // (m, a) -> a>0 ? (m, m-a') : (m+a', m)
				const float s = m + ap, d = m - ap;
				const bool apos = a > 0.0f;
				r[0][x][j] = apos ? m : s;
				r[1][x][j] = apos ? d : m;
			}
		}
	}
	// ---- floor value per bin; spectrum = floor * residue in place (audio.rs:1035-1037)
	const float kf0 = (float)(4 * (int)lane);
#pragma unroll
	for (int c = 0; c < 2; c++) {
		if (c < NCH) {
			if (unused[c]) {
#pragma unroll
				for (int x = 0; x < 4; x++)
					r[c][x] = r[c][x] * 0.0f; // zero floor (audio.rs:1021-1024)
			} else {
#pragma unroll
				for (int x = 0; x < 4; x++) {
					const uint2_t sw = *reinterpret_cast<const uint2_t *>(
							img + F.off.sid16 + 8u * ((fslot[c] * 4 + x) * 64u + lane));
					const uint32_t s16[4] = {sw.x & 0xffffu, sw.x >> 16, sw.y & 0xffffu, sw.y >> 16};
					float4_t fl;
#pragma unroll
					for (int j = 0; j < 4; j++) {
						const float4_t ent = lds4(scrb + 4096 * c, s16[j]);
						const float z = __builtin_fmaf(kf0 + (float)(256 * x + j), ent.x, ent.y); // exact: |k*dy| < 2^18
						const int q = (int)(z * ent.z);
						const uint32_t idx = (uint32_t)((q << 2) + __float_as_int(ent.w));
						fl[j] = *reinterpret_cast<const float *>(img + F.off.inv_db + idx);
					}
					const float2_t lo2 = pk_mul(float2_t{fl.x, fl.y}, float2_t{r[c][x].x, r[c][x].y});
					const float2_t hi2 = pk_mul(float2_t{fl.z, fl.w}, float2_t{r[c][x].z, r[c][x].w});
					r[c][x] = float4_t{lo2.x, lo2.y, hi2.x, hi2.y};
				}
			}
		}
	}
	lds_fence();

	// ---- IMDCT step 1 (imdct.rs:337-371) in the load layout; exchange with the mirror lane -> layout B
	float2_t P[2][8];
	const uint32_t mirror = (63u - lane) << 2;
#pragma unroll
	for (int x = 0; x < 4; x++) {
		const uint32_t m = 64u * x + lane;
		const float2_t au = lds2(img + F.off.apair, 8u * m);          // (A[2m], A[2m+1])
		const float2_t al = lds2(img + F.off.apair, 8u * (511u - m)); // (A[1022-2m], A[1023-2m])
#pragma unroll
		for (int c = 0; c < 2; c++) {
			if (c < NCH) {
				const float2_t Xa = float2_t{r[c][x].x, r[c][x].y}, Xb = float2_t{r[c][x].z, r[c][x].w};
				const float2_t U = pk_add(pk_mul_M3(Xa, au), pk_mul_M4(Xb, au)); // pair 511 - m
				P[c][x] = pk_add(pk_mul_M5(Xb, al), pk_mul_M6(Xa, al));          // pair m
				P[c][7 - x].x = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(U.x)));
				P[c][7 - x].y = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(U.y)));
			}
		}
	}
	// ---- step 2 (imdct.rs:385-430) and stages l = 0, 1 (imdct.rs:445-452)
#pragma unroll
	for (int x = 0; x < 4; x++) {
		const float2_t t = lds2(img + F.off.tw_s2, 8u * (64u * x + lane));
#pragma unroll
		for (int c = 0; c < 2; c++)
			if (c < NCH)
				bfly2(P[c][x + 4], P[c][x], t);
	}
#pragma unroll
	for (int b = 0; b < 2; b++) {
		const float2_t t = lds2(img + F.off.tw_l0, 8u * (64u * b + lane));
#pragma unroll
		for (int c = 0; c < 2; c++)
			if (c < NCH) {
				bfly2(P[c][2 + b], P[c][b], t);
				bfly2(P[c][6 + b], P[c][4 + b], t);
			}
	}
	{
		const float2_t t = lds2(img + F.off.tw_l1, 8u * lane);
#pragma unroll
		for (int c = 0; c < 2; c++)
			if (c < NCH) {
#pragma unroll
				for (int x = 1; x < 8; x += 2)
					bfly2(P[c][x], P[c][x - 1], t);
			}
	}
	// ---- T2: layout B -> C.  slot(p) = p ^ ((p>>6 & 3) << 3)
	const uint32_t X3b = lane >> 3, lo3 = lane & 7u;
#pragma unroll
	for (int c = 0; c < 2; c++)
		if (c < NCH) {
#pragma unroll
			for (int x = 0; x < 8; x++) {
				const uint32_t slot = 64u * x + (lane ^ ((x & 3u) << 3));
				*reinterpret_cast<float2_t *>(scrb + 4096 * c + 8u * slot) = P[c][x];
			}
		}
	lds_fence();
	float2_t Q[2][8];
#pragma unroll
	for (int c = 0; c < 2; c++)
		if (c < NCH) {
#pragma unroll
			for (int y = 0; y < 8; y++) {
				const uint32_t slot = 64u * X3b + lo3 + 8u * ((uint32_t)y ^ (X3b & 3u));
				Q[c][y] = lds2(scrb + 4096 * c, 8u * slot);
			}
		}
	lds_fence();
	// ---- stages l = 2, 3, 4 (imdct.rs:454-477)
#pragma unroll
	for (int yy = 0; yy < 4; yy++) {
		const float2_t t = lds2(img + F.off.tw_l2, 8u * (8u * yy + lo3));
#pragma unroll
		for (int c = 0; c < 2; c++)
			if (c < NCH)
				bfly2(Q[c][4 + yy], Q[c][yy], t);
	}
#pragma unroll
	for (int b = 0; b < 2; b++) {
		const float2_t t = lds2(img + F.off.tw_l3, 8u * (8u * b + lo3));
#pragma unroll
		for (int c = 0; c < 2; c++)
			if (c < NCH) {
				bfly2(Q[c][2 + b], Q[c][b], t);
				bfly2(Q[c][6 + b], Q[c][4 + b], t);
			}
	}
	{
		const float2_t t = lds2(img + F.off.tw_l4, 8u * lo3);
#pragma unroll
		for (int c = 0; c < 2; c++)
			if (c < NCH) {
#pragma unroll
				for (int y = 1; y < 8; y += 2)
					bfly2(Q[c][y], Q[c][y - 1], t);
			}
	}
	// ---- T3: layout C -> D.  slot(p) = 8 nu + (z ^ (nu>>2 & 7)), nu = p>>3, z = p&7
#pragma unroll
	for (int c = 0; c < 2; c++)
		if (c < NCH) {
#pragma unroll
			for (int y = 0; y < 8; y++) {
				const uint32_t nu = 8u * X3b + y;
				const uint32_t slot = 8u * nu + (lo3 ^ ((nu >> 2) & 7u));
				*reinterpret_cast<float2_t *>(scrb + 4096 * c + 8u * slot) = Q[c][y];
			}
		}
	lds_fence();
	float2_t Z[2][8]; // Z[j] = (u[16 lane + 2j], u[16 lane + 2j + 1])
#pragma unroll
	for (int c = 0; c < 2; c++)
		if (c < NCH) {
#pragma unroll
			for (int zz = 0; zz < 8; zz++) {
				const uint32_t slot = 8u * lane + ((uint32_t)zz ^ ((lane >> 2) & 7u));
				Z[c][zz] = lds2(scrb + 4096 * c, 8u * slot);
			}
		}
	lds_fence();
	// ---- fused last three stages (imdct.rs:234-288), lane-local, 28 packed operations per channel
	{
		const float a2s = *reinterpret_cast<const float *>(img + F.off.a2);
		const float2_t a2 = float2_t{a2s, a2s};
#pragma unroll
		for (int c = 0; c < 2; c++)
			if (c < NCH) {
				float2_t *z = Z[c];
				float2_t t0, t1;
				t0 = pk_add(z[7], z[3]);
				z[3] = pk_sub(z[7], z[3]);
				z[7] = t0;
				t0 = pk_add(z[6], z[2]);
				t1 = pk_sub(z[6], z[2]);                    // (k11, k00)
				z[2] = pk_mul(pk_add_A2(t1, t1), a2);       // ((k11-k00) a2, (k00+k11) a2)
				z[6] = t0;
				t0 = pk_add(z[5], z[1]);
				z[1] = pk_add_A3(z[1], z[5]);               // (z3 - z11, z10 - z2)
				z[5] = t0;
				t0 = pk_add(z[4], z[0]);
				t1 = pk_add_A4(z[0], z[4]);                 // (k11, k00)
				z[0] = pk_mul(pk_add_A5(t1, t1), a2);       // ((k00-k11) a2, (k00+k11) a2)
				z[4] = t0;
#pragma unroll
				for (int b = 4; b >= 0; b -= 4) { // imdct.rs:202-232 on w[0..8) = z[b..b+4)
					const float2_t A = pk_add(z[b + 3], z[b + 1]);  // (y1, y0)
					const float2_t Bm = pk_sub(z[b + 3], z[b + 1]); // (k11, k00)
					const float2_t Cc = pk_add(z[b + 2], z[b]);     // (y3, y2)
					const float2_t Dm = pk_sub(z[b + 2], z[b]);     // (k33, k22)
					z[b + 3] = pk_add(A, Cc);
					z[b + 2] = pk_sub(A, Cc);
					z[b + 1] = pk_add_A2(Bm, Dm);                   // (k11 - k22, k00 + k33)
					z[b] = pk_add_A6(Bm, Dm);                       // (k11 + k22, k00 - k33)
				}
			}
	}
	// ---- T4: layout D -> bit-reverse gather.  slot(p) = (p & ~127) | ((p & 3) << 5) | ((p >> 2) & 31)
	{
		const uint32_t base = 128u * (lane >> 4) + 2u * (lane & 15u);
#pragma unroll
		for (int c = 0; c < 2; c++)
			if (c < NCH) {
#pragma unroll
				for (int zz = 0; zz < 8; zz++) {
					const uint32_t slot = base + 32u * (zz & 3) + (zz >> 2);
					*reinterpret_cast<float2_t *>(scrb + 4096 * c + 8u * slot) = Z[c][zz];
				}
			}
	}
	lds_fence();
	// rho = rev6(lane); s0 = slot(2 rho): pairs 2v, 2v+256, 255-2v, 511-2v of m' = 2 lane + c2
	const uint32_t rho = __builtin_bitreverse32(lane) >> 26;
	const uint32_t s0 = ((rho & 1u) << 6) | (rho >> 1);
#pragma unroll
	for (int c2 = 0; c2 < 2; c2++) {
		const float4_t Cq = lds4(img + F.off.c4, 16u * (64u * c2 + lane));
		const float4_t Bl = lds4(img + F.off.b_lo, 16u * (64u * c2 + lane));
		const float4_t Bh = lds4(img + F.off.b_hi, 16u * (64u * c2 + lane));
		const uint32_t sa = 128u * c2 + s0; // slot of pair 2v
		const uint32_t sb = 255u - sa;      // slot of pair 255 - 2v
#pragma unroll
		for (int c = 0; c < 2; c++)
			if (c < NCH) {
				const char *sc = scrb + 4096 * c;
				const float2_t pq = lds2(sc, 8u * sa), pq256 = lds2(sc, 8u * (sa + 256u));   // (E3,E2), (E1,E0)
				const float2_t p255 = lds2(sc, 8u * sb), p511 = lds2(sc, 8u * (sb + 256u));  // (D3,D2), (D1,D0)
				float2_t Dn1, En1, Dn2, En2;
				step7_block(p511, pq, float2_t{Cq.x, Cq.y}, Dn1, En1);    // (D1',D0'), (E3',E2')
				step7_block(p255, pq256, float2_t{Cq.z, Cq.w}, Dn2, En2); // (D3',D2'), (E1',E0')
				// step 8 (imdct.rs:618-657): q = 511-2m', 510-2m', 1+2m', 2m'
				R[c][c2][0] = step8(Dn1, float2_t{Bl.x, Bl.y});
				R[c][c2][1] = step8(Dn2, float2_t{Bl.z, Bl.w});
				R[c][c2][2] = step8(En2, float2_t{Bh.x, Bh.y});
				R[c][c2][3] = step8(En1, float2_t{Bh.z, Bh.w});
			}
	}
	lds_fence();
	// ---- publish the un-windowed right half for the successor wave (own scratch, [channel][c2][lane] float4)
#pragma unroll
	for (int c = 0; c < 2; c++)
		if (c < NCH) {
#pragma unroll
			for (int c2 = 0; c2 < 2; c2++)
				*reinterpret_cast<float4_t *>(scrb + 4096 * c + 16u * (64u * c2 + lane)) =
					float4_t{R[c][c2][0].y, R[c][c2][1].y, R[c][c2][2].y, R[c][c2][3].y};
		}
}

// ---- phase 2: overlap-add with the predecessor's right half, sample conversion, stores, state hand-over
template <int NCH, int FMT, bool RIGHT_ONLY>
__device__ __forceinline__ void long_phase2(const LwDevTables &T, const LwBatchDev &B, const LwFastArgs &F, const char *img,
		float *scr, uint32_t lane, uint32_t wave, uint32_t item, const LwPacketRec &rec, const LwFastUnit &un,
		float2_t (&R)[2][2][4])
{
	const int chn[2] = {un.ch_a, un.ch_b >= 0 ? un.ch_b : un.ch_a};
	// pb at this lane's q positions: group 0 = q in [4 lane, +4), group 1 = q in [508 - 4 lane, +4) (ascending)
#define LW_PB_LO0(c) float4_t{R[c][0][3].y, R[c][0][2].y, R[c][1][3].y, R[c][1][2].y}
#define LW_PB_LO1(c) float4_t{R[c][1][1].y, R[c][1][0].y, R[c][0][1].y, R[c][0][0].y}
	if (RIGHT_ONLY) {
		const uint32_t hs = F.items[item].halo;
#pragma unroll
		for (int c = 0; c < 2; c++)
			if (c < NCH) {
				float *dst = F.halo + ((size_t)hs * T.ch + chn[c]) * 512u;
				*reinterpret_cast<float4_t *>(dst + 4u * lane) = LW_PB_LO0(c);
				*reinterpret_cast<float4_t *>(dst + 508u - 4u * lane) = LW_PB_LO1(c);
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
		const float2_t k32768 = float2_t{32768.0f, 32768.0f};
#pragma unroll
		for (int c = 0; c < 2; c++)
			if (c < NCH) {
				float2_t pp[2][2]; // previous pb: pp[c2][0] = (k=0, k=1), pp[c2][1] = (k=2, k=3)
				if (from_lds) {
#pragma unroll
					for (int c2 = 0; c2 < 2; c2++) {
						const float4_t v = *reinterpret_cast<const float4_t *>(
								reinterpret_cast<const char *>(pscr) + 4096 * c + 16u * (64u * c2 + lane));
						pp[c2][0] = float2_t{v.x, v.y};
						pp[c2][1] = float2_t{v.z, v.w};
					}
				} else {
					const float4_t g0 = *reinterpret_cast<const float4_t *>(gsrc[c] + 4u * lane);        // q = 4l .. 4l+3
					const float4_t g1 = *reinterpret_cast<const float4_t *>(gsrc[c] + 508u - 4u * lane); // q = 508-4l ..
					pp[0][1] = float2_t{g0.y, g0.x}; // (c2=0: k=2 -> q=4l+1, k=3 -> q=4l)
					pp[1][1] = float2_t{g0.w, g0.z}; // (c2=1: k=2 -> 4l+3, k=3 -> 4l+2)
					pp[1][0] = float2_t{g1.y, g1.x}; // (c2=1: k=0 -> 509-4l, k=1 -> 508-4l)
					pp[0][0] = float2_t{g1.w, g1.z}; // (c2=0: k=0 -> 511-4l, k=1 -> 510-4l)
				}
				// ---- window + overlap-add (audio.rs:1116-1118): (out[q], out[1023-q]) = (pa s[q] + pb' s[r], -pa s[r] + pb' s[q])
				float2_t O[2][4];
#pragma unroll
				for (int c2 = 0; c2 < 2; c2++) {
					const float4_t w0 = lds4(img + F.off.win, 32u * (64u * c2 + lane));
					const float4_t w1 = lds4(img + F.off.win, 32u * (64u * c2 + lane) + 16u);
					const float2_t S2[4] = {float2_t{w0.x, w0.y}, float2_t{w0.z, w0.w}, float2_t{w1.x, w1.y}, float2_t{w1.z, w1.w}};
#pragma unroll
					for (int k = 0; k < 4; k++) {
						const float2_t o1 = pk_mul_M4(R[c][c2][k], S2[k]); // (pa s[q], -pa s[r])
						const float2_t o2 = (k & 1) ? pk_mul_M12hi(pp[c2][k >> 1], S2[k]) : pk_mul_M12lo(pp[c2][k >> 1], S2[k]);
						O[c2][k] = pk_add(o1, o2);
					}
				}
				// positions: [4l..4l+3] = .x of (0,3) (0,2) (1,3) (1,2); [508-4l..] = .x of (1,1) (1,0) (0,1) (0,0)
				//            [512+4l..] = .y of (0,0) (0,1) (1,0) (1,1); [1020-4l..] = .y of (1,2) (1,3) (0,2) (0,3)
				const uint32_t p0 = 4u * lane, p1 = 508u - 4u * lane, p2 = 512u + 4u * lane, p3 = 1020u - 4u * lane;
				if (FMT == LW_OUT_F32_PLANAR) {
					float *o = reinterpret_cast<float *>(F.out) + rec.out_off + (uint32_t)chn[c] * 1024u;
					*reinterpret_cast<float4_t *>(o + p0) = float4_t{O[0][3].x, O[0][2].x, O[1][3].x, O[1][2].x};
					*reinterpret_cast<float4_t *>(o + p1) = float4_t{O[1][1].x, O[1][0].x, O[0][1].x, O[0][0].x};
					*reinterpret_cast<float4_t *>(o + p2) = float4_t{O[0][0].y, O[0][1].y, O[1][0].y, O[1][1].y};
					*reinterpret_cast<float4_t *>(o + p3) = float4_t{O[1][2].y, O[1][3].y, O[0][2].y, O[0][3].y};
				} else {
					// samples.rs:92-103: x*32768, truncate toward zero (v_cvt_i32_f32: saturating, NaN -> 0), clamp to
					// i16 by the saturating pack v_cvt_pk_i16_i32 -- equal to the reference's compare/clamp/`as i16`
					int iq[2][4], im[2][4];
#pragma unroll
					for (int c2 = 0; c2 < 2; c2++)
#pragma unroll
						for (int k = 0; k < 4; k++) {
							const float2_t t = pk_mul(O[c2][k], k32768);
							iq[c2][k] = (int)t.x;
							im[c2][k] = (int)t.y;
						}
					if (FMT == LW_OUT_I16_PLANAR) {
						typedef short short2_t __attribute__((ext_vector_type(2)));
						union {
							short2_t s;
							uint32_t u;
						} a, b;
						int16_t *o = reinterpret_cast<int16_t *>(F.out) + rec.out_off + (uint32_t)chn[c] * 1024u;
						a.s = __builtin_amdgcn_cvt_pk_i16(iq[0][3], iq[0][2]);
						b.s = __builtin_amdgcn_cvt_pk_i16(iq[1][3], iq[1][2]);
						*reinterpret_cast<uint2_t *>(o + p0) = uint2_t{a.u, b.u};
						a.s = __builtin_amdgcn_cvt_pk_i16(iq[1][1], iq[1][0]);
						b.s = __builtin_amdgcn_cvt_pk_i16(iq[0][1], iq[0][0]);
						*reinterpret_cast<uint2_t *>(o + p1) = uint2_t{a.u, b.u};
						a.s = __builtin_amdgcn_cvt_pk_i16(im[0][0], im[0][1]);
						b.s = __builtin_amdgcn_cvt_pk_i16(im[1][0], im[1][1]);
						*reinterpret_cast<uint2_t *>(o + p2) = uint2_t{a.u, b.u};
						a.s = __builtin_amdgcn_cvt_pk_i16(im[1][2], im[1][3]);
						b.s = __builtin_amdgcn_cvt_pk_i16(im[0][2], im[0][3]);
						*reinterpret_cast<uint2_t *>(o + p3) = uint2_t{a.u, b.u};
					} else {
						int16_t *o = reinterpret_cast<int16_t *>(F.out) + rec.out_off + (uint32_t)chn[c];
						const int g0[4] = {iq[0][3], iq[0][2], iq[1][3], iq[1][2]}, g1[4] = {iq[1][1], iq[1][0], iq[0][1], iq[0][0]};
						const int g2[4] = {im[0][0], im[0][1], im[1][0], im[1][1]}, g3[4] = {im[1][2], im[1][3], im[0][2], im[0][3]};
#pragma unroll
						for (int i = 0; i < 4; i++) {
							o[(p0 + i) * T.ch] = (int16_t)min(max(g0[i], -32768), 32767);
							o[(p1 + i) * T.ch] = (int16_t)min(max(g1[i], -32768), 32767);
							o[(p2 + i) * T.ch] = (int16_t)min(max(g2[i], -32768), 32767);
							o[(p3 + i) * T.ch] = (int16_t)min(max(g3[i], -32768), 32767);
						}
					}
				}
			}
	}
	// ---- raw right half to the stream's state slot and/or to the td block a generic successor reads
	const bool to_state = rec.state_out >= 0, to_td = (rec.flags & LW_RF_WRITE_TD) != 0;
	if (to_state || to_td) {
#pragma unroll
		for (int c = 0; c < 2; c++)
			if (c < NCH) {
				const float4_t lo0 = LW_PB_LO0(c); // q = 4l .. 4l+3
				const float4_t lo1 = LW_PB_LO1(c); // q = 508-4l .. 511-4l
				const float4_t hi0 = float4_t{lo1.w, lo1.z, lo1.y, lo1.x}; // 1023-q for q = 511-4l .. 508-4l
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
#undef LW_PB_LO0
#undef LW_PB_LO1
}

template <int FMT, bool RIGHT_ONLY>
__global__ void __launch_bounds__(LW_WG) k_long(LwDevTables T, LwBatchDev B, LwFastArgs F)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t g = blockIdx.x * F.items_per_wg * F.n_units + wave;
	const uint32_t item = g / F.n_units, uidx = g - item * F.n_units;
	const bool valid = wave < F.items_per_wg * F.n_units && item < F.n_items;

	// ---- issue this wave's residue loads first (coalesced float4, lane holds groups m = 64x + lane) ...
	LwPacketRec rec{};
	LwFastUnit un{};
	float4_t r[2][4];
	if (valid) {
		rec = B.recs[F.items[item].pkt];
		un = F.units[uidx];
		const float4_t *s0 = reinterpret_cast<const float4_t *>(B.residue + rec.res_off + (uint32_t)un.ch_a * 1024u);
#pragma unroll
		for (int x = 0; x < 4; x++)
			r[0][x] = s0[64 * x + lane];
		if (un.ch_b >= 0) {
			const float4_t *s1 = reinterpret_cast<const float4_t *>(B.residue + rec.res_off + (uint32_t)un.ch_b * 1024u);
#pragma unroll
			for (int x = 0; x < 4; x++)
				r[1][x] = s1[64 * x + lane];
		}
	}
	// ---- ... then stage the table image while they are in flight
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
	float2_t R[2][2][4]; // [channel][c2][k] = (pa, pb) at q_k(m' = 2 lane + c2): un-windowed left / right halves
	const bool two = un.ch_b >= 0;
	if (valid) {
		if (two)
			long_phase1<2>(T, B, F, img, scrb, lane, rec, un, r, R);
		else
			long_phase1<1>(T, B, F, img, scrb, lane, rec, un, r, R);
	}
	__syncthreads();
	if (!valid)
		return;
	if (two)
		long_phase2<2, FMT, RIGHT_ONLY>(T, B, F, img, scr, lane, wave, item, rec, un, R);
	else
		long_phase2<1, FMT, RIGHT_ONLY>(T, B, F, img, scr, lane, wave, item, rec, un, R);
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
		F.items_per_wg = 1; // spread the few halo packets over the whole chip: one packet per workgroup
		const uint32_t grid = L.n_halo_items;
		hipLaunchKernelGGL((k_long<LW_OUT_I16_PLANAR, true>), dim3(grid), dim3(LW_WG), lds, st, T, B, F);
	}
	if (L.n_items) {
		F.items = L.d_items;
		F.n_items = L.n_items;
		F.items_per_wg = per_wg;
		const uint32_t grid = (L.n_items + per_wg - 1) / per_wg;
		if (fmt == LW_OUT_I16_PLANAR)
			hipLaunchKernelGGL((k_long<LW_OUT_I16_PLANAR, false>), dim3(grid), dim3(LW_WG), lds, st, T, B, F);
		else if (fmt == LW_OUT_I16_INTERLEAVED)
			hipLaunchKernelGGL((k_long<LW_OUT_I16_INTERLEAVED, false>), dim3(grid), dim3(LW_WG), lds, st, T, B, F);
		else
			hipLaunchKernelGGL((k_long<LW_OUT_F32_PLANAR, false>), dim3(grid), dim3(LW_WG), lds, st, T, B, F);
	}
}
