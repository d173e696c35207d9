// k_big<fmt, BS>: the long blocks of streams with blocksize_1 = 12 / 13 (4096 / 8192 points; header.rs:236-247 allows them, libvorbis
// never writes them) whose window slopes are both long -- floor curve, inverse coupling, floor x residue, IMDCT, window /
// overlap-add, sample conversion and state in ONE kernel, as k_long / k_short<L> do it for the smaller sizes (before: the generic
// kernels, ~16 barrier-separated radix-2 stages in LDS + a round trip of the time-domain block through HBM).
//
//   reference                                   here
//   src/audio.rs:526-555, :503-524 (floor 1)    floor_entry() / floor_bin(): a segment table per static interval, two FMAs per bin
//   src/audio.rs:762-777 (inverse coupling)     decouple()
//   src/imdct.rs:291-659 (inverse MDCT)         passes P0 .. P3 + E below
//   src/audio.rs:1082-1154 (overlap-add, state) phase E of the kernel
//   src/samples.rs:92-103 (conversion)          store_pair()
//   src/header_cached.rs:104-108 (bit reversal) computed (v_bfrev_b32) in phase E
//
// One workgroup of T = n / 32 threads (2 / 4 waves) works through the `passes` consecutive slots of one task (the block kernel's
// slot descriptors, lw_fast.hpp: the planner of lw_batch.cpp places consecutive blocks of a stream in consecutive slots and a
// recomputed predecessor, LW_SS_HALO, in front of a run that starts inside a stream), one channel at a time.  The n/4 complex
// pairs of the transform (pair q = floats 2q, 2q + 1 of the reference's work array) live in LDS between the passes, addressed by
// q' = n/4 - 1 - q and padded by one pair per eight; a thread holds eight pairs per pass and runs up to three butterfly stages
// on them in registers.  Every stage of distance D is the reference's butterfly (hi = q', lo = q' + D, twiddle A[r n / (2D)],
// r = q' mod 2D): step 2 is the stage of distance n/8.
//   P0  step 1 of j = t + T i (i < 4): the pairs q' = j and n/8 + j, the stage of distance n/8 and (n = 8192) of distance 512
//   P1  distances 256, 128, 64      q' = 512 (t / 64) + t % 64 + 64 i
//   P2  distances 32, 16, 8         q' = 64 (t / 8) + t % 8 + 8 i
//   P3  the fused last three stages on q' = 8 t + i (with the padding: conflict-free 8-byte accesses)
//   E   m = t + T e (e < 2): bit-reverse gather, step 7, step 8 -> (pa, pb) at p = 2m, 2m + 1, n/4 - 2 - 2m, n/4 - 1 - 2m; the
//       samples n/4 - 1 - p and n/4 + p need the predecessor's pb(p): the SAME thread computed it in the previous slot, so
//       the hand-over between consecutive blocks of a stream never leaves the registers
// Arithmetic: single f32 operations in the reference's order (-ffp-contract=off); the thread-level numpy model of exactly this
// data movement (tests/big_model.py) reproduces the oracle bit for bit on the CPU.
#include "lw_fast.hpp"
#include "lw_kernels.hpp"

typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));

struct LwBigArgs {
	const float *residue;
	const uint16_t *floors;
	const LwShortSlot *slots;
	float *state, *td;
	void *out;
	const float *A, *Bt, *C, *window, *inv_db;
	const uint32_t *bitrev; // (not read: phase E computes the bit reversal)
	const uint16_t *floor_x;
	uint32_t n_units, ch, fstride, state_stride, state_chan_stride, passes;
	uint32_t n_wg; // tasks x units = workgroups
	uint32_t fl_of[LW_FAST_MAX_FLOORS]; // floor index (header order) of each staged floor slot of the units
	LwFastUnit units[LW_FAST_WAVES];
};

namespace {

__device__ __forceinline__ uint32_t pad8(uint32_t q) { return q + (q >> 3); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a release fence over every address space: it would
// wait for the block's sample stores (and for the residue loads of the next block) at each of the ~13 barriers of a slot.
// Nothing the waves of a workgroup exchange goes through global memory.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// imdct.rs:445-477 on pairs (x = even float, y = odd float)
__device__ __forceinline__ void bfly2(float2_t &H, float2_t &L, const float2_t tw)
{
	const float k00 = H.y - L.y, k01 = H.x - L.x;
	H.x = H.x + L.x;
	H.y = H.y + L.y;
	L.y = k00 * tw.x - k01 * tw.y;
	L.x = k01 * tw.x + k00 * tw.y;
}

__device__ __forceinline__ float2_t ld2(const float *p) { return *reinterpret_cast<const float2_t *>(p); }
__device__ __forceinline__ float4_t ld4(const float *p) { return *reinterpret_cast<const float4_t *>(p); }

// three stages of distances D, D/2, D/4 on the pairs q' = q0 + (D/4) i; twiddles of the thread (loaded once per workgroup):
// ta[i] = A[(off + (D/4) i) N / (2D) ..], tb[i] = A[(off + (D/4) i) N / D ..], tc = A[off 2N / D ..], off = t % (D/4)
template <uint32_t D>
__device__ __forceinline__ void pass3(float2_t *V, uint32_t t, const float2_t (&ta)[4], const float2_t (&tb)[2], const float2_t tc)
{
	constexpr uint32_t STEP = D / 4;
	const uint32_t q0 = (t / STEP) * (2u * D) + t % STEP;
	float2_t R[8];
#pragma unroll
	for (uint32_t i = 0; i < 8; i++)
		R[i] = V[pad8(q0 + STEP * i)];
#pragma unroll
	for (uint32_t i = 0; i < 4; i++)
		bfly2(R[i], R[i + 4], ta[i]);
#pragma unroll
	for (uint32_t i = 0; i < 2; i++) {
		bfly2(R[i], R[i + 2], tb[i]);
		bfly2(R[4 + i], R[6 + i], tb[i]);
	}
#pragma unroll
	for (uint32_t i = 0; i < 8; i += 2)
		bfly2(R[i], R[i + 1], tc);
#pragma unroll
	for (uint32_t i = 0; i < 8; i++)
		V[pad8(q0 + STEP * i)] = R[i];
}

// imdct.rs:202-232 on w[0..8)
__device__ __forceinline__ void iter54(float *w)
{
	const float k00 = w[7] - w[3];
	const float y0 = w[7] + w[3];
	const float y2 = w[5] + w[1];
	const float k22 = w[5] - w[1];
	const float k33 = w[4] - w[0];
	const float k11 = w[6] - w[2];
	const float y1 = w[6] + w[2];
	const float y3 = w[4] + w[0];
	w[7] = y0 + y2;
	w[5] = y0 - y2;
	w[3] = k00 + k33;
	w[1] = k00 - k33;
	w[6] = y1 + y3;
	w[4] = y1 - y3;
	w[2] = k11 - k22;
	w[0] = k11 + k22;
}

// imdct.rs:234-288 on z[0..16) (z[15 - k] is the reference's z![-k])
__device__ __forceinline__ void last3(float *z, const float a2)
{
	float k00, k11;
	k00 = z[15] - z[7];
	k11 = z[14] - z[6];
	z[15] = z[15] + z[7];
	z[14] = z[14] + z[6];
	z[7] = k00;
	z[6] = k11;
	k00 = z[13] - z[5];
	k11 = z[12] - z[4];
	z[13] = z[13] + z[5];
	z[12] = z[12] + z[4];
	z[5] = (k00 + k11) * a2;
	z[4] = (k11 - k00) * a2;
	k00 = z[3] - z[11];
	k11 = z[10] - z[2];
	z[11] = z[11] + z[3];
	z[10] = z[10] + z[2];
	z[3] = k11;
	z[2] = k00;
	k00 = z[1] - z[9];
	k11 = z[8] - z[0];
	z[9] = z[9] + z[1];
	z[8] = z[8] + z[0];
	z[1] = (k00 + k11) * a2;
	z[0] = (k00 - k11) * a2;
	iter54(z + 8);
	iter54(z);
}

// audio.rs:762-777.  With c = (m > 0), d = (a > 0) the reference's four cases are v = m + ((c == d) ? -a : a) and
// (new_m, new_a) = d ? (m, v) : (v, m)   (x - y and x + (-y) are the same operation): no branches
__device__ __forceinline__ void decouple(float &m, float &a)
{
	const bool c = m > 0.0f, d = a > 0.0f;
	const float v = m + ((c == d) ? -a : a);
	const float nm = d ? m : v, na = d ? v : m;
	m = nm;
	a = na;
}

// Floor curve (audio.rs:526-555), the way k_long does it (lw_kernels_long.hip: floor_table / floor_bin): one 16-byte entry
// {dy, c0, 1/adx, w} per STATIC interval of the floor configuration (between consecutive posts in ascending x), describing the
// ACTIVE segment that covers it; the floor value of bin k is inverse_db[y(k)] with
//     y(k) = ((bits(fma(fma(k, dy, c0), 1/adx, w)) & 0x7fc) >> 2) - 1,       w = 2^21 + 1 + y_base
// -- two fused multiply-adds and one AND per bin.  render_line (audio.rs:503-524) is y = y0 + trunc((k - x0) dy / adx); both
// signs are written as a FLOOR of something non-negative:
//     dy >= 0:  y = y0 + floor(((k - x0) dy + 1/2) / adx)            c0 = 1/2 - x0 dy - adx/8,      y_base = y0
//     dy <  0:  y = y1 + floor(((x1 - k) |dy| + adx - 1/2) / adx)    c0 = 7 adx/8 - 1/2 - x1 dy,    y_base = y1
// The numerators are integers + 1/2, so the quotient's fraction lies in [1/(2 adx), 1 - 1/(2 adx)]; the - adx/8 moves it to
// [-1/8 + 1/(2 adx), 7/8 - 1/(2 adx)], which rounds to a multiple of 1/4 (one ulp in [2^21, 2^22)) in [0, 3/4]: the integer part
// is never touched.  Both inner sums are exact in f32 up to adx = 4096 (multiples of 1/8 below 2^21); the outer product's error
// is below 256 * 2^-23 < 1/(2 * 4096).  tests/test_big_model.py checks every adx <= 4096 with every offset inside it, with the
// reciprocal one ulp off either way, as v_rcp_f32 may be.
#define LW_BIG_FLOOR_W0 2097153.0f // 2^21 + 1
#define LW_BIG_FLOOR_MASK 0x7fcu

// posts: x | y << 16 of the floor's posts in ascending x, `active`: bit s = post s is used by this packet (bit 64 in `a64`); the
// entry of the static interval behind post s: from the last active post at or below it to the next active one above it
__device__ __forceinline__ float4_t floor_entry(const uint32_t *posts, unsigned long long active, bool a64, uint32_t s)
{
	const unsigned long long lowmask = s >= 63u ? ~0ull : (2ull << s) - 1ull;
	const unsigned long long below = active & lowmask, above_m = s >= 63u ? 0ull : active & ~lowmask;
	const uint32_t lo = (s == 64u && a64) ? 64u : 63u - (uint32_t)__builtin_clzll(below); // (post 0 is always active)
	const bool above = above_m != 0ull || (a64 && s < 64u); // otherwise: flat to n/2 behind the last active post (audio.rs:546-548)
	const uint32_t hi = above_m ? (uint32_t)__builtin_ctzll(above_m) : 64u;
	const uint32_t pl = posts[lo], ph = posts[above ? hi : lo];
	const float xlo = (float)(pl & 0xffffu), xhi = (float)(ph & 0xffffu);
	const int ylo = (int)((pl >> 16) & 0xffu), yhi = (int)((ph >> 16) & 0xffu);
	const float dy = (float)(yhi - ylo);
	const float adx = above ? xhi - xlo : 1.0f;
	const bool down = yhi < ylo;
	float4_t ent;
	ent.x = dy;
	ent.y = down ? (0.875f * adx - 0.5f) - xhi * dy : (0.5f - 0.125f * adx) - xlo * dy;
	ent.z = above ? __builtin_amdgcn_rcpf(adx) : 1.0f;
	ent.w = (float)(down ? yhi : ylo) + LW_BIG_FLOOR_W0;
	return ent;
}

__device__ __forceinline__ float floor_bin(const float *inv_s, float kf, float4_t ent)
{
	const float t = __builtin_fmaf(__builtin_fmaf(kf, ent.x, ent.y), ent.z, ent.w);
	return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(inv_s) + ((__float_as_uint(t) & LW_BIG_FLOOR_MASK) - 4u));
}

template <int FMT>
__device__ __forceinline__ void store_pair(void *out, uint32_t elem0, uint32_t pos, uint32_t stride, float a, float b)
{
	if (FMT == LW_OUT_F32_PLANAR) {
		*reinterpret_cast<float2_t *>(reinterpret_cast<float *>(out) + elem0 + pos) = float2_t{a, b};
	} else {
		// samples.rs:92-103: x * 32768, truncate toward zero (v_cvt_i32_f32: saturating, NaN -> 0), clamp to i16 by the saturating
		// pack (clamping the integer = clamping the float first: the bounds are integers)
		typedef short short2_t __attribute__((ext_vector_type(2)));
		union {
			short2_t s;
			uint32_t u;
		} v;
		v.s = __builtin_amdgcn_cvt_pk_i16((int)(a * 32768.0f), (int)(b * 32768.0f));
		int16_t *o = reinterpret_cast<int16_t *>(out) + elem0;
		if (FMT == LW_OUT_I16_PLANAR) {
			*reinterpret_cast<uint32_t *>(o + pos) = v.u;
		} else {
			o[pos * stride] = v.s.x;
			o[(pos + 1u) * stride] = v.s.y;
		}
	}
}

} // namespace

// the fields of a slot descriptor the kernel uses (the same for every thread: kept in scalar registers)
struct BigSlot {
	uint32_t res_off, floor_off, out_off, prev_arg, prev_stride, kind, prev_kind, flags;
	int32_t state_out;
};

__device__ __forceinline__ BigSlot load_slot(const LwShortSlot *p)
{
	const uint4 *sp = reinterpret_cast<const uint4 *>(p);
	const uint4 d0 = sp[0], d1 = sp[1], d2 = sp[2];
	BigSlot s;
	s.res_off = __builtin_amdgcn_readfirstlane(d0.x);
	s.floor_off = __builtin_amdgcn_readfirstlane(d0.y);
	s.out_off = __builtin_amdgcn_readfirstlane(d0.z);
	s.prev_arg = __builtin_amdgcn_readfirstlane(d0.w);
	s.state_out = (int32_t)__builtin_amdgcn_readfirstlane(d1.x);
	const uint32_t x = __builtin_amdgcn_readfirstlane(d2.x);
	s.prev_stride = x & 0xffffu;
	s.kind = (x >> 16) & 0xffu;
	s.prev_kind = x >> 24;
	s.flags = __builtin_amdgcn_readfirstlane(d2.y);
	return s;
}

// Three waves per SIMD (168 registers: no spills with the table values re-read where they are used) = six / three workgroups per CU
template <int FMT, int BS>
__global__ void __launch_bounds__(1 << (BS - 5)) __attribute__((amdgpu_waves_per_eu(3, 3))) k_big(LwBigArgs F)
{
	constexpr uint32_t n = 1u << BS, n2 = n / 2, n4 = n / 4, n8 = n / 8, T = n / 32, NPAD = n4 + n4 / 8;
	__shared__ __attribute__((aligned(16))) float U[n2];         // floor x residue of the channel in work
	__shared__ __attribute__((aligned(16))) float2_t V[NPAD];    // the transform's pairs
	__shared__ __attribute__((aligned(16))) float4_t tab[2][68]; // floor segment entry of every static interval, per channel
	__shared__ float inv_s[256];
	__shared__ uint32_t posts[2][68];       // x | y << 16 of the floor posts, per channel
	__shared__ unsigned long long amask[2]; // bit s: post s is active (posts 0 .. 63: a ballot of the first wave)
	__shared__ int unused_s[2], a64_s[2];   // the floor is unused / post 64 is active
	// The tables of the block size (header_cached.rs:34-110; 28 / 56 KB) stay in L2.  (Staged in LDS per workgroup they cost the
	// residency they were meant to pay for: 78.8 us per 4096 blocks of 4096 points instead of 77.7; kept in ~100 registers per
	// thread the rest of the code was serialised or spilled: 133 us -- profiles/r04_k_big_variants.txt.)
	const float *const At = F.A, *const Bs = F.Bt, *const Cs = F.C, *const Ws = F.window;
	const uint32_t t = threadIdx.x;
	const uint32_t task = blockIdx.x / F.n_units, uidx = blockIdx.x - task * F.n_units;
	const LwFastUnit un = F.units[uidx];
	const bool two = un.ch_b >= 0;
	const uint32_t chn[2] = {(uint32_t)un.ch_a, (uint32_t)(two ? un.ch_b : un.ch_a)};
	const uint32_t nch = two ? 2u : 1u;
	const uint32_t Fp[2] = {un.F_a, two ? un.F_b : 0u};
	for (uint32_t i = t; i < 256u; i += T)
		inv_s[i] = F.inv_db[i];
	// ---- once per workgroup: the x of the thread's floor post and the static floor interval (largest post index s with
	// x[s] <= k among ALL posts of the floor configuration) of each of its 16 bins, one byte each
	uint32_t my_x[2] = {0u, 0u}, sid[2][4];
#pragma unroll
	for (uint32_t c = 0; c < 2; c++) {
		if (c < nch && t < Fp[c]) {
			my_x[c] = F.floor_x[F.fl_of[c == 0 ? un.floor_a : un.floor_b] * LW_XSTRIDE + t];
			posts[c][t] = my_x[c]; // (through LDS: every thread walks all the posts, without a chain of dependent HBM loads)
		}
	}
	lds_barrier();
#pragma unroll
	for (uint32_t c = 0; c < 2; c++) {
		uint32_t cnt[4] = {1u, 1u, 1u, 1u}; // posts at or below the first bin of each group (post 0 has x = 0)
		const uint32_t Fc = c < nch ? Fp[c] : 0u;
		for (uint32_t s_ = 1; s_ < Fc; s_++) {
			const uint32_t x = posts[c][s_];
#pragma unroll
			for (uint32_t i = 0; i < 4; i++)
				cnt[i] += x <= 4u * (t + T * i) ? 1u : 0u;
		}
#pragma unroll
		for (uint32_t i = 0; i < 4; i++) {
			const uint32_t k0 = 4u * (t + T * i);
			uint32_t lo = cnt[i] - 1u, w = lo;
#pragma unroll
			for (uint32_t j = 1; j < 4; j++) { // (the x are distinct integers: at most one post per bin)
				if (lo + 1u < Fc && posts[c][lo + 1u] <= k0 + j)
					lo++;
				w |= lo << (8u * j);
			}
			sid[c][i] = w;
		}
	}
	lds_barrier();
	// the previous block's right part as this thread needs it: pb(p) at p = 2m, 2m + 1, n/4 - 2 - 2m, n/4 - 1 - 2m, m = t + T e
	float pbp[2][2][4];
#pragma unroll
	for (int c = 0; c < 2; c++)
#pragma unroll
		for (int e = 0; e < 2; e++)
#pragma unroll
			for (int k = 0; k < 4; k++)
				pbp[c][e][k] = 0.0f;
	const LwShortSlot *slots = F.slots + (size_t)task * F.passes;
	BigSlot nxt = load_slot(slots);
	for (uint32_t pass = 0; pass < F.passes; pass++) {
		const BigSlot cur = nxt;
		nxt = load_slot(slots + (pass + 1u < F.passes ? pass + 1u : pass)); // (in scalar registers by the time the next slot starts)
		const bool block = cur.kind == LW_SS_BLOCK || cur.kind == LW_SS_HALO; // (the same for every thread of the workgroup)
		// ---- HBM loads, all at once: residues, floor records, the stored right part in front of the first block of a run
		float4_t r[2][4];
		uint32_t my_e[2] = {0u, 0u};
#pragma unroll
		for (uint32_t c = 0; c < 2; c++) {
#pragma unroll
			for (uint32_t i = 0; i < 4; i++)
				r[c][i] = float4_t{0.0f, 0.0f, 0.0f, 0.0f};
			if (c >= nch || !block)
				continue;
			const float4_t *src = reinterpret_cast<const float4_t *>(F.residue + cur.res_off + chn[c] * n2);
#pragma unroll
			for (uint32_t i = 0; i < 4; i++)
				r[c][i] = __builtin_nontemporal_load(&src[t + T * i]);
			my_e[c] = t < Fp[c] ? (uint32_t)F.floors[cur.floor_off + chn[c] * F.fstride + t] : 0u;
			if (cur.kind == LW_SS_BLOCK && (cur.prev_kind == LW_SP_STATE || cur.prev_kind == LW_SP_TD)) {
				const float *ps = cur.prev_kind == LW_SP_STATE
					? F.state + ((size_t)cur.prev_arg * 2u + ((cur.flags & LW_RF_PARITY_IN) ? 1u : 0u)) * F.state_stride + chn[c] * F.state_chan_stride
					: F.td + cur.prev_arg + chn[c] * cur.prev_stride;
#pragma unroll
				for (uint32_t e = 0; e < 2; e++) { // pb(p) = right part at n/4 - 1 - p
					const uint32_t m = t + T * e;
					const float2_t a = ld2(ps + (n4 - 2u - 2u * m)), b = ld2(ps + 2u * m);
					pbp[c][e][0] = a.y;
					pbp[c][e][1] = a.x;
					pbp[c][e][2] = b.y;
					pbp[c][e][3] = b.x;
				}
			}
		}
		// ---- floor segment table (audio.rs:536-548 walks the active posts in ascending x): one thread per post
#pragma unroll
		for (uint32_t c = 0; c < 2; c++) {
			if (c >= nch || !block)
				continue;
			const bool on = t < Fp[c] && (my_e[c] & LW_POST_ACTIVE) != 0;
			if (t < Fp[c])
				posts[c][t] = my_x[c] | ((my_e[c] & 0xffu) << 16);
			if (t < 64u) {
				const unsigned long long m = __ballot(on);
				if (t == 0u) {
					amask[c] = m;
					unused_s[c] = my_e[c] == LW_FLOOR_UNUSED;
					if (Fp[c] <= 64u)
						a64_s[c] = 0;
				}
			} else if (t == 64u && Fp[c] > 64u) {
				a64_s[c] = on ? 1 : 0;
			}
		}
		lds_barrier();
#pragma unroll
		for (uint32_t c = 0; c < 2; c++) {
			if (c >= nch || !block)
				continue;
			if (t < Fp[c] && !unused_s[c])
				tab[c][t] = floor_entry(posts[c], amask[c], a64_s[c] != 0, t);
		}
		// ---- inverse coupling (audio.rs:762-777): ch_a = magnitude, ch_b = angle
		if (block && two && un.coupled) {
#pragma unroll
			for (uint32_t i = 0; i < 4; i++) {
				float m[4] = {r[0][i].x, r[0][i].y, r[0][i].z, r[0][i].w}, a[4] = {r[1][i].x, r[1][i].y, r[1][i].z, r[1][i].w};
#pragma unroll
				for (int j = 0; j < 4; j++)
					decouple(m[j], a[j]);
				r[0][i] = float4_t{m[0], m[1], m[2], m[3]};
				r[1][i] = float4_t{a[0], a[1], a[2], a[3]};
			}
		}
		lds_barrier();
#pragma unroll
		for (uint32_t c = 0; c < 2; c++) { // (unrolled: everything indexed by the channel stays in registers; a single-channel unit
		                                   // passes the second channel's barriers idle -- a barrier inside a skipped body would hang)
			const bool work = block && c < nch;
			// (the thread index behind an opaque copy, refreshed before every phase's table reads: the factors are then re-read from
			// LDS where they are used instead of being hoisted out of the loop and kept in ~100 registers)
			uint32_t tl = t;
			asm volatile("" : "+v"(tl));
			// ---- spectrum = floor x residue (audio.rs:1035-1037; zero floor of an unused channel :1021-1024) -> U
			if (work) {
				const bool unused = unused_s[c] != 0;
#pragma unroll
				for (uint32_t i = 0; i < 4; i++) {
					const uint32_t k0 = 4u * (tl + T * i);
					float4_t ent[4];
#pragma unroll
					for (uint32_t j = 0; j < 4; j++) // (the four entries first, then the four table values: two LDS round trips per group)
						ent[j] = tab[c][(sid[c][i] >> (8u * j)) & 0xffu];
					float f[4];
#pragma unroll
					for (uint32_t j = 0; j < 4; j++)
						f[j] = unused ? 0.0f : floor_bin(inv_s, (float)(k0 + j), ent[j]);
					const float4_t rr = r[c][i];
					*reinterpret_cast<float4_t *>(U + k0) = float4_t{f[0] * rr.x, f[1] * rr.y, f[2] * rr.z, f[3] * rr.w};
				}
			}
			lds_barrier();
			// ---- P0: step 1 (imdct.rs:337-371) of j = t + T i, the stage of distance n/8 (step 2, :385-430), n = 8192: of distance 512
			if (work) {
				float2_t H[4], Lo[4];
#pragma unroll
				for (uint32_t i = 0; i < 4; i++) {
					const uint32_t j = tl + T * i;
					const float4_t x = *reinterpret_cast<const float4_t *>(U + 4u * j);             // x0 = .x, x2 = .z
					const float4_t y = *reinterpret_cast<const float4_t *>(U + (n2 - 4u - 4u * j)); // u[e] = .y, u[e + 2] = .w
					const float2_t a0 = ld2(At + 2u * j), a1 = ld2(At + n4 + 2u * j);
					H[i] = float2_t{x.x * a0.y + x.z * a0.x, x.x * a0.x - x.z * a0.y};
					const float me2 = -y.w, me0 = -y.y;
					Lo[i] = float2_t{me2 * a1.y + me0 * a1.x, me2 * a1.x - me0 * a1.y};
				}
#pragma unroll
				for (uint32_t i = 0; i < 4; i++)
					bfly2(H[i], Lo[i], ld2(At + 4u * (tl + T * i)));
				if (BS == 13) {
#pragma unroll
					for (uint32_t i = 0; i < 2; i++) {
						const float2_t tw = ld2(At + 8u * (tl + T * i));
						bfly2(H[i], H[i + 2], tw);
						bfly2(Lo[i], Lo[i + 2], tw);
					}
				}
#pragma unroll
				for (uint32_t i = 0; i < 4; i++) {
					V[pad8(tl + T * i)] = H[i];
					V[pad8(n8 + tl + T * i)] = Lo[i];
				}
			}
			lds_barrier();
			asm volatile("" : "+v"(tl));
			if (work) {
				float2_t ta[4], tb[2];
#pragma unroll
				for (uint32_t i = 0; i < 4; i++)
					ta[i] = ld2(At + (tl % 64u + 64u * i) * (n / 512u));
#pragma unroll
				for (uint32_t i = 0; i < 2; i++)
					tb[i] = ld2(At + (tl % 64u + 64u * i) * (n / 256u));
				pass3<256>(V, tl, ta, tb, ld2(At + (tl % 64u) * (n / 128u)));
			}
			lds_barrier();
			asm volatile("" : "+v"(tl));
			if (work) {
				float2_t ta[4], tb[2];
#pragma unroll
				for (uint32_t i = 0; i < 4; i++)
					ta[i] = ld2(At + (tl % 8u + 8u * i) * (n / 64u));
#pragma unroll
				for (uint32_t i = 0; i < 2; i++)
					tb[i] = ld2(At + (tl % 8u + 8u * i) * (n / 32u));
				pass3<32>(V, tl, ta, tb, ld2(At + (tl % 8u) * (n / 16u)));
			}
			lds_barrier();
			// ---- P3: imdct.rs:234-288 on the pairs q' = 8t .. 8t + 7 (z[2k], z[2k + 1] = pair 8t + 7 - k)
			asm volatile("" : "+v"(tl));
			if (work) {
				float z[16];
#pragma unroll
				for (uint32_t k = 0; k < 8; k++) {
					const float2_t v = V[pad8(8u * tl + 7u - k)];
					z[2 * k] = v.x;
					z[2 * k + 1] = v.y;
				}
				last3(z, At[n8]);
#pragma unroll
				for (uint32_t k = 0; k < 8; k++)
					V[pad8(8u * tl + 7u - k)] = float2_t{z[2 * k], z[2 * k + 1]};
			}
			lds_barrier();
			asm volatile("" : "+v"(tl));
			// ---- E: bit-reverse gather (imdct.rs:490-528), step 7 (:533-580), step 8 (:589-658), window / overlap-add, stores
			if (work) {
				const bool samples = cur.kind == LW_SS_BLOCK && cur.prev_kind != LW_SP_NONE;
				const uint32_t elem0 = FMT == LW_OUT_I16_INTERLEAVED ? cur.out_off + chn[c] : cur.out_off + chn[c] * n2;
				const uint32_t stride = FMT == LW_OUT_I16_INTERLEAVED ? F.ch : 1u;
				float *st_dst = (cur.kind == LW_SS_BLOCK && cur.state_out >= 0)
					? F.state + ((size_t)cur.state_out * 2u + ((cur.flags & LW_RF_PARITY_OUT) ? 1u : 0u)) * F.state_stride + chn[c] * F.state_chan_stride
					: nullptr;
				float *td_dst = (cur.kind == LW_SS_BLOCK && (cur.flags & LW_SF_WRITE_TD)) ? F.td + 2u * (size_t)cur.res_off + chn[c] * n + n2 : nullptr;
#pragma unroll
				for (uint32_t e = 0; e < 2; e++) {
					const uint32_t m = tl + T * e, m0 = n / 16u - 1u - m;
					// header_cached.rs:104-108: bitrev[i] = (reverse of i's 32 bits >> (32 - ld n + 3)) << 2; the pair holding floats
					// (k, k + 1) of the reference's array: q' = n/4 - 1 - k/2
					const uint32_t k1 = (__brev(2u * m) >> (35 - BS)) << 1, k1p = (__brev(2u * m + 1u) >> (35 - BS)) << 1;
					const uint32_t k0 = (__brev(2u * m0) >> (35 - BS)) << 1, k0p = (__brev(2u * m0 + 1u) >> (35 - BS)) << 1;
					const float2_t Pa = V[pad8(n4 - 1u - k1)], Pb = V[pad8(n4 - 1u - k1p)];
					const float2_t Pc = V[pad8(n4 - 2u - k0)], Pd = V[pad8(n4 - 2u - k0p)];
					const float4_t Cq = ld4(Cs + 4u * m), Bl = ld4(Bs + 4u * m), Bh = ld4(Bs + (n2 - 4u - 4u * m));
					float ve[4] = {Pb.y, Pb.x, Pa.y, Pa.x}; // v[e .. e + 3], e = n/2 - 4 - 4m
					float vd[4] = {Pd.y, Pd.x, Pc.y, Pc.x}; // v[d .. d + 3], d = 4m
					{
						const float a02 = vd[0] - ve[2], a11 = vd[1] + ve[3];
						const float b0_ = Cq.y * a02 + Cq.x * a11, b1_ = Cq.y * a11 - Cq.x * a02;
						const float b2 = vd[0] + ve[2], b3 = vd[1] - ve[3];
						vd[0] = b2 + b0_;
						vd[1] = b3 + b1_;
						ve[2] = b2 - b0_;
						ve[3] = b1_ - b3;
					}
					{
						const float a02 = vd[2] - ve[0], a11 = vd[3] + ve[1];
						const float b0_ = Cq.w * a02 + Cq.z * a11, b1_ = Cq.w * a11 - Cq.z * a02;
						const float b2 = vd[2] + ve[0], b3 = vd[3] - ve[1];
						vd[2] = b2 + b0_;
						vd[3] = b3 + b1_;
						ve[0] = b2 - b0_;
						ve[1] = b1_ - b3;
					}
					// step 8 at p = 2m, 2m + 1 (B[4m ..]) and n/4 - 2 - 2m, n/4 - 1 - 2m (B[n/2 - 4 - 4m ..])
					float pa[4], pb[4];
					pa[0] = vd[0] * Bl.y - vd[1] * Bl.x;
					pb[0] = (-vd[0]) * Bl.x - vd[1] * Bl.y;
					pa[1] = vd[2] * Bl.w - vd[3] * Bl.z;
					pb[1] = (-vd[2]) * Bl.z - vd[3] * Bl.w;
					pa[2] = ve[0] * Bh.y - ve[1] * Bh.x;
					pb[2] = (-ve[0]) * Bh.x - ve[1] * Bh.y;
					pa[3] = ve[2] * Bh.w - ve[3] * Bh.z;
					pb[3] = (-ve[2]) * Bh.z - ve[3] * Bh.w;
					if (samples) {
						// audio.rs:1116-1118: sample i = cur[i] * w[i] + prev_right[i] * w[n/2 - 1 - i]; with q = n/4 - 1 - p the block's
						// left half is pa(p) at q and -pa(p) at n/2 - 1 - q, the predecessor's right part pb'(p) at both
						const float2_t wA = ld2(Ws + (n4 - 2u - 2u * m)), wB = ld2(Ws + (n4 + 2u * m));
						const float2_t wC = ld2(Ws + 2u * m), wD = ld2(Ws + (n2 - 2u - 2u * m));
						const float *pp = pbp[c][e];
						// p = 2m: q = n/4 - 1 - 2m (w = wA.y, mirror wB.x); p = 2m + 1: q = n/4 - 2 - 2m (wA.x, wB.y)
						const float s0 = (pa[0] * wA.y) + (pp[0] * wB.x), s0m = ((-pa[0]) * wB.x) + (pp[0] * wA.y);
						const float s1 = (pa[1] * wA.x) + (pp[1] * wB.y), s1m = ((-pa[1]) * wB.y) + (pp[1] * wA.x);
						// p = n/4 - 2 - 2m: q = 2m + 1 (wC.y, mirror wD.x); p = n/4 - 1 - 2m: q = 2m (wC.x, wD.y)
						const float s2 = (pa[2] * wC.y) + (pp[2] * wD.x), s2m = ((-pa[2]) * wD.x) + (pp[2] * wC.y);
						const float s3 = (pa[3] * wC.x) + (pp[3] * wD.y), s3m = ((-pa[3]) * wD.y) + (pp[3] * wC.x);
						const uint32_t mt = tl + T * e;
						store_pair<FMT>(F.out, elem0, n4 - 2u - 2u * mt, stride, s1, s0);
						store_pair<FMT>(F.out, elem0, n4 + 2u * mt, stride, s0m, s1m);
						store_pair<FMT>(F.out, elem0, 2u * mt, stride, s3, s2);
						store_pair<FMT>(F.out, elem0, n2 - 2u - 2u * mt, stride, s2m, s3m);
					}
					// the raw right part (audio.rs:1121, :1142-1147): pb(p) at n/4 - 1 - p and mirrored at n/4 + p
#pragma unroll
					for (int w = 0; w < 2; w++) {
						float *dst = w == 0 ? st_dst : td_dst;
						if (dst) {
							const uint32_t mt = tl + T * e;
							*reinterpret_cast<float2_t *>(dst + (n4 - 2u - 2u * mt)) = float2_t{pb[1], pb[0]};
							*reinterpret_cast<float2_t *>(dst + (n4 + 2u * mt)) = float2_t{pb[0], pb[1]};
							*reinterpret_cast<float2_t *>(dst + 2u * mt) = float2_t{pb[3], pb[2]};
							*reinterpret_cast<float2_t *>(dst + (n2 - 2u - 2u * mt)) = float2_t{pb[2], pb[3]};
						}
					}
#pragma unroll
					for (int k = 0; k < 4; k++)
						pbp[c][e][k] = pb[k];
				}
			}
			// (the next channel's spectrum goes to U, last read in P0; its P0 writes V behind the barrier that follows the spectrum;
			// the next slot's floor tables were last read in the spectrum phase, barriers ago)
		}
	}
}

template <int BS>
static hipError_t launch_big(const LwBigArgs &F, int fmt, hipStream_t st)
{
	const dim3 g(F.n_wg), b(1u << (BS - 5));
	LwBigArgs A = F;
	if (fmt == LW_OUT_I16_PLANAR)
		return lw_launch_k(k_big<LW_OUT_I16_PLANAR, BS>, g, b, 0, st, A);
	if (fmt == LW_OUT_I16_INTERLEAVED)
		return lw_launch_k(k_big<LW_OUT_I16_INTERLEAVED, BS>, g, b, 0, st, A);
	return lw_launch_k(k_big<LW_OUT_F32_PLANAR, BS>, g, b, 0, st, A);
}

hipError_t lw_launch_big(const LwDevTables &T, const LwBatchDev &B, const LwShortLaunch &L, void *out, int fmt, hipStream_t st)
{
	if (L.n_tasks == 0)
		return hipSuccess;
	const LwDevBs &tb = T.bs[1];
	if ((L.lanes != 128 && L.lanes != 256) || tb.n != 32u * L.lanes)
		return hipErrorInvalidValue;
	LwBigArgs F{};
	F.residue = B.residue;
	F.floors = B.floors;
	F.slots = L.d_slots;
	F.state = B.state;
	F.td = B.td;
	F.out = out;
	F.A = tb.A;
	F.Bt = tb.B;
	F.C = tb.C;
	F.window = tb.window;
	F.inv_db = T.inv_db;
	F.bitrev = tb.bitrev;
	F.floor_x = T.floor_x;
	F.n_units = L.n_units;
	F.ch = T.ch;
	F.fstride = T.fstride;
	F.state_stride = T.state_stride;
	F.state_chan_stride = T.state_chan_stride;
	F.passes = L.passes ? L.passes : 1u;
	for (uint32_t i = 0; i < LW_FAST_MAX_FLOORS; i++)
		F.fl_of[i] = L.fl_of[i];
	for (uint32_t u = 0; u < L.n_units && u < LW_FAST_WAVES; u++)
		F.units[u] = L.units[u];
	F.n_wg = L.n_tasks * L.n_units;
	return L.lanes == 128 ? launch_big<12>(F, fmt, st) : launch_big<13>(F, fmt, st);
}
