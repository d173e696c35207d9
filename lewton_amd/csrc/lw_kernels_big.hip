// k_big<fmt, BS>: the long blocks of streams with blocksize_1 = 12 / 13 (4096 / 8192 points; header.rs:236-247 allows them, libvorbis
// never writes them) whose window slopes are both long -- floor curve, inverse coupling, floor x residue, IMDCT, window /
// overlap-add, sample conversion and state in ONE kernel, as k_long / k_short<L> do it for the smaller sizes (before: the generic
// kernels, ~16 barrier-separated radix-2 stages in LDS + a round trip of the time-domain block through HBM).
//
//   reference                                   here
//   src/audio.rs:526-555, :503-524 (floor 1)    floor_group(): interval search + render_line's closed form
//   src/audio.rs:762-777 (inverse coupling)     decouple()
//   src/imdct.rs:291-659 (inverse MDCT)         passes P0 .. P3 + E below
//   src/audio.rs:1082-1154 (overlap-add, state) phase E of the kernel
//   src/samples.rs:92-103 (conversion)          to_i16()
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
	const uint32_t *bitrev;
	const uint16_t *floor_x;
	uint32_t n_units, ch, fstride, state_stride, state_chan_stride, passes;
	uint32_t fl_of[LW_FAST_MAX_FLOORS]; // floor index (header order) of each staged floor slot of the units
	LwFastUnit units[LW_FAST_WAVES];
};

namespace {

__device__ __forceinline__ uint32_t pad8(uint32_t q) { return q + (q >> 3); }

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

// three stages of distances D, D/2, D/4 on the pairs q' = q0 + (D/4) i
template <uint32_t D, uint32_t N>
__device__ __forceinline__ void pass3(float2_t *V, uint32_t t, const float *A)
{
	constexpr uint32_t STEP = D / 4;
	const uint32_t off = t % STEP, q0 = (t / STEP) * (2u * D) + off;
	float2_t ta[4], tb[2], tc;
#pragma unroll
	for (uint32_t i = 0; i < 4; i++)
		ta[i] = ld2(A + (off + STEP * i) * (N / (2u * D)));
#pragma unroll
	for (uint32_t i = 0; i < 2; i++)
		tb[i] = ld2(A + (off + STEP * i) * (N / D));
	tc = ld2(A + off * (2u * N / D));
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

// audio.rs:762-777
__device__ __forceinline__ void decouple(float &m, float &a)
{
	float nm, na;
	if (m > 0.0f) {
		if (a > 0.0f) {
			nm = m;
			na = m - a;
		} else {
			nm = m + a;
			na = m;
		}
	} else {
		if (a > 0.0f) {
			nm = m;
			na = m + a;
		} else {
			nm = m - a;
			na = m;
		}
	}
	m = nm;
	a = na;
}

// samples.rs:92-103: x * 32768, clamp to [-32768, 32767], truncate toward zero; NaN -> 0
__device__ __forceinline__ int16_t to_i16(float x)
{
	const float t = x * 32768.0f;
	if (t > 32767.0f)
		return 32767;
	if (t < -32768.0f)
		return -32768;
	return (int16_t)(int)t;
}

// The floor values (indices into the inverse-dB table) of the bins k0 .. k0 + 3: the interval of k0 by binary search over the
// K active posts (ascending x; entry = x | y << 16), at most one step forward per bin (the x are distinct integers), and
// render_line's closed form y0 +- (|dy| (k - x0)) / adx (audio.rs:503-524) with the integer division by a
// reciprocal and a correction: |dy| (k - x0) < 2^21 is exact in f32 and v_rcp_f32 is good to 1 ulp, so the truncated product
// is off by at most one.
__device__ __forceinline__ void floor_group(const uint32_t *pxy, int K, uint32_t k0, int (&y)[4])
{
	int lo = 0, hi = K - 1;
	while (lo < hi) {
		const int mid = (lo + hi + 1) >> 1;
		if ((pxy[mid] & 0xffffu) <= k0)
			lo = mid;
		else
			hi = mid - 1;
	}
	uint32_t cur = pxy[lo], nxt = pxy[lo + 1 < K ? lo + 1 : lo];
#pragma unroll
	for (uint32_t j = 0; j < 4; j++) {
		const uint32_t k = k0 + j;
		if (lo + 1 < K && (nxt & 0xffffu) <= k) {
			lo++;
			cur = nxt;
			nxt = pxy[lo + 1 < K ? lo + 1 : lo];
		}
		const int x0 = (int)(cur & 0xffffu), y0 = (int)(cur >> 16);
		if (lo == K - 1) {
			y[j] = y0; // flat extension to n/2, audio.rs:546-548
		} else {
			const int x1 = (int)(nxt & 0xffffu), y1 = (int)(nxt >> 16);
			const int dy = y1 - y0, adx = x1 - x0;
			const int ady = dy < 0 ? -dy : dy;
			const int num = ady * ((int)k - x0);
			int q = (int)((float)num * __builtin_amdgcn_rcpf((float)adx));
			const int rem = num - q * adx;
			if (rem < 0)
				q--;
			else if (rem >= adx)
				q++;
			y[j] = dy < 0 ? y0 - q : y0 + q;
		}
	}
}

template <int FMT>
__device__ __forceinline__ void store_pair(void *out, uint32_t elem0, uint32_t pos, uint32_t stride, float a, float b)
{
	if (FMT == LW_OUT_F32_PLANAR) {
		*reinterpret_cast<float2_t *>(reinterpret_cast<float *>(out) + elem0 + pos) = float2_t{a, b};
	} else if (FMT == LW_OUT_I16_PLANAR) {
		*reinterpret_cast<uint32_t *>(reinterpret_cast<int16_t *>(out) + elem0 + pos) =
			(uint32_t)(uint16_t)to_i16(a) | ((uint32_t)(uint16_t)to_i16(b) << 16);
	} else {
		int16_t *o = reinterpret_cast<int16_t *>(out) + elem0;
		o[pos * stride] = to_i16(a);
		o[(pos + 1u) * stride] = to_i16(b);
	}
}

} // namespace

template <int FMT, int BS>
__global__ void __launch_bounds__(1 << (BS - 5)) k_big(LwBigArgs F)
{
	constexpr uint32_t n = 1u << BS, n2 = n / 2, n4 = n / 4, n8 = n / 8, T = n / 32;
	__shared__ __attribute__((aligned(16))) float U[n2];
	__shared__ __attribute__((aligned(16))) float2_t V[n4 + n4 / 8];
	__shared__ float inv_s[256];
	__shared__ uint32_t pxy[2][68];
	__shared__ uint8_t act[2][68];
	__shared__ int Kp[2];
	const uint32_t tid = threadIdx.x;
	const uint32_t task = blockIdx.x / F.n_units, uidx = blockIdx.x - task * F.n_units;
	const LwFastUnit un = F.units[uidx];
	const bool two = un.ch_b >= 0;
	const uint32_t chn[2] = {(uint32_t)un.ch_a, (uint32_t)(two ? un.ch_b : un.ch_a)};
	const uint32_t nch = two ? 2u : 1u;
	for (uint32_t i = tid; i < 256u; i += T)
		inv_s[i] = F.inv_db[i];
	// the previous block's right part as this thread needs it: pb(p) at p = 2m, 2m + 1, n/4 - 2 - 2m, n/4 - 1 - 2m, m = t + T e
	float pbp[2][2][4];
#pragma unroll
	for (int c = 0; c < 2; c++)
#pragma unroll
		for (int e = 0; e < 2; e++)
#pragma unroll
			for (int k = 0; k < 4; k++)
				pbp[c][e][k] = 0.0f;
	for (uint32_t pass = 0; pass < F.passes; pass++) {
		// launder the thread index once per slot: the table addresses derived from it are recomputed where they are used instead
		// of being hoisted out of the loop and kept in registers
		uint32_t t = tid;
		asm volatile("" : "+v"(t));
		const uint4 *sp = reinterpret_cast<const uint4 *>(F.slots + ((size_t)task * F.passes + pass));
		const uint4 d0 = sp[0], d1 = sp[1], d2 = sp[2];
		const uint32_t res_off = d0.x, floor_off = d0.y, out_off = d0.z, prev_arg = d0.w;
		const int32_t state_out = (int32_t)d1.x;
		const uint32_t prev_stride = d2.x & 0xffffu, kind = (d2.x >> 16) & 0xffu, prev_kind = d2.x >> 24, flags = d2.y;
		if (kind != LW_SS_BLOCK && kind != LW_SS_HALO)
			continue; // (the same for every thread of the workgroup)
		// ---- HBM loads, all at once: residues, floor records, the stored right part in front of the first block of a run
		float4_t r[2][4];
		uint32_t my_e[2] = {0u, 0u}, my_x[2] = {0u, 0u};
		bool unused[2] = {true, true};
#pragma unroll
		for (uint32_t c = 0; c < 2; c++) {
			if (c >= nch) {
#pragma unroll
				for (uint32_t i = 0; i < 4; i++)
					r[c][i] = float4_t{0.0f, 0.0f, 0.0f, 0.0f};
				continue;
			}
			const float4_t *s = reinterpret_cast<const float4_t *>(F.residue + res_off + chn[c] * n2);
#pragma unroll
			for (uint32_t i = 0; i < 4; i++)
				r[c][i] = __builtin_nontemporal_load(&s[t + T * i]);
			const uint16_t *frec = F.floors + floor_off + chn[c] * F.fstride;
			const uint32_t fl = F.fl_of[c == 0 ? un.floor_a : un.floor_b], Fp = c == 0 ? un.F_a : un.F_b;
			unused[c] = frec[0] == LW_FLOOR_UNUSED;
			if (t < Fp) {
				my_e[c] = frec[t];
				my_x[c] = F.floor_x[fl * LW_XSTRIDE + t];
			}
			if (kind == LW_SS_BLOCK && (prev_kind == LW_SP_STATE || prev_kind == LW_SP_TD)) {
				const float *src = prev_kind == LW_SP_STATE
					? F.state + ((size_t)prev_arg * 2u + ((flags & LW_RF_PARITY_IN) ? 1u : 0u)) * F.state_stride + chn[c] * F.state_chan_stride
					: F.td + prev_arg + chn[c] * prev_stride;
#pragma unroll
				for (uint32_t e = 0; e < 2; e++) { // pb(p) = right part at n/4 - 1 - p
					const uint32_t m = t + T * e;
					const float2_t a = ld2(src + (n4 - 2u - 2u * m)), b = ld2(src + 2u * m);
					pbp[c][e][0] = a.y;
					pbp[c][e][1] = a.x;
					pbp[c][e][2] = b.y;
					pbp[c][e][3] = b.x;
				}
			}
		}
		// ---- active floor posts in ascending x (audio.rs:536-545 walks exactly these): one thread per post
#pragma unroll
		for (uint32_t c = 0; c < 2; c++) {
			const uint32_t Fc = (c >= nch || unused[c]) ? 0u : (c == 0 ? un.F_a : un.F_b);
			if (t < Fc)
				act[c][t] = (my_e[c] & LW_POST_ACTIVE) ? 1 : 0;
		}
		__syncthreads();
#pragma unroll
		for (uint32_t c = 0; c < 2; c++) {
			const uint32_t Fc = (c >= nch || unused[c]) ? 0u : (c == 0 ? un.F_a : un.F_b);
			if (t < Fc) {
				int rank = 0;
				for (uint32_t u = 0; u < t; u++)
					rank += act[c][u];
				const bool on = (my_e[c] & LW_POST_ACTIVE) != 0;
				if (on)
					pxy[c][rank] = my_x[c] | ((my_e[c] & 0xffu) << 16);
				if (t == Fc - 1u)
					Kp[c] = rank + (on ? 1 : 0);
			}
			if (Fc == 0u && t == 0u)
				Kp[c] = 0;
		}
		__syncthreads();
		// ---- inverse coupling (audio.rs:762-777): ch_a = magnitude, ch_b = angle
		if (two && un.coupled) {
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
#pragma unroll
		for (uint32_t c = 0; c < 2; c++) { // (unrolled: everything indexed by the channel stays in registers)
			if (c >= nch)
				continue;
			// ---- spectrum = floor x residue (audio.rs:1035-1037; zero floor of an unused channel :1021-1024) -> U
			{
				const int K = Kp[c];
#pragma unroll
				for (uint32_t i = 0; i < 4; i++) {
					const uint32_t k0 = 4u * (t + T * i);
					float f[4];
					if (unused[c]) {
						f[0] = f[1] = f[2] = f[3] = 0.0f;
					} else {
						int y[4];
						floor_group(pxy[c], K, k0, y);
#pragma unroll
						for (int j = 0; j < 4; j++)
							f[j] = inv_s[y[j]];
					}
					const float4_t rr = r[c][i];
					*reinterpret_cast<float4_t *>(U + k0) = float4_t{f[0] * rr.x, f[1] * rr.y, f[2] * rr.z, f[3] * rr.w};
				}
			}
			__syncthreads();
			// ---- P0: step 1 (imdct.rs:337-371) of j = t + T i, the stage of distance n/8 (step 2, :385-430), n = 8192: of distance 512
			{
				float2_t H[4], Lo[4];
#pragma unroll
				for (uint32_t i = 0; i < 4; i++) {
					const uint32_t j = t + T * i;
					const float2_t a0 = ld2(F.A + 2u * j), a1 = ld2(F.A + n4 + 2u * j);
					const float4_t x = *reinterpret_cast<const float4_t *>(U + 4u * j);             // x0 = .x, x2 = .z
					const float4_t y = *reinterpret_cast<const float4_t *>(U + (n2 - 4u - 4u * j)); // u[e] = .y, u[e + 2] = .w
					H[i] = float2_t{x.x * a0.y + x.z * a0.x, x.x * a0.x - x.z * a0.y};
					const float me2 = -y.w, me0 = -y.y;
					Lo[i] = float2_t{me2 * a1.y + me0 * a1.x, me2 * a1.x - me0 * a1.y};
				}
#pragma unroll
				for (uint32_t i = 0; i < 4; i++)
					bfly2(H[i], Lo[i], ld2(F.A + 4u * (t + T * i)));
				if (BS == 13) {
#pragma unroll
					for (uint32_t i = 0; i < 2; i++) {
						const float2_t tw = ld2(F.A + 8u * (t + T * i));
						bfly2(H[i], H[i + 2], tw);
						bfly2(Lo[i], Lo[i + 2], tw);
					}
				}
#pragma unroll
				for (uint32_t i = 0; i < 4; i++) {
					V[pad8(t + T * i)] = H[i];
					V[pad8(n8 + t + T * i)] = Lo[i];
				}
			}
			__syncthreads();
			pass3<256, n>(V, t, F.A);
			__syncthreads();
			pass3<32, n>(V, t, F.A);
			__syncthreads();
			// ---- P3: imdct.rs:234-288 on the pairs q' = 8t .. 8t + 7 (z[2k], z[2k + 1] = pair 8t + 7 - k)
			{
				float z[16];
				const float a2 = F.A[n8];
#pragma unroll
				for (uint32_t k = 0; k < 8; k++) {
					const float2_t v = V[pad8(8u * t + 7u - k)];
					z[2 * k] = v.x;
					z[2 * k + 1] = v.y;
				}
				last3(z, a2);
#pragma unroll
				for (uint32_t k = 0; k < 8; k++)
					V[pad8(8u * t + 7u - k)] = float2_t{z[2 * k], z[2 * k + 1]};
			}
			__syncthreads();
			// ---- E: bit-reverse gather (imdct.rs:490-528), step 7 (:533-580), step 8 (:589-658), window / overlap-add, stores
			const uint32_t elem0 = FMT == LW_OUT_I16_INTERLEAVED ? out_off + chn[c] : out_off + chn[c] * n2;
			const uint32_t stride = FMT == LW_OUT_I16_INTERLEAVED ? F.ch : 1u;
			float *st_dst = (kind == LW_SS_BLOCK && state_out >= 0)
				? F.state + ((size_t)state_out * 2u + ((flags & LW_RF_PARITY_OUT) ? 1u : 0u)) * F.state_stride + chn[c] * F.state_chan_stride
				: nullptr;
			float *td_dst = (kind == LW_SS_BLOCK && (flags & LW_SF_WRITE_TD)) ? F.td + 2u * (size_t)res_off + chn[c] * n + n2 : nullptr;
			const bool samples = kind == LW_SS_BLOCK && prev_kind != LW_SP_NONE;
#pragma unroll
			for (uint32_t e = 0; e < 2; e++) {
				const uint32_t m = t + T * e, m0 = n / 16u - 1u - m;
				const uint2 b1 = *reinterpret_cast<const uint2 *>(F.bitrev + 2u * m), b0 = *reinterpret_cast<const uint2 *>(F.bitrev + 2u * m0);
				const float4_t Cq = ld4(F.C + 4u * m), Bl = ld4(F.Bt + 4u * m), Bh = ld4(F.Bt + (n2 - 4u - 4u * m));
				// the pair holding floats (k, k + 1) of the reference's array: q' = n/4 - 1 - k/2
				const float2_t Pa = V[pad8(n4 - 1u - b1.x / 2u)], Pb = V[pad8(n4 - 1u - b1.y / 2u)];
				const float2_t Pc = V[pad8(n4 - 2u - b0.x / 2u)], Pd = V[pad8(n4 - 2u - b0.y / 2u)];
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
					const float2_t wA = ld2(F.window + (n4 - 2u - 2u * m)), wB = ld2(F.window + (n4 + 2u * m));
					const float2_t wC = ld2(F.window + 2u * m), wD = ld2(F.window + (n2 - 2u - 2u * m));
					const float *pp = pbp[c][e];
					// p = 2m: q = n/4 - 1 - 2m (w = wA.y, mirror wB.x); p = 2m + 1: q = n/4 - 2 - 2m (wA.x, wB.y)
					const float s0 = (pa[0] * wA.y) + (pp[0] * wB.x), s0m = ((-pa[0]) * wB.x) + (pp[0] * wA.y);
					const float s1 = (pa[1] * wA.x) + (pp[1] * wB.y), s1m = ((-pa[1]) * wB.y) + (pp[1] * wA.x);
					// p = n/4 - 2 - 2m: q = 2m + 1 (wC.y, mirror wD.x); p = n/4 - 1 - 2m: q = 2m (wC.x, wD.y)
					const float s2 = (pa[2] * wC.y) + (pp[2] * wD.x), s2m = ((-pa[2]) * wD.x) + (pp[2] * wC.y);
					const float s3 = (pa[3] * wC.x) + (pp[3] * wD.y), s3m = ((-pa[3]) * wD.y) + (pp[3] * wC.x);
					store_pair<FMT>(F.out, elem0, n4 - 2u - 2u * m, stride, s1, s0);
					store_pair<FMT>(F.out, elem0, n4 + 2u * m, stride, s0m, s1m);
					store_pair<FMT>(F.out, elem0, 2u * m, stride, s3, s2);
					store_pair<FMT>(F.out, elem0, n2 - 2u - 2u * m, stride, s2m, s3m);
				}
				// the raw right part (audio.rs:1121, :1142-1147): pb(p) at n/4 - 1 - p and mirrored at n/4 + p
#pragma unroll
				for (int w = 0; w < 2; w++) {
					float *dst = w == 0 ? st_dst : td_dst;
					if (dst) {
						*reinterpret_cast<float2_t *>(dst + (n4 - 2u - 2u * m)) = float2_t{pb[1], pb[0]};
						*reinterpret_cast<float2_t *>(dst + (n4 + 2u * m)) = float2_t{pb[0], pb[1]};
						*reinterpret_cast<float2_t *>(dst + 2u * m) = float2_t{pb[3], pb[2]};
						*reinterpret_cast<float2_t *>(dst + (n2 - 2u - 2u * m)) = float2_t{pb[2], pb[3]};
					}
				}
#pragma unroll
				for (int k = 0; k < 4; k++)
					pbp[c][e][k] = pb[k];
			}
			// (the next channel's spectrum goes to U, last read in P0 two barriers ago; its P0 writes V behind one more barrier)
		}
	}
}

template <int BS>
static hipError_t launch_big(const LwBigArgs &F, uint32_t n_wg, int fmt, hipStream_t st)
{
	const dim3 g(n_wg), b(1u << (BS - 5));
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
	return L.lanes == 128 ? launch_big<12>(F, L.n_tasks * L.n_units, fmt, st) : launch_big<13>(F, L.n_tasks * L.n_units, fmt, st);
}
