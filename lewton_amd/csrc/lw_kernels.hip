// HIP kernels of the Vorbis audio-packet synthesis path for gfx950 (MI355X) -- GENERIC kernels.
//
// These kernels handle every block size (64..8192), every window shape and any channel count /
// coupling list; they are the correctness backbone and the fallback.  The specialised long-block
// kernel lives in lw_kernels_long.hip.
//
// Arithmetic contract: every f32 operation below is the same individually
// rounded operation, on the same operands, as in the reference; this file is compiled with
// -ffp-contract=off so hipcc never forms v_fma_f32.  Parallelism only ever comes from running
// independent index tuples of one step at the same time.
//
// Reference restated (paths relative to RustAudio/lewton 0.10.2):
//   k_decouple       src/audio.rs:762-777, :990-1002
//   k_imdct_generic  src/audio.rs:526-555 (floor curve, closed form of render_line :503-524),
//                    :1006-1039 (floor x residue), src/imdct.rs:291-659 (all steps)
//   k_ola_generic    src/audio.rs:1082-1154 (overlap add, state), src/samples.rs:32-103 (conversion)
#include "lw_kernels.hpp"

#define LW_BLOCK 256
// workgroup size of k_decouple / k_ola_generic (64-thread workgroups measured slower: 20.7 / 14.2 us vs 11.9 / 11.6 us on the
// mixed short/long configuration)
#define LW_ELEMENTWISE_BLOCK 256

// ---------------------------------------------------------------------------------------------
// inverse coupling
// ---------------------------------------------------------------------------------------------
// packet of generic task `t` (dense lists of the batch) or t itself when every packet is generic
__device__ __forceinline__ bool generic_packet(const LwBatchDev &B, uint32_t t, uint32_t &pkt)
{
	if (!B.gen_small) {
		pkt = t;
		return t < B.n_packets;
	}
	if (t < B.n_gen_small)
		pkt = B.gen_small[t];
	else if (t - B.n_gen_small < B.n_gen_large)
		pkt = B.gen_large[t - B.n_gen_small];
	else
		return false;
	return true;
}

__global__ void __launch_bounds__(LW_BLOCK) k_decouple(LwDevTables T, LwBatchDev B, uint32_t skip_mask)
{
	uint32_t pkt;
	if (!generic_packet(B, blockIdx.x, pkt))
		return;
	const LwPacketRec rec = B.recs[pkt];
	if (rec.flags & skip_mask)
		return;
	const uint32_t n2 = (1u << rec.bs) >> 1;
	const uint32_t s0 = T.couple_off[rec.mode], s1 = T.couple_off[rec.mode + 1];
	const float *src = B.residue + rec.res_off;
	float *dst = B.decoupled + rec.res_off;
	for (uint32_t k = threadIdx.x; k < n2; k += blockDim.x) {
		for (uint32_t c = 0; c < T.ch; c++)
			dst[c * n2 + k] = src[c * n2 + k];
		for (uint32_t s = s1; s-- > s0;) { // reverse step order, audio.rs:991-992
			const uint32_t mi = T.couple[2 * s] * n2 + k, ai = T.couple[2 * s + 1] * n2 + k;
			const float m = dst[mi], a = dst[ai];
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
			dst[mi] = nm;
			dst[ai] = na;
		}
	}
}

// ---------------------------------------------------------------------------------------------
// k_prep: the canonicalising pre-pass of the specialised kernels (LwPrepPlan, lw_fast.hpp; round 6).  One workgroup per listed
// packet.  Out: B.decoupled = the packet's residue vectors after ALL coupling steps of its mode (audio.rs:990-1002, reverse
// order), and for the channels whose action is LW_PREP_PREMUL already multiplied with their floor curve (audio.rs:1035-1037:
// floor-1 curve by the closed form of render_line, SURVEY 9.3, or the host-evaluated floor-0 curve of B.fcurve);
// floors_out (if given) = the packet's floor records with the UNIT floor (posts 0 and 1 active at inverse-dB index 255 = 1.0)
// for those channels, a copy for the others.  An unused floor (audio.rs:1021-1024) stays marked: the kernels zero the channel.
// Every thread owns the bins tid, tid + 256, ...: no synchronisation between the coupling and the multiply.
// ---------------------------------------------------------------------------------------------
// floor-1 value of bin k from the active posts (px ascending, py = final_y * multiplier): closed form of render_line
__device__ __forceinline__ int prep_floor_y(const uint16_t *px, const uint8_t *py, int K, uint32_t k)
{
	int lo = 0, hi = K - 1; // largest i with px[i] <= k
	while (lo < hi) {
		const int mid = (lo + hi + 1) >> 1;
		if (px[mid] <= k)
			lo = mid;
		else
			hi = mid - 1;
	}
	if (lo == K - 1)
		return py[lo]; // flat extension to n/2, audio.rs:546-548
	const int x0 = px[lo], x1 = px[lo + 1], y0 = py[lo], y1 = py[lo + 1];
	const int dy = y1 - y0, adx = x1 - x0;
	const int ady = dy < 0 ? -dy : dy;
	const int off = (ady * ((int)k - x0)) / adx;
	return dy < 0 ? y0 - off : y0 + off;
}

__device__ __forceinline__ void prep_couple(float &m, float &a) // audio.rs:762-777
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

// LDS: every thread keeps four consecutive bins of every channel in its own column col[c][tid] (16-byte accesses of consecutive
// lanes: conflict-free, and private -- no synchronisation between the coupling steps); behind the columns the inverse-dB table and,
// per channel that is multiplied here, the active posts of its floor in ascending x -- all of them built up front behind ONE barrier
// (a barrier and a dependent round trip per channel made this kernel 45 us for 4096 stereo packets: three times its traffic's worth)
#define LW_PREP_MAX_CH 16u
// Per multiplied channel: the active posts in ascending x and one entry {dy, c0, 1 / adx, w} per interval between them -- the
// two-FMA form of render_line the wave-pipeline kernels use (floor_table / floor_bin in lw_kernels_long.hip: y(k) from the
// mantissa of fma(fma(k, dy, c0), 1 / adx, w); proven equal to the integer form for every dy and every adx <= 4096 with the
// reciprocal one ulp off either way, tests/test_fast_model.py, tests/test_big_model.py).  An interval longer than that (a post
// far beyond the block: range bits up to 15) keeps the integer division (ent.z = 0 marks it).
struct LwPrepPosts {
	float4 ent[LW_XSTRIDE];
	uint16_t px[LW_XSTRIDE];
	uint8_t py[LW_XSTRIDE + 2];
	int32_t K;
	int32_t pad;
};
#define LW_PREP_W0 2097153.0f // 2^21 + 1
#define LW_PREP_MASK 0x7fcu

// floor value of bin k inside active interval r (px[r] <= k < px[r + 1], or r = K - 1: flat to the end, audio.rs:546-548)
__device__ __forceinline__ float prep_floor_bin(const LwPrepPosts &P, const float *inv_db, int r, uint32_t k)
{
	const float4 e = P.ent[r];
	if (e.z != 0.0f) {
		const float t = __builtin_fmaf(__builtin_fmaf((float)k, e.x, e.y), e.z, e.w);
		return inv_db[((__float_as_uint(t) & LW_PREP_MASK) >> 2) - 1u];
	}
	const int x0 = P.px[r], x1 = P.px[r + 1], y0 = P.py[r], y1 = P.py[r + 1];
	const int dy = y1 - y0, adx = x1 - x0;
	const int ady = dy < 0 ? -dy : dy;
	const int off = (ady * ((int)k - x0)) / adx; // closed form of render_line
	return inv_db[dy < 0 ? y0 - off : y0 + off];
}
__global__ void __launch_bounds__(LW_ELEMENTWISE_BLOCK) k_prep(LwDevTables T, LwBatchDev B, const uint32_t *list, const uint8_t *action,
		uint16_t *floors_out)
{
	extern __shared__ __attribute__((aligned(16))) char prep_smem[];
	const uint32_t tid = threadIdx.x, ch = T.ch, nthr = blockDim.x;
	float4 *col = reinterpret_cast<float4 *>(prep_smem); // [ch][256]
	float *inv_db = reinterpret_cast<float *>(prep_smem + (size_t)ch * nthr * 16u); // [256]
	LwPrepPosts *posts = reinterpret_cast<LwPrepPosts *>(inv_db + 256); // [ch]
	const uint32_t pkt = list[blockIdx.x];
	const LwPacketRec rec = B.recs[pkt];
	if (rec.flags & LW_RF_SKIP)
		return;
	const uint32_t n2 = (1u << rec.bs) >> 1;
	const uint32_t s0 = T.couple_off[rec.mode], s1 = T.couple_off[rec.mode + 1];
	const float *src = B.residue + rec.res_off;
	float *dst = B.decoupled + rec.res_off;
	const uint8_t *arow = action + (size_t)rec.mode * ch;
	// ---- my four bins of every channel are requested first (the first turn of the loop below) ...
	const uint32_t k_first = 4u * tid;
	if (k_first < n2)
		for (uint32_t c = 0; c < ch; c++)
			col[c * nthr + tid] = *reinterpret_cast<const float4 *>(src + c * n2 + k_first);
	// ---- ... then the tables: inverse dB, floor records out, and the active posts of the channels multiplied here, one WAVE per
	//      channel in turn (lanes = posts: the rank of an active post is the number of active posts below it, a ballot)
	inv_db[tid] = T.inv_db[tid];
	const uint32_t wave = tid >> 6, lane = tid & 63u;
	for (uint32_t c = wave; c < ch; c += nthr >> 6) {
		// (everything that does not depend on another load is requested before anything is looked at: the kernel's time is the
		// length of its chain of dependent round trips -- list -> record -> {residues, floor record, floor number} -> {posts' x})
		const uint16_t *frec = B.floors + rec.floor_off + c * T.fstride;
		const uint16_t e = lane < T.fstride ? frec[lane] : (uint16_t)0;
		const uint16_t e64 = T.fstride > 64 ? frec[64] : (uint16_t)0;
		const uint32_t act = arow[c];
		const uint32_t fl = T.mode_floor[rec.mode * ch + c];
		const uint32_t F = T.floor_F[fl];
		const uint16_t x = T.floor_x[fl * LW_XSTRIDE + lane], x64 = T.floor_x[fl * LW_XSTRIDE + 64];
		const uint16_t e0 = (uint16_t)__builtin_amdgcn_readfirstlane((uint32_t)e);
		const bool premul = act == LW_PREP_PREMUL && e0 != LW_FLOOR_UNUSED;
		if (floors_out) {
			uint16_t *fout = floors_out + rec.floor_off + c * T.fstride;
			if (!premul) {
				if (lane < T.fstride)
					fout[lane] = e;
				for (uint32_t i = lane + 64u; i < T.fstride; i += 64u)
					fout[i] = frec[i];
			} else if (lane < 2) {
				fout[lane] = (uint16_t)(LW_POST_ACTIVE | 255u);
			}
		}
		if (premul && e0 != LW_FLOOR_EXPLICIT) { // (audio.rs:536-545 walks exactly these)
			const bool on = lane < F && (e & LW_POST_ACTIVE);
			const unsigned long long M = __ballot(on);
			const int rank = __builtin_popcountll(M & ((1ull << lane) - 1ull));
			if (on) {
				posts[c].px[rank] = x;
				posts[c].py[rank] = (uint8_t)(e & 0xff);
			}
			int K = __builtin_popcountll(M);
			if (F > 64 && lane == 0 && (e64 & LW_POST_ACTIVE)) { // the 65th post (header.rs:873)
				posts[c].px[K] = x64;
				posts[c].py[K] = (uint8_t)(e64 & 0xff);
				K++;
			}
			if (lane == 0)
				posts[c].K = K;
			// (the wave's own LDS stores above are visible to its loads below: in-order LDS, a compiler fence is enough)
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			K = __builtin_amdgcn_readfirstlane(K);
			for (int r = (int)lane; r < K; r += 64) {
				const LwPrepPosts &P = posts[c];
				const bool last = r == K - 1;
				const float xlo = (float)P.px[r], ylo = (float)P.py[r];
				const float xhi = last ? xlo + 1.0f : (float)P.px[r + 1], yhi = last ? ylo : (float)P.py[r + 1];
				const float dy = yhi - ylo, adx = xhi - xlo;
				const bool down = yhi < ylo;
				float4 en;
				en.x = dy;
				en.y = down ? (0.875f * adx - 0.5f) - xhi * dy : (0.5f - 0.125f * adx) - xlo * dy; // exact (22 bits at most)
				en.z = last ? 1.0f : adx <= 4096.0f ? __builtin_amdgcn_rcpf(adx) : 0.0f;
				en.w = (down ? yhi : ylo) + LW_PREP_W0;
				posts[c].ent[r] = en;
			}
		}
	}
	__syncthreads();
	for (uint32_t k0 = 0; k0 < n2; k0 += 4u * nthr) { // (n2 >= 32: whole float4s; threads beyond the block idle)
		const uint32_t k = k0 + 4u * tid;
		if (k >= n2)
			break;
		// ---- all channels of my four bins in my column, every coupling step there (reverse order, audio.rs:991-992)
		if (k0)
			for (uint32_t c = 0; c < ch; c++)
				col[c * nthr + tid] = *reinterpret_cast<const float4 *>(src + c * n2 + k);
		for (uint32_t s = s1; s-- > s0;) {
			float4 &m = col[T.couple[2 * s] * nthr + tid], &a = col[T.couple[2 * s + 1] * nthr + tid];
			float4 mv = m, av = a;
			prep_couple(mv.x, av.x);
			prep_couple(mv.y, av.y);
			prep_couple(mv.z, av.z);
			prep_couple(mv.w, av.w);
			m = mv;
			a = av;
		}
		// ---- x floor curve for the channels the kernels cannot stage (audio.rs:1035-1037), and out
		for (uint32_t c = 0; c < ch; c++) {
			float4 v = col[c * nthr + tid];
			if (arow[c] == LW_PREP_PREMUL) {
				const uint16_t e0 = B.floors[rec.floor_off + c * T.fstride];
				if (e0 == LW_FLOOR_EXPLICIT) { // floor 0: the curve evaluated by the host stage (audio.rs:160-212)
					const float4 f = *reinterpret_cast<const float4 *>(B.fcurve + rec.res_off + c * n2 + k);
					v.x = f.x * v.x;
					v.y = f.y * v.y;
					v.z = f.z * v.z;
					v.w = f.w * v.w;
				} else if (e0 != LW_FLOOR_UNUSED) {
					const LwPrepPosts &P = posts[c];
					const int K = P.K;
					int lo = 0, hi = K - 1; // largest r with px[r] <= k: searched once, my other three bins step along
					while (lo < hi) {
						const int mid = (lo + hi + 1) >> 1;
						if (P.px[mid] <= k)
							lo = mid;
						else
							hi = mid - 1;
					}
					float f[4];
#pragma unroll
					for (uint32_t j = 0; j < 4; j++) {
						while (lo + 1 < K && P.px[lo + 1] <= k + j)
							lo++;
						f[j] = prep_floor_bin(P, inv_db, lo, k + j);
					}
					v.x = f[0] * v.x;
					v.y = f[1] * v.y;
					v.z = f[2] * v.z;
					v.w = f[3] * v.w;
				}
			}
			*reinterpret_cast<float4 *>(dst + c * n2 + k) = v;
		}
	}
}

// the same for more channels than the columns hold (LW_PREP_MAX_CH): through B.decoupled itself, one bin per thread and turn
__global__ void __launch_bounds__(LW_ELEMENTWISE_BLOCK) k_prep_wide(LwDevTables T, LwBatchDev B, const uint32_t *list, const uint8_t *action,
		uint16_t *floors_out)
{
	__shared__ uint16_t px[LW_XSTRIDE];
	__shared__ uint8_t py[LW_XSTRIDE + 2];
	__shared__ uint8_t act[LW_XSTRIDE + 2];
	__shared__ int s_K;
	const uint32_t pkt = list[blockIdx.x];
	const LwPacketRec rec = B.recs[pkt];
	if (rec.flags & LW_RF_SKIP)
		return;
	const uint32_t tid = threadIdx.x, n2 = (1u << rec.bs) >> 1;
	const uint32_t s0 = T.couple_off[rec.mode], s1 = T.couple_off[rec.mode + 1];
	const float *src = B.residue + rec.res_off;
	float *dst = B.decoupled + rec.res_off;
	for (uint32_t k = tid; k < n2; k += blockDim.x) {
		for (uint32_t c = 0; c < T.ch; c++)
			dst[c * n2 + k] = src[c * n2 + k];
		for (uint32_t s = s1; s-- > s0;) // reverse step order, audio.rs:991-992
			prep_couple(dst[T.couple[2 * s] * n2 + k], dst[T.couple[2 * s + 1] * n2 + k]);
	}
	const uint8_t *arow = action + (size_t)rec.mode * T.ch;
	for (uint32_t c = 0; c < T.ch; c++) { // (every condition below is the same for all threads of the workgroup)
		const uint16_t *frec = B.floors + rec.floor_off + c * T.fstride;
		uint16_t *fout = floors_out ? floors_out + rec.floor_off + c * T.fstride : nullptr;
		const uint16_t e0 = frec[0];
		if (arow[c] != LW_PREP_PREMUL || e0 == LW_FLOOR_UNUSED) {
			if (fout)
				for (uint32_t i = tid; i < T.fstride; i += blockDim.x)
					fout[i] = frec[i];
			continue;
		}
		float *v = dst + c * n2;
		if (e0 == LW_FLOOR_EXPLICIT) {
			const float *fc = B.fcurve + rec.res_off + c * n2;
			for (uint32_t k = tid; k < n2; k += blockDim.x)
				v[k] = fc[k] * v[k];
		} else {
			const uint32_t fl = T.mode_floor[rec.mode * T.ch + c], F = T.floor_F[fl];
			uint16_t my_e = 0;
			if (tid < F) {
				my_e = frec[tid];
				act[tid] = (my_e & LW_POST_ACTIVE) ? 1 : 0;
			}
			__syncthreads();
			if (tid < F) {
				int rank = 0;
				for (uint32_t t = 0; t < tid; t++)
					rank += act[t];
				if (my_e & LW_POST_ACTIVE) {
					px[rank] = T.floor_x[fl * LW_XSTRIDE + tid];
					py[rank] = (uint8_t)(my_e & 0xff);
				}
				if (tid == F - 1)
					s_K = rank + ((my_e & LW_POST_ACTIVE) ? 1 : 0);
			}
			__syncthreads();
			const int K = s_K;
			for (uint32_t k = tid; k < n2; k += blockDim.x)
				v[k] = T.inv_db[prep_floor_y(px, py, K, k)] * v[k];
			__syncthreads(); // (px / py are rebuilt for the next channel)
		}
		if (fout && tid < 2)
			fout[tid] = (uint16_t)(LW_POST_ACTIVE | 255u);
	}
}

hipError_t lw_launch_prep(const LwDevTables &T, const LwBatchDev &B, const uint32_t *d_list, uint32_t n_list, const uint8_t *d_action,
		uint16_t *d_floors_out, hipStream_t st)
{
	if (n_list == 0)
		return hipSuccess;
	if (T.ch > LW_PREP_MAX_CH)
		return lw_launch_k(k_prep_wide, dim3(n_list), dim3(LW_ELEMENTWISE_BLOCK), 0, st, T, B, d_list, d_action, d_floors_out);
	static LwPerDeviceOnce once;
	{
		const hipError_t e = once.run([] {
			return hipFuncSetAttribute((const void *)k_prep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LW_PREP_MAX_CH * (LW_ELEMENTWISE_BLOCK * 16 + sizeof(LwPrepPosts)) + 1024));
		});
		if (e != hipSuccess)
			return e;
	}
	return lw_launch_k(k_prep, dim3(n_list), dim3(LW_ELEMENTWISE_BLOCK), (size_t)T.ch * (LW_ELEMENTWISE_BLOCK * 16 + sizeof(LwPrepPosts)) + 1024, st, T,
			B, d_list, d_action, d_floors_out);
}

// ---------------------------------------------------------------------------------------------
// floor curve + multiply + IMDCT, one workgroup per (packet, channel)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void bfly(float *u, uint32_t hi, uint32_t lo, float t0, float t1)
{
	const float k00 = u[hi] - u[lo];
	const float k01 = u[hi - 1] - u[lo - 1];
	u[hi] = u[hi] + u[lo];
	u[hi - 1] = u[hi - 1] + u[lo - 1];
	u[lo] = k00 * t0 - k01 * t1;
	u[lo - 1] = k01 * t0 + k00 * t1;
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

template <int TPB>
__device__ __forceinline__ void stage_sync()
{
	if (TPB == LW_BLOCK)
		__syncthreads();
	else
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // one wave: its LDS operations complete in order
}

// One (packet, channel) block: floor curve, floor x residue, IMDCT (imdct.rs:291-659) by TPB threads in `smem`; the n
// time-domain values go to `out` (HBM, or LDS for the fused small-block kernel).
// TPB threads work on one (packet, channel) block: TPB = LW_BLOCK (one block per workgroup, barriers between the stages) for
// the large block sizes, TPB = 64 (one wave per block; the stages are ordered by the wave's own LDS ordering, no barrier)
// for block sizes up to 2^LW_SMALL_BS -- a 256-point short block has 32 butterflies per stage, a 256-thread workgroup and
// its barriers were 4x the work of the transform itself.
struct LwBlockTables { // tables of one block size + the inverse-dB table: in HBM / L2, or copies in LDS
	const float *A, *Bt, *C, *inv_db;
	const uint32_t *bitrev;
};

__device__ __forceinline__ LwBlockTables global_tables(const LwDevTables &T, const LwPacketRec &rec)
{
	const LwDevBs &tb = T.bs[(rec.flags & LW_RF_LONG) ? 1 : 0];
	return LwBlockTables{tb.A, tb.B, tb.C, T.inv_db, tb.bitrev};
}

template <int TPB>
__device__ __forceinline__ void imdct_block(const LwDevTables &T, const LwBatchDev &B, const LwPacketRec &rec, uint32_t c, uint32_t tid,
		float *smem, float *out, float *tap_spec, int use_decoupled, int partner, int role, const LwBlockTables tabs,
		int fl_known = -1, int F_known = -1)
{
	const uint32_t bs = rec.bs, n = 1u << bs, n2 = n >> 1, n4 = n >> 2, n8 = n >> 3;
	float *u = smem, *v = smem + n2;
	uint16_t *px = (uint16_t *)(smem + 2 * n2);
	uint8_t *py = (uint8_t *)(px + LW_XSTRIDE);
	int *s_Kp = (int *)(py + LW_XSTRIDE + 2); // 4-byte aligned: 2 * n2 floats + 66 * 2 + 68 bytes

	// ---- the residue values this thread will multiply (short blocks, task descriptors: the addresses are known from the one
	//      load that brought the task) are requested first, together with the floor record: one round trip for everything
	const float *src = (use_decoupled ? B.decoupled : B.residue) + rec.res_off + c * n2;
	const float *psrc = partner >= 0 ? B.residue + rec.res_off + (uint32_t)partner * n2 : nullptr;
	const bool prefetched = TPB == 64 && F_known >= 0 && n2 <= 4u * TPB;
	float pre_r[4] = {0.0f, 0.0f, 0.0f, 0.0f}, pre_p[4] = {0.0f, 0.0f, 0.0f, 0.0f};
	if (TPB == 64 && prefetched) {
#pragma unroll
		for (uint32_t q = 0; q < 4; q++) {
			const uint32_t k = tid + q * TPB;
			if (k < n2) {
				pre_r[q] = src[k];
				if (psrc)
					pre_p[q] = psrc[k];
			}
		}
	}
	// ---- active floor posts, ascending x (audio.rs:536-545 walks exactly these)
	const uint16_t *frec = B.floors + rec.floor_off + c * T.fstride;
	const uint32_t fl = fl_known >= 0 ? (uint32_t)fl_known : T.mode_floor[rec.mode * T.ch + c];
	uint8_t *act = (uint8_t *)(s_Kp + 1);
	// one thread per post (F <= 65): a serial walk by thread 0 is a chain of ~2 F dependent global loads, tens of
	// microseconds -- it used to be most of this kernel's run time.  With the post count known from the task descriptor the
	// posts are requested before the record's first entry has said whether the floor is used at all.
	uint16_t my_e = 0, my_x = 0;
	if (F_known >= 0) {
		const uint32_t sidx = tid; // (TPB = 64; a 65th post is reloaded below like in the other path)
		if (sidx < (uint32_t)F_known) {
			my_e = frec[sidx];
			my_x = T.floor_x[fl * LW_XSTRIDE + sidx];
		}
	}
	const uint16_t e0 = frec[0];
	const bool unused = e0 == LW_FLOOR_UNUSED;
	const bool explicit_curve = e0 == LW_FLOOR_EXPLICIT; // floor 0: curve evaluated by the host stage
	const uint32_t F = (unused || explicit_curve) ? 0u : F_known >= 0 ? (uint32_t)F_known : T.floor_F[fl];
	for (uint32_t s0 = 0; s0 < F; s0 += TPB) { // one iteration unless TPB = 64 and F = 65
		const uint32_t sidx = s0 + tid;
		if (sidx < F) {
			if (!(F_known >= 0 && s0 == 0)) {
				my_e = frec[sidx];
				my_x = T.floor_x[fl * LW_XSTRIDE + sidx];
			}
			act[sidx] = (my_e & LW_POST_ACTIVE) ? 1 : 0;
		}
	}
	stage_sync<TPB>();
	for (uint32_t s0 = 0; s0 < F; s0 += TPB) {
		const uint32_t sidx = s0 + tid;
		if (sidx < F) {
			int rank = 0;
			for (uint32_t t = 0; t < sidx; t++)
				rank += act[t];
			// my_e / my_x hold the LAST round's post: reload in the (TPB = 64, F = 65) case for the earlier round
			const bool last_round = s0 + TPB >= F;
			const uint16_t e = last_round ? my_e : frec[sidx];
			const uint16_t x = last_round ? my_x : T.floor_x[fl * LW_XSTRIDE + sidx];
			if (e & LW_POST_ACTIVE) {
				px[rank] = x;
				py[rank] = (uint8_t)(e & 0xff);
			}
			if (sidx == F - 1)
				*s_Kp = rank + ((e & LW_POST_ACTIVE) ? 1 : 0);
		}
	}
	if (F == 0 && tid == 0)
		*s_Kp = 0;
	stage_sync<TPB>();
	const int K = *s_Kp;

	// ---- spectrum = floor * residue (audio.rs:1035-1037); zero floor for an unused channel (:1021-1024)
	// residue of this channel after inverse coupling: from k_decouple's buffer, or -- `partner` >= 0: this channel takes part
	// in exactly one coupling step of the mode with that channel (role 1 = magnitude, 2 = angle) -- computed here from the two
	// raw vectors (audio.rs:762-777), or the raw vector itself
#pragma unroll 4
	for (uint32_t k = tid, q = 0; k < n2; k += TPB, q++) {
		float f;
		if (unused) {
			f = 0.0f;
		} else if (explicit_curve) {
			f = B.fcurve[rec.res_off + c * n2 + k];
		} else {
			int lo = 0, hi = K - 1; // largest i with px[i] <= k
			while (lo < hi) {
				const int mid = (lo + hi + 1) >> 1;
				if (px[mid] <= k)
					lo = mid;
				else
					hi = mid - 1;
			}
			int y;
			if (lo == K - 1) {
				y = py[lo]; // flat extension to n/2, audio.rs:546-548
			} else {
				const int x0 = px[lo], x1 = px[lo + 1], y0 = py[lo], y1 = py[lo + 1];
				const int dy = y1 - y0, adx = x1 - x0;
				const int ady = dy < 0 ? -dy : dy;
				const int off = (ady * ((int)k - x0)) / adx; // closed form of render_line
				y = dy < 0 ? y0 - off : y0 + off;
			}
			f = tabs.inv_db[y];
		}
		float r = TPB == 64 && prefetched ? pre_r[q & 3u] : src[k];
		if (psrc) {
			const float pr = TPB == 64 && prefetched ? pre_p[q & 3u] : psrc[k];
			const float m = role == 1 ? r : pr, a = role == 1 ? pr : r;
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
			r = role == 1 ? nm : na;
		}
		const float x = f * r;
		u[k] = x;
		if (tap_spec)
			tap_spec[rec.res_off + c * n2 + k] = x;
	}
	stage_sync<TPB>();

	// twiddles stay in L2 (copies in LDS were measured slower: fewer resident workgroups, no gain per stage)
	const float *A = tabs.A, *Bt = tabs.Bt, *C = tabs.C;
	// ---- imdct.rs:337-371: X = u -> v
	for (uint32_t j = tid; j < n8; j += TPB) {
		const float x0 = u[4 * j], x2 = u[4 * j + 2];
		v[n2 - 1 - 2 * j] = x0 * A[2 * j] - x2 * A[2 * j + 1];
		v[n2 - 2 - 2 * j] = x0 * A[2 * j + 1] + x2 * A[2 * j];
		const uint32_t e = n2 - 3 - 4 * j, a = n4 + 2 * j, d = n4 - 2 - 2 * j;
		const float me2 = -u[e + 2], me0 = -u[e];
		v[d + 1] = me2 * A[a] - me0 * A[a + 1];
		v[d] = me2 * A[a + 1] + me0 * A[a];
	}
	stage_sync<TPB>();
	// ---- imdct.rs:385-430 (step 2): v -> u
	for (uint32_t i = tid; i < (n >> 4); i += TPB) {
		const uint32_t a = n2 - 8 - 8 * i, lo = 4 * i, hi = n4 + 4 * i;
		{
			const float p = v[hi + 1] - v[lo + 1], q = v[hi] - v[lo];
			u[hi + 1] = v[hi + 1] + v[lo + 1];
			u[hi] = v[hi] + v[lo];
			u[lo + 1] = p * A[a + 4] - q * A[a + 5];
			u[lo] = q * A[a + 4] + p * A[a + 5];
		}
		{
			const float p = v[hi + 3] - v[lo + 3], q = v[hi + 2] - v[lo + 2];
			u[hi + 3] = v[hi + 3] + v[lo + 3];
			u[hi + 2] = v[hi + 2] + v[lo + 2];
			u[lo + 3] = p * A[a] - q * A[a + 1];
			u[lo + 2] = q * A[a] + p * A[a + 1];
		}
	}
	stage_sync<TPB>();
	// ---- imdct.rs:445-477 (step 3): stages 0 and 1 always run, then up to ld-7 (for bs 6/7 this is the
	//      literal behaviour of the reference: see DESIGN.md "small block sizes")
	const int last_stage = (int)bs - 7 > 1 ? (int)bs - 7 : 1;
	for (int l = 0; l <= last_stage; l++) {
		const uint32_t per_s = n >> (l + 4);
		if (l == 1 && per_s < 4) // imdct_step3_inner_r_loop runs lim>>2 groups of four (imdct.rs:93)
			break;
		const uint32_t k0 = n >> (l + 2), k1 = 1u << (l + 3);
		for (uint32_t b = tid; b < n8; b += TPB) {
			const uint32_t s = b / per_s, r = b - s * per_s;
			const uint32_t hi = n2 - 1 - k0 * s - 2 * r;
			bfly(u, hi, hi - (k0 >> 1), A[r * k1], A[r * k1 + 1]);
		}
		stage_sync<TPB>();
	}
	// ---- imdct.rs:234-288 fused last three stages, one 16-float group per thread
	{
		const float a2 = A[n8];
		for (uint32_t g = tid; g < (n >> 5); g += TPB) {
			float *z = u + (n2 - 16 - 16 * g); // z[15 - k] is the reference's z![-k]
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
	}
	stage_sync<TPB>();
	// ---- imdct.rs:490-528 bit-reverse: u -> v
	for (uint32_t t = tid; t < (n >> 4); t += TPB) {
		uint32_t k = tabs.bitrev[2 * t];
		const uint32_t d1 = n2 - 4 - 4 * t, d0 = n4 - 4 - 4 * t;
		v[d1 + 3] = u[k];
		v[d1 + 2] = u[k + 1];
		v[d0 + 3] = u[k + 2];
		v[d0 + 2] = u[k + 3];
		k = tabs.bitrev[2 * t + 1];
		v[d1 + 1] = u[k];
		v[d1] = u[k + 1];
		v[d0 + 1] = u[k + 2];
		v[d0] = u[k + 3];
	}
	stage_sync<TPB>();
	// ---- imdct.rs:533-580 step 7, in place on v
	for (uint32_t m = tid; m < (n >> 4); m += TPB) {
		const uint32_t d = 4 * m, e = n2 - 4 - 4 * m;
		{
			const float a02 = v[d] - v[e + 2], a11 = v[d + 1] + v[e + 3];
			const float b0 = C[d + 1] * a02 + C[d] * a11, b1 = C[d + 1] * a11 - C[d] * a02;
			const float b2 = v[d] + v[e + 2], b3 = v[d + 1] - v[e + 3];
			v[d] = b2 + b0;
			v[d + 1] = b3 + b1;
			v[e + 2] = b2 - b0;
			v[e + 3] = b1 - b3;
		}
		{
			const float a02 = v[d + 2] - v[e], a11 = v[d + 3] + v[e + 1];
			const float b0 = C[d + 3] * a02 + C[d + 2] * a11, b1 = C[d + 3] * a11 - C[d + 2] * a02;
			const float b2 = v[d + 2] + v[e], b3 = v[d + 3] - v[e + 1];
			v[d + 2] = b2 + b0;
			v[d + 3] = b3 + b1;
			v[e] = b2 - b0;
			v[e + 1] = b1 - b3;
		}
	}
	stage_sync<TPB>();
	// ---- imdct.rs:589-658 step 8 + output mapping -> time-domain block in HBM
	for (uint32_t p = tid; p < n4; p += TPB) {
		const float w0 = v[2 * p], w1 = v[2 * p + 1];
		const float pa = w0 * Bt[2 * p + 1] - w1 * Bt[2 * p];
		const float pb = (-w0) * Bt[2 * p] - w1 * Bt[2 * p + 1];
		const uint32_t q = n4 - 1 - p;
		out[q] = pa;
		out[n2 - 1 - q] = -pa;
		out[n2 + q] = pb;
		out[n - 1 - q] = pb;
	}
}

template <int TPB>
__global__ void __launch_bounds__(LW_BLOCK)
k_imdct_generic(LwDevTables T, LwBatchDev B, float *tap_spec, int use_decoupled, uint32_t skip_mask, uint32_t task_floats)
{
	// use_decoupled: 0 = raw residues (no coupling in the stream), 1 = k_decouple's buffer, 2 = inverse coupling done here,
	// each channel from its own and its partner's raw vector (T.pair_coupling streams: one launch and one round trip of
	// the residues through HBM less)
	extern __shared__ __attribute__((aligned(16))) float smem_all[];
	const uint32_t task = TPB == LW_BLOCK ? blockIdx.x : blockIdx.x * (LW_BLOCK / TPB) + threadIdx.x / TPB;
	if (TPB != LW_BLOCK && B.gen_tasks) { // short blocks with host-packed tasks: one load, then floor record and residues at once
		if (task >= B.n_gen_small * T.ch)
			return;
		const LwGenTask t = B.gen_tasks[task];
		if ((t.rec.flags & skip_mask) || t.rec.bs > LW_SMALL_BS)
			return;
		const bool inl2 = use_decoupled == 2;
		imdct_block<TPB>(T, B, t.rec, t.c, threadIdx.x % TPB, smem_all + (threadIdx.x / TPB) * task_floats,
				B.td + 2u * t.rec.res_off + t.c * (1u << t.rec.bs), tap_spec, use_decoupled == 1, inl2 ? (int)t.partner : -1,
				inl2 ? (int)t.role : 0, global_tables(T, t.rec), (int)t.fl, (int)t.F);
		return;
	}
	const uint32_t *list = TPB == LW_BLOCK ? B.gen_large : B.gen_small;
	const uint32_t n_list = TPB == LW_BLOCK ? B.n_gen_large : B.n_gen_small;
	if (task >= (list ? n_list : B.n_packets) * T.ch)
		return;
	const uint32_t pkt = list ? list[task / T.ch] : task / T.ch, c = task % T.ch;
	const LwPacketRec rec = B.recs[pkt];
	if (rec.flags & skip_mask)
		return;
	if ((rec.bs <= LW_SMALL_BS) != (TPB != LW_BLOCK))
		return; // the other instantiation handles this block size
	float *smem = smem_all + (TPB == LW_BLOCK ? 0u : (threadIdx.x / TPB) * task_floats);
	const bool inl = use_decoupled == 2;
	imdct_block<TPB>(T, B, rec, c, threadIdx.x % TPB, smem, B.td + 2u * rec.res_off + c * (1u << rec.bs), tap_spec,
			use_decoupled == 1, inl ? T.mode_partner[rec.mode * T.ch + c] : -1, inl ? T.mode_role[rec.mode * T.ch + c] : 0,
			global_tables(T, rec));
}

// ---------------------------------------------------------------------------------------------
// window + overlap-add + state + sample conversion, one workgroup per (packet, channel)
// ---------------------------------------------------------------------------------------------
// samples.rs:92-103: x*32768, clamp to [-32768, 32767], truncate toward zero; NaN -> 0
__device__ __forceinline__ int16_t to_i16(float x)
{
	const float t = x * 32768.0f;
	if (t > 32767.0f)
		return 32767;
	if (t < -32768.0f)
		return -32768;
	return (int16_t)(int)t; // in range: truncation toward zero; NaN -> 0 (v_cvt_i32_f32), like Rust `as`
}

template <int FMT>
__global__ void __launch_bounds__(LW_BLOCK) k_ola_generic(LwDevTables T, LwBatchDev B, void *out_v, uint32_t skip_mask)
{
	// one workgroup per PACKET, all channels: the chain of dependent loads (list -> record -> predecessor's record ->
	// data) is paid once per packet, and a stereo short block (2 x 128 samples) fills the 256 threads exactly
	if (B.ola) { // one descriptor = everything this workgroup needs (the planner packed it): one load, then the samples
		if (blockIdx.x >= B.n_gen_ola)
			return;
		const LwOlaDesc o = B.ola[blockIdx.x];
		const uint32_t n = o.n, ls = o.ls, rs = o.rs, re = o.re, plen = o.plen;
		const float *cur0 = B.td + o.cur_off;
		if (o.prev_kind != 0 && rs > ls) {
			const uint32_t m = rs - ls;
			const float *slope = T.bs[(o.flags & LW_RF_SLOPE_BS1) ? 1 : 0].window;
			const float *prev0 = (o.prev_kind == 1 ? B.td : B.state) + o.prev_off;
			const uint32_t prev_stride = o.prev_stride;
			// four samples per thread and step (16-byte loads, one 8- or 16-byte store) when everything is a multiple of four
			// -- it always is for block sizes >= 64; the long blocks next to short ones move 40 KB each through here
			const bool quads = FMT != LW_OUT_I16_INTERLEAVED && ((ls | m | plen | n | prev_stride | o.prev_off | o.cur_off | o.out_off) & 3u) == 0 &&
				((uintptr_t)out_v & 15u) == 0;
			if (quads) {
				const uint32_t mq = m >> 2;
				for (uint32_t q = threadIdx.x; q < mq * T.ch; q += blockDim.x) {
					const uint32_t c = q / mq, i = 4u * (q - c * mq);
					float4 x = *(const float4 *)(cur0 + c * n + ls + i);
					if (i < plen) { // (plen is a multiple of four as well: the four samples are inside together)
						const float4 s = *(const float4 *)(slope + i), r = *(const float4 *)(slope + plen - 4u - i);
						const float4 p = *(const float4 *)(prev0 + c * prev_stride + i);
						x.x = (x.x * s.x) + (p.x * r.w); // audio.rs:1116-1118, slope[plen - 1 - (i + k)] = r[3 - k]
						x.y = (x.y * s.y) + (p.y * r.z);
						x.z = (x.z * s.z) + (p.z * r.y);
						x.w = (x.w * s.w) + (p.w * r.x);
					}
					if (FMT == LW_OUT_I16_PLANAR) {
						const uint32_t lo = (uint32_t)(uint16_t)to_i16(x.x) | ((uint32_t)(uint16_t)to_i16(x.y) << 16);
						const uint32_t hi = (uint32_t)(uint16_t)to_i16(x.z) | ((uint32_t)(uint16_t)to_i16(x.w) << 16);
						*(uint2 *)((int16_t *)out_v + o.out_off + c * m + i) = make_uint2(lo, hi);
					} else {
						*(float4 *)((float *)out_v + o.out_off + c * m + i) = x;
					}
				}
			} else
			for (uint32_t e = threadIdx.x; e < m * T.ch; e += blockDim.x) {
				const uint32_t c = e / m, i = e - c * m;
				float x = cur0[c * n + ls + i];
				if (i < plen)
					x = (x * slope[i]) + (prev0[c * prev_stride + i] * slope[plen - 1 - i]); // audio.rs:1116-1118
				if (FMT == LW_OUT_I16_PLANAR)
					((int16_t *)out_v)[o.out_off + c * m + i] = to_i16(x);
				else if (FMT == LW_OUT_I16_INTERLEAVED)
					((int16_t *)out_v)[o.out_off + i * T.ch + c] = to_i16(x);
				else
					((float *)out_v)[o.out_off + c * m + i] = x;
			}
		}
		if (o.state_out >= 0 && re > rs) { // audio.rs:1121, :1142-1147: the raw (un-windowed) right part
			const uint32_t par = (o.flags & LW_RF_PARITY_OUT) ? 1u : 0u, len = re - rs;
			float *st = B.state + ((size_t)o.state_out * 2 + par) * T.state_stride;
			if (((len | rs | n | o.cur_off | T.state_chan_stride | T.state_stride) & 3u) == 0) {
				const uint32_t lq = len >> 2;
				for (uint32_t q = threadIdx.x; q < lq * T.ch; q += blockDim.x) {
					const uint32_t c = q / lq, i = 4u * (q - c * lq);
					*(float4 *)(st + c * T.state_chan_stride + i) = *(const float4 *)(cur0 + c * n + rs + i);
				}
			} else
			for (uint32_t e = threadIdx.x; e < len * T.ch; e += blockDim.x) {
				const uint32_t c = e / len, i = e - c * len;
				st[c * T.state_chan_stride + i] = cur0[c * n + rs + i];
			}
		}
		return;
	}
	uint32_t pkt = blockIdx.x;
	if (B.gen_ola) {
		if (pkt >= B.n_gen_ola)
			return;
		pkt = B.gen_ola[pkt];
	} else if (pkt >= B.n_packets) {
		return;
	}
	LwPacketRec rec = B.recs[pkt];
	if (rec.flags & LW_RF_TDONLY)
		rec.flags &= (uint8_t)~LW_RF_FAST; // its time-domain block comes from the specialised kernel, the rest happens here
	if (rec.flags & skip_mask)
		return;
	const uint32_t n = 1u << rec.bs;
	const float *cur0 = B.td + 2u * rec.res_off;
	const uint32_t ls = rec.ls, rs = rec.rs, re = rec.re, plen = rec.plen;
	if (rec.prev != -1 && rs > ls) {
		const uint32_t m = rs - ls;
		const float *slope = T.bs[(rec.flags & LW_RF_SLOPE_BS1) ? 1 : 0].window;
		const float *prev0;
		uint32_t prev_stride;
		if (rec.prev >= 0) {
			const LwPacketRec pr = B.recs[rec.prev];
			prev0 = B.td + 2u * pr.res_off + pr.rs;
			prev_stride = 1u << pr.bs;
		} else {
			const uint32_t slot = (uint32_t)(-(rec.prev + 2));
			const uint32_t par = (rec.flags & LW_RF_PARITY_IN) ? 1u : 0u;
			prev0 = B.state + ((size_t)slot * 2 + par) * T.state_stride;
			prev_stride = T.state_chan_stride;
		}
		for (uint32_t e = threadIdx.x; e < m * T.ch; e += blockDim.x) {
			const uint32_t c = e / m, i = e - c * m;
			float x = cur0[c * n + ls + i];
			if (i < plen)
				x = (x * slope[i]) + (prev0[c * prev_stride + i] * slope[plen - 1 - i]); // audio.rs:1116-1118
			if (FMT == LW_OUT_I16_PLANAR)
				((int16_t *)out_v)[rec.out_off + c * m + i] = to_i16(x);
			else if (FMT == LW_OUT_I16_INTERLEAVED)
				((int16_t *)out_v)[rec.out_off + i * T.ch + c] = to_i16(x);
			else
				((float *)out_v)[rec.out_off + c * m + i] = x;
		}
	}
	if (rec.state_out >= 0 && re > rs) { // audio.rs:1121, :1142-1147: the raw (un-windowed) right part
		const uint32_t par = (rec.flags & LW_RF_PARITY_OUT) ? 1u : 0u, len = re - rs;
		float *st = B.state + ((size_t)rec.state_out * 2 + par) * T.state_stride;
		for (uint32_t e = threadIdx.x; e < len * T.ch; e += blockDim.x) {
			const uint32_t c = e / len, i = e - c * len;
			st[c * T.state_chan_stride + i] = cur0[c * n + rs + i];
		}
	}
}

hipError_t lw_launch_generic_imdct(const LwDevTables &T, const LwBatchDev &B, float *tap_spec, hipStream_t st, uint32_t max_n,
		bool any_coupling, bool include_fast)
{
	if (B.n_packets == 0)
		return hipSuccess;
	const uint32_t skip_mask = include_fast ? LW_RF_SKIP : (LW_RF_SKIP | LW_RF_FAST);
	// inverse coupling: inside the transform kernel when every channel is in at most one step (and nobody taps the
	// intermediate vectors), otherwise by k_decouple into B.decoupled first
	const int coupling = !any_coupling ? 0 : (T.pair_coupling && !tap_spec) ? 2 : 1;
	if (coupling == 1 && (!B.gen_small || B.n_gen_small + B.n_gen_large))
		hipLaunchKernelGGL(k_decouple, dim3(B.gen_small ? B.n_gen_small + B.n_gen_large : B.n_packets), dim3(LW_ELEMENTWISE_BLOCK), 0, st, T,
				B, skip_mask);
	static LwPerDeviceOnce once;
	{
		const hipError_t e = once.run([] {
			hipError_t r = hipFuncSetAttribute((const void *)k_imdct_generic<LW_BLOCK>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
			if (r == hipSuccess)
				r = hipFuncSetAttribute((const void *)k_imdct_generic<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
			return r;
		});
		if (e != hipSuccess)
			return e; // the caller reports this launch; the next one tries again
	}
	// large block sizes: one (packet, channel) block per workgroup; small ones: four per workgroup, one per wave
	const uint32_t n_large = (B.gen_large ? B.n_gen_large : B.n_packets) * T.ch;
	const uint32_t n_small = (B.gen_small ? B.n_gen_small : B.n_packets) * T.ch;
	if (max_n > (1u << LW_SMALL_BS) && n_large) {
		const size_t lds = ((size_t)max_n + (LW_XSTRIDE * 4 + 24 + 3) / 4 + 4) * sizeof(float);
		const uint32_t zero = 0u;
		const hipError_t e = lw_launch_k(k_imdct_generic<LW_BLOCK>, dim3(n_large), dim3(LW_BLOCK), lds, st, T, B, tap_spec, coupling,
				skip_mask, zero);
		if (e != hipSuccess)
			return e;
	}
	if (n_small) {
		const uint32_t small_n = std::min(max_n, 1u << LW_SMALL_BS);
		const uint32_t task_floats = (small_n + (LW_XSTRIDE * 4 + 24 + 3) / 4 + 7u) & ~3u;
		const uint32_t per_wg = LW_BLOCK / 64;
		return lw_launch_k(k_imdct_generic<64>, dim3((n_small + per_wg - 1) / per_wg), dim3(LW_BLOCK),
				(size_t)per_wg * task_floats * sizeof(float), st, T, B, tap_spec, coupling, skip_mask, task_floats);
	}
	return hipSuccess;
}

void lw_launch_generic_ola(const LwDevTables &T, const LwBatchDev &B, void *out, int fmt, hipStream_t st, bool include_fast)
{
	if (B.n_packets == 0)
		return;
	const uint32_t skip_mask = include_fast ? LW_RF_SKIP : (LW_RF_SKIP | LW_RF_FAST);
	const dim3 g(B.gen_ola ? B.n_gen_ola : B.n_packets), b(LW_ELEMENTWISE_BLOCK);
	if (g.x == 0)
		return;
	if (fmt == LW_OUT_I16_PLANAR)
		hipLaunchKernelGGL(k_ola_generic<LW_OUT_I16_PLANAR>, g, b, 0, st, T, B, out, skip_mask);
	else if (fmt == LW_OUT_I16_INTERLEAVED)
		hipLaunchKernelGGL(k_ola_generic<LW_OUT_I16_INTERLEAVED>, g, b, 0, st, T, B, out, skip_mask);
	else
		hipLaunchKernelGGL(k_ola_generic<LW_OUT_F32_PLANAR>, g, b, 0, st, T, B, out, skip_mask);
}
