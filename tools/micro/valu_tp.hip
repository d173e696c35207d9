// Aggregate VALU throughput per SIMD on gfx950: W waves per SIMD each issue N independent packed / plain f32 ops;
// cycles per instruction per SIMD = (t_last_end - t_first_start) / (W * N).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
typedef float float2_t __attribute__((ext_vector_type(2)));
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define PK "v_pk_mul_f32 %0, %0, %8\n\tv_pk_mul_f32 %1, %1, %8\n\tv_pk_mul_f32 %2, %2, %8\n\tv_pk_mul_f32 %3, %3, %8\n\t" \
           "v_pk_add_f32 %4, %4, %9\n\tv_pk_add_f32 %5, %5, %9\n\tv_pk_add_f32 %6, %6, %9\n\tv_pk_add_f32 %7, %7, %9\n\t"
#define SC "v_mul_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\t" \
           "v_add_f32 %4, %4, %9\n\tv_add_f32 %5, %5, %9\n\tv_add_f32 %6, %6, %9\n\tv_add_f32 %7, %7, %9\n\t"
#define MV "v_mov_b32 %0, %8\n\tv_mov_b32 %1, %8\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %8\n\t" \
           "v_mov_b32 %4, %9\n\tv_mov_b32 %5, %9\n\tv_mov_b32 %6, %9\n\tv_mov_b32 %7, %9\n\t"
template <int MODE>
__global__ void k(unsigned long long *out, float seed, int reps)
{
	float2_t a = {seed, seed}, b = a, c = a, d = a, e = a, f = a, g = a, h = a, m = {1.0f, 1.0f}, z = {0.f, 0.f};
	float sa = seed, sb = seed, sc = seed, sd = seed, se = seed, sf = seed, sg = seed, sh = seed;
	__syncthreads();
	unsigned long long t0, t1;
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
	for (int r = 0; r < reps; r++) {
		if (MODE == 0)
			asm volatile(R64(PK) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(m), "v"(z));
		else if (MODE == 1)
			asm volatile(R64(SC) : "+v"(sa), "+v"(sb), "+v"(sc), "+v"(sd), "+v"(se), "+v"(sf), "+v"(sg), "+v"(sh) : "v"(m.x), "v"(z.x));
		else
			asm volatile(R64(MV) : "+v"(sa), "+v"(sb), "+v"(sc), "+v"(sd), "+v"(se), "+v"(sf), "+v"(sg), "+v"(sh) : "v"(m.x), "v"(z.x));
	}
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
	if (a.x + b.x + c.x + d.x + e.x + f.x + g.x + h.x + sa + sb + sc + sd + se + sf + sg + sh == 12345.f)
		out[1000] = 1;
	if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) {
		out[2 * (threadIdx.x >> 6)] = t0;
		out[2 * (threadIdx.x >> 6) + 1] = t1;
	}
}
int main()
{
	unsigned long long *d, h[64];
	(void)hipMalloc(&d, 16384);
	const char *names[3] = {"v_pk_mul/add_f32", "v_mul/add_f32", "v_mov_b32"};
	for (int waves = 4; waves <= 16; waves *= 2)
		for (int mode = 0; mode < 3; mode++) {
			const int reps = 4;
			for (int rep = 0; rep < 3; rep++) {
				if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64 * waves), 0, 0, d, 1.0f, reps);
				else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64 * waves), 0, 0, d, 1.0f, reps);
				else hipLaunchKernelGGL(k<2>, dim3(1), dim3(64 * waves), 0, 0, d, 1.0f, reps);
				(void)hipDeviceSynchronize();
			}
			(void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
			unsigned long long lo = ~0ull, hi = 0;
			for (int w = 0; w < waves; w++) { lo = std::min(lo, h[2 * w]); hi = std::max(hi, h[2 * w + 1]); }
			const double n = 512.0 * reps; // instructions per wave
			printf("%2d waves/CU (%d per SIMD) %-18s span %8llu cycles -> %.2f cycles per instruction per SIMD, wave0 alone %.2f\n", waves, waves / 4,
					names[mode], hi - lo, (double)(hi - lo) / (n * waves / 4), (double)(h[1] - h[0]) / n);
		}
	return 0;
}
