// Do VALU and LDS work overlap on a gfx950 CU?  16 waves; each repeats a block of 64 packed-f32 ops (V), 16 LDS ops (L),
// or both interleaved 4:1 (VL).  span(VL) ~ max(V, L) means the pipes run concurrently, ~ V + L means they serialise.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
typedef float float2_t __attribute__((ext_vector_type(2)));
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define V4 "v_pk_mul_f32 %0, %0, %[m]\n\tv_pk_add_f32 %1, %1, %[z]\n\tv_pk_mul_f32 %2, %2, %[m]\n\tv_pk_add_f32 %3, %3, %[z]\n\t"
#define LW "ds_write_b64 %[ad], %[d] offset:0\n\t"
#define LR "ds_read_b64 %[q], %[ad] offset:0\n\t"
#define LR4 "ds_read_b128 %[q4], %[ad] offset:0\n\t"
#define LB "ds_bpermute_b32 %[b], %[ba], %[d0]\n\t"
template <int MODE, int LK>
__global__ void k(unsigned long long *out, float seed, int reps)
{
	__shared__ float2_t buf[16 * 64 * 2];
	float2_t a = {seed, seed}, b = a, c = a, d = a, m = {1.0f, 1.0f}, z = {0.f, 0.f}, q = a;
	float4 q4;
	float bp = seed;
	const uint32_t addr = (uint32_t)(uintptr_t)(&buf[threadIdx.x * (LK == 2 ? 2 : 1)]);
	const uint32_t bpa = ((threadIdx.x * 4) ^ 32) & 255;
	__syncthreads();
	unsigned long long t0, t1;
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
	for (int r = 0; r < reps; r++) {
#define OPS : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(a), "+v"(b), "+v"(c), "+v"(d), [q] "=v"(q), [q4] "=v"(q4), [b] "=v"(bp) : "v"(m), "v"(z), "v"(addr), "v"(bpa), [d] "v"(m) : "memory"
#define FIN "s_waitcnt lgkmcnt(0)"
		if (MODE == 0) asm volatile(R16(V4) FIN :"+v"(a), "+v"(b), "+v"(c), "+v"(d), [q] "=v"(q), [q4] "=v"(q4), [b] "=v"(bp) : [m] "v"(m), [z] "v"(z), [ad] "v"(addr), [ba] "v"(bpa), [d] "v"(m), [d0] "v"(seed) : "memory");
		else if (MODE == 1) {
			if (LK == 0) asm volatile(R16(LW) FIN :"+v"(a), "+v"(b), "+v"(c), "+v"(d), [q] "=v"(q), [q4] "=v"(q4), [b] "=v"(bp) : [m] "v"(m), [z] "v"(z), [ad] "v"(addr), [ba] "v"(bpa), [d] "v"(m), [d0] "v"(seed) : "memory");
			else if (LK == 1) asm volatile(R16(LR) FIN :"+v"(a), "+v"(b), "+v"(c), "+v"(d), [q] "=v"(q), [q4] "=v"(q4), [b] "=v"(bp) : [m] "v"(m), [z] "v"(z), [ad] "v"(addr), [ba] "v"(bpa), [d] "v"(m), [d0] "v"(seed) : "memory");
			else if (LK == 2) asm volatile(R16(LR4) FIN :"+v"(a), "+v"(b), "+v"(c), "+v"(d), [q] "=v"(q), [q4] "=v"(q4), [b] "=v"(bp) : [m] "v"(m), [z] "v"(z), [ad] "v"(addr), [ba] "v"(bpa), [d] "v"(m), [d0] "v"(seed) : "memory");
			else asm volatile(R16(LB) FIN :"+v"(a), "+v"(b), "+v"(c), "+v"(d), [q] "=v"(q), [q4] "=v"(q4), [b] "=v"(bp) : [m] "v"(m), [z] "v"(z), [ad] "v"(addr), [ba] "v"(bpa), [d] "v"(m), [d0] "v"(seed) : "memory");
		} else {
			if (LK == 0) asm volatile(R16(V4 LW) FIN :"+v"(a), "+v"(b), "+v"(c), "+v"(d), [q] "=v"(q), [q4] "=v"(q4), [b] "=v"(bp) : [m] "v"(m), [z] "v"(z), [ad] "v"(addr), [ba] "v"(bpa), [d] "v"(m), [d0] "v"(seed) : "memory");
			else if (LK == 1) asm volatile(R16(V4 LR) FIN :"+v"(a), "+v"(b), "+v"(c), "+v"(d), [q] "=v"(q), [q4] "=v"(q4), [b] "=v"(bp) : [m] "v"(m), [z] "v"(z), [ad] "v"(addr), [ba] "v"(bpa), [d] "v"(m), [d0] "v"(seed) : "memory");
			else if (LK == 2) asm volatile(R16(V4 LR4) FIN :"+v"(a), "+v"(b), "+v"(c), "+v"(d), [q] "=v"(q), [q4] "=v"(q4), [b] "=v"(bp) : [m] "v"(m), [z] "v"(z), [ad] "v"(addr), [ba] "v"(bpa), [d] "v"(m), [d0] "v"(seed) : "memory");
			else asm volatile(R16(V4 LB) FIN :"+v"(a), "+v"(b), "+v"(c), "+v"(d), [q] "=v"(q), [q4] "=v"(q4), [b] "=v"(bp) : [m] "v"(m), [z] "v"(z), [ad] "v"(addr), [ba] "v"(bpa), [d] "v"(m), [d0] "v"(seed) : "memory");
		}
	}
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
	if (a.x + b.x + c.x + d.x + q.x + q4.x + bp == 12345.f)
		out[1000] = 1;
	if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) {
		out[2 * (threadIdx.x >> 6)] = t0;
		out[2 * (threadIdx.x >> 6) + 1] = t1;
	}
}
template <int MODE, int LK>
static unsigned long long run(unsigned long long *d, int waves)
{
	unsigned long long h[64];
	for (int rep = 0; rep < 3; rep++) {
		hipLaunchKernelGGL((k<MODE, LK>), dim3(1), dim3(64 * waves), 0, 0, d, 1.0f, 8);
		(void)hipDeviceSynchronize();
	}
	(void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
	unsigned long long lo = ~0ull, hi = 0;
	for (int w = 0; w < waves; w++) { lo = std::min(lo, h[2 * w]); hi = std::max(hi, h[2 * w + 1]); }
	return hi - lo;
}
int main()
{
	unsigned long long *d;
	(void)hipMalloc(&d, 16384);
	const char *names[4] = {"ds_write_b64", "ds_read_b64", "ds_read_b128", "ds_bpermute_b32"};
	for (int waves = 4; waves <= 16; waves *= 4) {
		const unsigned long long v = run<0, 0>(d, waves);
		unsigned long long l[4] = {run<1, 0>(d, waves), run<1, 1>(d, waves), run<1, 2>(d, waves), run<1, 3>(d, waves)};
		unsigned long long vl[4] = {run<2, 0>(d, waves), run<2, 1>(d, waves), run<2, 2>(d, waves), run<2, 3>(d, waves)};
		for (int i = 0; i < 4; i++)
			printf("%2d waves/CU: 8 x (64 pk + 16 %-16s): V %6llu  L %6llu  V+L interleaved %6llu cycles (max %llu, sum %llu)\n", waves, names[i],
					v, l[i], vl[i], std::max(v, l[i]), v + l[i]);
	}
	return 0;
}
