// Calibrates s_memtime against the wall clock and measures issue cost of v_pk_mul_f32 / v_mul_f32 / LDS ops (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(unsigned long long ticks, unsigned long long *out)
{
	unsigned long long t0 = __builtin_readcyclecounter(), t;
	unsigned long long m0, m1;
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(m0));
	do {
		asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(m1));
	} while (m1 - m0 < ticks);
	t = __builtin_readcyclecounter();
	if (threadIdx.x == 0) {
		out[0] = m1 - m0;
		out[1] = t - t0;
	}
}
typedef float float2_t __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void issue(unsigned long long *out, float seed)
{
	float2_t a = {seed, seed + 1}, b = {seed + 2, seed + 3}, c = {seed, seed}, d = {seed, seed};
	float2_t e = a, f = b, g = c, h = d;
	__shared__ float2_t lds[2048];
	lds[threadIdx.x] = a;
	__syncthreads();
	unsigned long long m0, m1;
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(m0));
	for (int i = 0; i < 256; i++) {
		if (MODE == 0) { // 8 independent pk_mul
			asm volatile("v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4"
					: "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));
			asm volatile("v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4"
					: "+v"(f), "+v"(g), "+v"(h), "+v"(e) : "v"(a));
		} else if (MODE == 1) { // 8 independent v_mul_f32
			asm volatile("v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %3, %3, %4"
					: "+v"(a.x), "+v"(b.x), "+v"(c.x), "+v"(d.x) : "v"(e.x));
			asm volatile("v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %3, %3, %4"
					: "+v"(f.x), "+v"(g.x), "+v"(h.x), "+v"(e.x) : "v"(a.x));
		} else if (MODE == 2) { // 8 ds_read_b64
			float2_t r0, r1, r2, r3, r4, r5, r6, r7;
			unsigned addr = threadIdx.x * 8;
			asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:512\n\tds_read_b64 %2, %8 offset:1024\n\tds_read_b64 %3, %8 offset:1536\n\t"
			             "ds_read_b64 %4, %8 offset:2048\n\tds_read_b64 %5, %8 offset:2560\n\tds_read_b64 %6, %8 offset:3072\n\tds_read_b64 %7, %8 offset:3584\n\ts_waitcnt lgkmcnt(0)"
					: "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(addr));
			a += r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
		} else if (MODE == 3) { // 8 ds_write_b64
			unsigned addr = threadIdx.x * 8;
			asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %1 offset:512\n\tds_write_b64 %0, %1 offset:1024\n\tds_write_b64 %0, %1 offset:1536\n\t"
			             "ds_write_b64 %0, %1 offset:2048\n\tds_write_b64 %0, %1 offset:2560\n\tds_write_b64 %0, %1 offset:3072\n\tds_write_b64 %0, %1 offset:3584\n\ts_waitcnt lgkmcnt(0)"
					:: "v"(addr), "v"(a) : "memory");
		} else if (MODE == 4) { // 8 ds_read_b128
			float4 r0, r1, r2, r3, r4, r5, r6, r7;
			unsigned addr = threadIdx.x * 16;
			asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %8 offset:3072\n\t"
			             "ds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\tds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\ts_waitcnt lgkmcnt(0)"
					: "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(addr));
			a.x += r0.x + r1.x + r2.x + r3.x + r4.x + r5.x + r6.x + r7.x;
		} else if (MODE == 5) { // 8 ds_bpermute
			int r0 = threadIdx.x, ad = ((63 - threadIdx.x) & 63) << 2;
			for (int k = 0; k < 8; k++)
				r0 = __builtin_amdgcn_ds_bpermute(ad, r0);
			a.x += r0;
		}
	}
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(m1));
	if (a.x + b.x + c.x + d.x + e.x + f.x + g.x + h.x == 12345.f)
		out[9] = 1;
	if (threadIdx.x == 0 && blockIdx.x == 0)
		out[0] = m1 - m0;
}
int main()
{
	unsigned long long *d, h[16];
	hipMalloc(&d, 128);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	for (int rep = 0; rep < 2; rep++) {
		hipEventRecord(e0);
		hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, 2000000ull, d);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
		printf("spin: %llu memtime ticks, %llu readcyclecounter ticks in %.3f ms -> memtime %.1f MHz, cyclecounter %.1f MHz\n", h[0], h[1], ms,
				h[0] / ms / 1e3, h[1] / ms / 1e3);
	}
	const char *names[6] = {"v_pk_mul_f32", "v_mul_f32", "ds_read_b64", "ds_write_b64", "ds_read_b128", "ds_bpermute_b32"};
	for (int waves = 1; waves <= 32; waves *= 2) {
		const int grid = waves == 32 ? 512 : 1, thr = waves == 32 ? 1024 : 64 * waves; // 32: two 16-wave blocks per CU
		for (int mode = 0; mode < 6; mode++) {
			for (int rep = 0; rep < 2; rep++) {
				switch (mode) {
				case 0: hipLaunchKernelGGL(issue<0>, dim3(grid), dim3(thr), 0, 0, d, 1.0f); break;
				case 1: hipLaunchKernelGGL(issue<1>, dim3(grid), dim3(thr), 0, 0, d, 1.0f); break;
				case 2: hipLaunchKernelGGL(issue<2>, dim3(grid), dim3(thr), 0, 0, d, 1.0f); break;
				case 3: hipLaunchKernelGGL(issue<3>, dim3(grid), dim3(thr), 0, 0, d, 1.0f); break;
				case 4: hipLaunchKernelGGL(issue<4>, dim3(grid), dim3(thr), 0, 0, d, 1.0f); break;
				case 5: hipLaunchKernelGGL(issue<5>, dim3(grid), dim3(thr), 0, 0, d, 1.0f); break;
				}
				hipDeviceSynchronize();
			}
			hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
			printf("waves/CU %2d  %-16s %7.2f memtime ticks per wave-instruction (wave 0, 2048 instr)\n", waves, names[mode], h[0] / 2048.0);
		}
	}
	return 0;
}
