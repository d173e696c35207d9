// Straight-line vs looped VALU issue rate on gfx950: is a long unrolled kernel (each instruction executed once per
// wave) limited by instruction fetch rather than by the VALU?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_t __attribute__((ext_vector_type(2)));
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R256(x) R16(R16(x))
#define BODY "v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4\n\t" \
             "v_pk_add_f32 %0, %0, %5\n\tv_pk_add_f32 %1, %1, %5\n\tv_pk_add_f32 %2, %2, %5\n\tv_pk_add_f32 %3, %3, %5\n\t"
template <int MODE>
__global__ void k(unsigned long long *out, float seed, int reps)
{
	float2_t a = {seed, seed + 1}, b = {seed + 2, seed + 3}, c = {seed, seed}, d = {seed, seed}, e = {1.0f, 1.0f}, f = {0.f, 0.f};
	unsigned long long m0, m1;
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(m0));
	if (MODE == 0) { // 2048 instructions straight line, executed `reps` times
		for (int r = 0; r < reps; r++)
			asm volatile(R256(BODY) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
	} else { // loop of 8 instructions, 256 * reps iterations
		for (int i = 0; i < 256 * reps; i++)
			asm volatile(BODY : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
	}
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(m1));
	if (a.x + b.x + c.x + d.x == 12345.f)
		out[9] = 1;
	if (threadIdx.x == 0 && blockIdx.x == 0)
		out[0] = m1 - m0;
}
int main()
{
	unsigned long long *d, h[16];
	(void)hipMalloc(&d, 128);
	for (int reps = 1; reps <= 4; reps *= 4)
		for (int grid = 1; grid <= 256; grid *= 256)
			for (int waves = 1; waves <= 16; waves *= 4)
				for (int mode = 0; mode < 2; mode++) {
					for (int rep = 0; rep < 3; rep++) {
						if (mode == 0)
							hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64 * waves), 0, 0, d, 1.0f, reps);
						else
							hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64 * waves), 0, 0, d, 1.0f, reps);
						(void)hipDeviceSynchronize();
					}
					(void)hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
					printf("grid %3d waves/block %2d reps %d %-13s %6.2f cycles per wave-instruction\n", grid, waves, reps,
							mode ? "loop(8)" : "straight(2048)", h[0] / (2048.0 * reps));
				}
	return 0;
}
