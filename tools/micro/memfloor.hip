// What does the memory system allow for ONE launch that reads 8 KiB + writes 4 KiB per wave, 4096 waves (the long-block
// kernel's HBM traffic for a 4096-packet batch)?  Variants: all waves at once (16 waves/WG, 256 WGs), 2 packets per
// wave with prefetch (8 waves/WG), 4 packets per wave (4 waves/WG).  8 buffer sets are rotated (> 256 MiB MALL).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));
template <int PKTS>
__global__ void __launch_bounds__(1024) k(const float4_t *__restrict__ in, uint2_t *__restrict__ out, int waves_per_wg)
{
	const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const unsigned unit0 = (blockIdx.x * waves_per_wg + wave) * PKTS;
	float4_t r[PKTS][8];
#pragma unroll
	for (int p = 0; p < PKTS; p++)
#pragma unroll
		for (int x = 0; x < 8; x++)
			r[p][x] = in[(size_t)(unit0 + p) * 512 + 64 * x + lane];
#pragma unroll
	for (int p = 0; p < PKTS; p++) {
#pragma unroll
		for (int x = 0; x < 8; x++) {
			const float4_t v = r[p][x] * 32768.0f;
			uint2_t o;
			o.x = ((unsigned)(int)v.x & 0xffffu) | ((unsigned)(int)v.y << 16);
			o.y = ((unsigned)(int)v.z & 0xffffu) | ((unsigned)(int)v.w << 16);
			out[(size_t)(unit0 + p) * 512 + 64 * x + lane] = o;
		}
	}
}
int main()
{
	const int NP = 4096, NB = 8;
	std::vector<float4_t *> in(NB);
	std::vector<uint2_t *> out(NB);
	for (int b = 0; b < NB; b++) {
		(void)hipMalloc(&in[b], (size_t)NP * 8192 + (48 << 20)); // padding so that the sets do not alias in the MALL
		(void)hipMalloc(&out[b], (size_t)NP * 4096 + (16 << 20));
		(void)hipMemset(in[b], 0, (size_t)NP * 8192);
	}
	hipStream_t st;
	(void)hipStreamCreate(&st);
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	for (int variant = 0; variant < 6; variant++) {
		const int pk = variant == 1 ? 2 : (variant == 2 ? 4 : 1);
		const int wpw = variant == 3 ? 4 : (variant == 4 ? 1 : (variant == 5 ? 8 : 16 / pk)); // waves per workgroup
		const int grid = 4096 / (pk * wpw);
		hipGraph_t g;
		hipGraphExec_t ge;
		(void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
		for (int b = 0; b < NB; b++) {
			if (pk == 1)
				hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64 * wpw), 0, st, in[b], out[b], wpw);
			else if (pk == 2)
				hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64 * wpw), 0, st, in[b], out[b], wpw);
			else
				hipLaunchKernelGGL(k<4>, dim3(grid), dim3(64 * wpw), 0, st, in[b], out[b], wpw);
		}
		(void)hipStreamEndCapture(st, &g);
		(void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
		for (int w = 0; w < 3; w++)
			(void)hipGraphLaunch(ge, st);
		(void)hipStreamSynchronize(st);
		(void)hipEventRecord(e0, st);
		const int REP = 20;
		for (int w = 0; w < REP; w++)
			(void)hipGraphLaunch(ge, st);
		(void)hipEventRecord(e1, st);
		(void)hipEventSynchronize(e1);
		float ms;
		(void)hipEventElapsedTime(&ms, e0, e1);
		const double us = ms * 1e3 / (REP * NB);
		printf("grid %4d packets/wave %d (waves/WG %2d): %.2f us per 4096-packet launch -> %.0f GB/s of 50.3 MB\n", grid, pk, wpw, us, 50.33e6 / us / 1e3);
	}
	return 0;
}
