// Dependent-load latency as k_entropy sees it: W workgroups of one wave, every lane chases its own pointer chain through a
// table of `kb` KB (L1 / L2 / HBM resident by size), `steps` dependent loads.  Prints ns and cycles per load.
//   hipcc --offload-arch=gfx950 -O3 chase.hip -o chase && ./chase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void __launch_bounds__(64) k_chase(const uint32_t *tab, uint32_t mask, uint32_t steps, uint32_t lanes, uint32_t *out, long long *cyc)
{
	if (threadIdx.x >= lanes)
		return;
	uint32_t p = (blockIdx.x * 64 + threadIdx.x) * 2654435761u & mask;
	const long long t0 = __builtin_readcyclecounter();
	for (uint32_t s = 0; s < steps; s++)
		p = tab[p] & mask;
	const long long t1 = __builtin_readcyclecounter();
	out[blockIdx.x * 64 + threadIdx.x] = p;
	if (threadIdx.x == 0)
		cyc[blockIdx.x] = t1 - t0;
}

int main()
{
	const uint32_t steps = 2000;
	uint32_t *d_out;
	long long *d_cyc;
	hipMalloc(&d_out, 4096 * 64 * 4);
	hipMalloc(&d_cyc, 4096 * 8);
	for (uint32_t kb : {16u, 256u, 2048u, 65536u}) {
		const uint32_t n = kb * 256;
		std::vector<uint32_t> h(n);
		uint32_t x = 12345;
		for (uint32_t i = 0; i < n; i++) {
			x = x * 1664525u + 1013904223u;
			h[i] = x >> 4;
		}
		uint32_t *d_tab;
		hipMalloc(&d_tab, n * 4);
		hipMemcpy(d_tab, h.data(), n * 4, hipMemcpyHostToDevice);
		for (uint32_t wgs : {64u, 1024u})
			for (uint32_t lanes : {1u, 64u}) {
				hipEvent_t e0, e1;
				hipEventCreate(&e0);
				hipEventCreate(&e1);
				for (int rep = 0; rep < 2; rep++) {
					hipEventRecord(e0, 0);
					hipLaunchKernelGGL(k_chase, dim3(wgs), dim3(64), 0, 0, d_tab, n - 1, steps, lanes, d_out, d_cyc);
					hipEventRecord(e1, 0);
					hipEventSynchronize(e1);
				}
				float ms;
				hipEventElapsedTime(&ms, e0, e1);
				long long c;
				hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
				printf("table %6u KB  %4u waves x %2u lanes: %7.1f ns per dependent load (%lld s_memtime ticks per load)\n", kb, wgs, lanes,
						ms * 1e6 / steps, c / steps);
			}
		hipFree(d_tab);
	}
	return 0;
}
