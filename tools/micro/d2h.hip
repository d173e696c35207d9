// How fast does PCM get back to the host?  16.8 MB (one 4096-packet batch of i16 stereo PCM) device -> pinned host memory:
// hipMemcpyAsync (SDMA), two hipMemcpyAsync halves on two streams, and a copy kernel storing straight into the pinned
// buffer (W workgroups of 256 threads, 16-byte stores).
//   hipcc --offload-arch=gfx950 -O3 d2h.hip -o d2h && ./d2h
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void __launch_bounds__(256) k_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
		dst[i] = src[i];
}

static double now()
{
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
	const size_t bytes = 4096ull * 2 * 1024 * 2;
	void *d, *h;
	hipMalloc(&d, bytes);
	hipHostMalloc(&h, bytes, hipHostMallocDefault);
	hipMemset(d, 1, bytes);
	hipStream_t s0, s1;
	hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
	hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
	const int reps = 50;
	for (int mode = 0; mode < 2; mode++) {
		hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s0);
		hipStreamSynchronize(s0);
		const double t0 = now();
		for (int r = 0; r < reps; r++) {
			if (mode == 0) {
				hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s0);
			} else {
				hipMemcpyAsync(h, d, bytes / 2, hipMemcpyDeviceToHost, s0);
				hipMemcpyAsync((char *)h + bytes / 2, (char *)d + bytes / 2, bytes / 2, hipMemcpyDeviceToHost, s1);
			}
		}
		hipStreamSynchronize(s0);
		hipStreamSynchronize(s1);
		const double dt = (now() - t0) / reps;
		printf("%-44s %6.1f us per 16.8 MB = %5.1f GB/s\n", mode == 0 ? "hipMemcpyAsync D2H, one stream" : "two halves on two streams", dt * 1e6,
				bytes / dt * 1e-9);
	}
	for (int wgs : {8, 16, 32, 64, 128, 256, 1024}) {
		hipLaunchKernelGGL(k_copy, dim3(wgs), dim3(256), 0, s0, (const uint4 *)d, (uint4 *)h, bytes / 16);
		hipStreamSynchronize(s0);
		const double t0 = now();
		for (int r = 0; r < reps; r++)
			hipLaunchKernelGGL(k_copy, dim3(wgs), dim3(256), 0, s0, (const uint4 *)d, (uint4 *)h, bytes / 16);
		hipStreamSynchronize(s0);
		const double dt = (now() - t0) / reps;
		printf("copy kernel into pinned host memory, %4d WGs  %6.1f us per 16.8 MB = %5.1f GB/s\n", wgs, dt * 1e6, bytes / dt * 1e-9);
	}
	// host -> device for comparison
	{
		hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s0);
		hipStreamSynchronize(s0);
		const double t0 = now();
		for (int r = 0; r < reps; r++)
			hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s0);
		hipStreamSynchronize(s0);
		const double dt = (now() - t0) / reps;
		printf("%-44s %6.1f us per 16.8 MB = %5.1f GB/s\n", "hipMemcpyAsync H2D, one stream", dt * 1e6, bytes / dt * 1e-9);
	}
	return 0;
}
