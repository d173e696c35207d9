// probe: is the speed of a device-to-host copy a property of the STREAM it is issued on?  N streams, one 16.8 MB copy at a time
// on each in turn (HIP events), then two at a time on every pair of neighbours.
//   hipcc --offload-arch=gfx950 -O2 -o d2h_streams d2h_streams.hip && GPU_MAX_HW_QUEUES=8 ./d2h_streams [streams]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x)                                                                          \
	do {                                                                               \
		hipError_t e_ = (x);                                                           \
		if (e_ != hipSuccess) {                                                        \
			printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);         \
			return 1;                                                                  \
		}                                                                              \
	} while (0)
__global__ void touch(char *p, size_t n)
{
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		p[i] = (char)i;
}
__global__ void spin(char *p, int n)
{
	int acc = 0;
	for (int i = 0; i < n; i++)
		acc += __builtin_amdgcn_readfirstlane(i) ^ acc;
	if (acc == 12345)
		p[0] = 1;
}
int main(int argc, char **argv)
{
	const int N = argc > 1 ? atoi(argv[1]) : 8;
	const size_t B = 4096ull * 4096; // one batch of i16 PCM
	std::vector<hipStream_t> st(N);
	std::vector<char *> d(N), h(N);
	std::vector<hipEvent_t> e0(N), e1(N);
	for (int i = 0; i < N; i++) {
		CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
		CK(hipMalloc(&d[i], B));
		CK(hipHostMalloc(&h[i], B));
		CK(hipEventCreate(&e0[i]));
		CK(hipEventCreate(&e1[i]));
		touch<<<1024, 256, 0, st[i]>>>(d[i], B); // (a kernel first: the stream owns a hardware queue before it copies)
		CK(hipStreamSynchronize(st[i]));
	}
	for (int round = 0; round < 4; round++) {
		printf("one at a time, round %d:", round);
		for (int i = 0; i < N; i++) {
			CK(hipEventRecord(e0[i], st[i]));
			CK(hipMemcpyAsync(h[i], d[i], B, hipMemcpyDeviceToHost, st[i]));
			CK(hipEventRecord(e1[i], st[i]));
			CK(hipStreamSynchronize(st[i]));
			float ms;
			CK(hipEventElapsedTime(&ms, e0[i], e1[i]));
			printf(" %4.0f", ms * 1e3);
		}
		printf(" us\n");
	}
	for (int round = 0; round < 3; round++) {
		printf("two at a time (i, i+1), round %d:", round);
		for (int i = 0; i + 1 < N; i++) {
			for (int k = i; k < i + 2; k++) {
				CK(hipEventRecord(e0[k], st[k]));
				CK(hipMemcpyAsync(h[k], d[k], B, hipMemcpyDeviceToHost, st[k]));
				CK(hipEventRecord(e1[k], st[k]));
			}
			float a, b;
			CK(hipStreamSynchronize(st[i]));
			CK(hipStreamSynchronize(st[i + 1]));
			CK(hipEventElapsedTime(&a, e0[i], e1[i]));
			CK(hipEventElapsedTime(&b, e0[i + 1], e1[i + 1]));
			printf(" %4.0f/%-4.0f", a * 1e3, b * 1e3);
		}
		printf(" us\n");
	}
	// a copy queued behind a kernel of its own stream, and a READY copy on another stream issued after it
	for (int rep = 0; rep < 6; rep++) {
		spin<<<256, 256, 0, st[0]>>>(d[0], 600000 >> (rep & 1)); // ~ hundreds of us
		CK(hipEventRecord(e0[0], st[0]));
		CK(hipMemcpyAsync(h[0], d[0], B, hipMemcpyDeviceToHost, st[0]));
		CK(hipEventRecord(e1[0], st[0]));
		for (int k = 1; k < 3; k++) {
			CK(hipEventRecord(e0[k], st[k]));
			CK(hipMemcpyAsync(h[k], d[k], B, hipMemcpyDeviceToHost, st[k]));
			CK(hipEventRecord(e1[k], st[k]));
		}
		float a, b, c, w, x;
		for (int k = 0; k < 3; k++)
			CK(hipStreamSynchronize(st[k]));
		CK(hipEventElapsedTime(&a, e0[0], e1[0]));
		CK(hipEventElapsedTime(&b, e0[1], e1[1]));
		CK(hipEventElapsedTime(&c, e0[2], e1[2]));
		CK(hipEventElapsedTime(&w, e0[1], e0[0])); // when the kernel ended, relative to the issue of the ready copies
		CK(hipEventElapsedTime(&x, e0[1], e1[0]));
		printf("kernel then copy on stream 0 (kernel ends at %4.0f, its copy at %4.0f), ready copies on streams 1, 2 take %4.0f / %4.0f us (stream 0's: %4.0f)\n",
		       w * 1e3, x * 1e3, b * 1e3, c * 1e3, a * 1e3);
	}
	// three in flight: the third issued while two run
	printf("three at a time (0, 1, 2):");
	for (int rep = 0; rep < 4; rep++) {
		for (int k = 0; k < 3; k++) {
			CK(hipEventRecord(e0[k], st[k]));
			CK(hipMemcpyAsync(h[k], d[k], B, hipMemcpyDeviceToHost, st[k]));
			CK(hipEventRecord(e1[k], st[k]));
		}
		for (int k = 0; k < 3; k++) {
			float a;
			CK(hipStreamSynchronize(st[k]));
			CK(hipEventElapsedTime(&a, e0[k], e1[k]));
			printf(" %4.0f", a * 1e3);
		}
		printf(" |");
	}
	printf(" us\n");
	return 0;
}
