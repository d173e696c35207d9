// probe: which compute units a CU-masked HIP stream (hipExtStreamCreateWithCUMask) runs on, read off the hardware ids of the
// waves themselves, and what two such streams do to each other's kernels.  gfx950 only.
//   hipcc --offload-arch=gfx950 -O2 -o cumask cumask.hip && ./cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
#include <chrono>

#define CK(x)                                                                              \
	do {                                                                                   \
		hipError_t e_ = (x);                                                               \
		if (e_ != hipSuccess) {                                                            \
			printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);             \
			return 1;                                                                      \
		}                                                                                  \
	} while (0)

__global__ void who(uint32_t *out, uint32_t spin)
{
	extern __shared__ char lds[];
	uint32_t xcc, hw;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
	uint32_t acc = 0;
	for (uint32_t i = 0; i < spin; i++) {
		lds[threadIdx.x] = (char)i;
		acc += lds[(threadIdx.x + 1) & 63];
	}
	if (threadIdx.x == 0)
		out[blockIdx.x] = (xcc & 15u) << 16 | (hw & 0xff00u) | (acc & 1u); // xcc | se sh cu
}

// a stand-in for a whole-CU kernel (k_long: 152 KB of LDS, 1024 threads) that takes `spin` rounds
__global__ void __launch_bounds__(1024) hog(uint32_t *out, uint32_t spin)
{
	extern __shared__ char lds[];
	uint32_t acc = 0;
	for (uint32_t i = 0; i < spin; i++) {
		lds[threadIdx.x] = (char)i;
		acc += lds[(threadIdx.x + 1) & 1023];
	}
	if (threadIdx.x == 0)
		out[blockIdx.x] = acc;
}

static int describe(const char *name, hipStream_t s, uint32_t *d_out, std::vector<uint32_t> &h)
{
	const uint32_t n = 4096;
	CK(hipMemsetAsync(d_out, 0xff, n * 4, s));
	who<<<n, 64, 32768, s>>>(d_out, 2000);
	CK(hipStreamSynchronize(s));
	h.resize(n);
	CK(hipMemcpy(h.data(), d_out, n * 4, hipMemcpyDeviceToHost));
	std::set<uint32_t> cus, xccs;
	for (uint32_t v : h) {
		cus.insert(v & ~1u);
		xccs.insert(v >> 16);
	}
	printf("%-28s %3zu CUs on XCCs {", name, cus.size());
	for (uint32_t x : xccs)
		printf(" %u", x);
	printf(" }\n");
	return 0;
}

int main()
{
	int n_cus = 0;
	CK(hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, 0));
	printf("device 0: %d CUs\n", n_cus);
	const int words = (n_cus + 31) / 32;
	uint32_t *d_out;
	CK(hipMalloc(&d_out, 1 << 20));
	std::vector<uint32_t> h;
	hipStream_t plain;
	CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
	if (describe("no mask", plain, d_out, h))
		return 1;
	struct M {
		const char *name;
		int mod, lo, hi; // bit i set when lo <= (mod ? i % mod : i) < hi
	} masks[] = {{"bits 0..127", 0, 0, 128}, {"bits 128..255", 0, 128, 256}, {"bits with i%8 < 4", 8, 0, 4},
		{"bits with i%8 >= 4", 8, 4, 8}, {"bits with i%8 == 0", 8, 0, 1}, {"bits 0..31", 0, 0, 32}};
	std::vector<hipStream_t> st;
	for (const M &m : masks) {
		std::vector<uint32_t> mask(words, 0);
		for (int i = 0; i < n_cus; i++) {
			const int k = m.mod ? i % m.mod : i;
			if (k >= m.lo && k < m.hi)
				mask[i / 32] |= 1u << (i % 32);
		}
		hipStream_t s;
		CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask.data()));
		st.push_back(s);
		if (describe(m.name, s, d_out, h))
			return 1;
	}
	// two tenants: a long-running small-LDS kernel on one stream, whole-CU workgroups on the other
	auto run_pair = [&](const char *name, hipStream_t a, hipStream_t b) -> int {
		hipEvent_t e0, e1;
		CK(hipEventCreate(&e0));
		CK(hipEventCreate(&e1));
		for (int rep = 0; rep < 2; rep++) {
			who<<<16384, 256, 16384, a>>>(d_out, 6000); // the foreign tenant: ~ hundreds of us
			CK(hipEventRecord(e0, b));
			hog<<<256, 1024, 152 * 1024, b>>>(d_out + 65536, 300);
			CK(hipEventRecord(e1, b));
			CK(hipStreamSynchronize(a));
			CK(hipStreamSynchronize(b));
		}
		float ms = 0;
		CK(hipEventElapsedTime(&ms, e0, e1));
		printf("%-44s whole-CU kernel next to a foreign kernel: %.1f us\n", name, ms * 1e3);
		return 0;
	};
	CK(hipFuncSetAttribute((const void *)hog, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
	{
		hipEvent_t e0, e1;
		CK(hipEventCreate(&e0));
		CK(hipEventCreate(&e1));
		for (int rep = 0; rep < 2; rep++) {
			CK(hipEventRecord(e0, plain));
			hog<<<256, 1024, 152 * 1024, plain>>>(d_out + 65536, 300);
			CK(hipEventRecord(e1, plain));
			CK(hipStreamSynchronize(plain));
		}
		float ms = 0;
		CK(hipEventElapsedTime(&ms, e0, e1));
		printf("%-44s whole-CU kernel alone, 256 workgroups: %.1f us\n", "no mask", ms * 1e3);
		for (int rep = 0; rep < 2; rep++) {
			CK(hipEventRecord(e0, st[2]));
			hog<<<256, 1024, 152 * 1024, st[2]>>>(d_out + 65536, 300);
			CK(hipEventRecord(e1, st[2]));
			CK(hipStreamSynchronize(st[2]));
		}
		CK(hipEventElapsedTime(&ms, e0, e1));
		printf("%-44s whole-CU kernel alone, 256 workgroups: %.1f us\n", "half the CUs (i%8 < 4)", ms * 1e3);
	}
	hipStream_t plain2;
	CK(hipStreamCreateWithFlags(&plain2, hipStreamNonBlocking));
	if (run_pair("two unmasked streams", plain, plain2))
		return 1;
	if (run_pair("masks i%8 < 4 / i%8 >= 4", st[2], st[3]))
		return 1;
	if (run_pair("masks bits 0..127 / 128..255", st[0], st[1]))
		return 1;
	return 0;
}
