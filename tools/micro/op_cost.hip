// Issue cost per SIMD of the individual instructions k_long is made of (gfx950): 16 waves per CU (4 per SIMD) each issue
// 8 independent chains x 64 x reps copies of ONE instruction; cycles per instruction per SIMD = span / (4 waves x N).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/op_cost.hip -o tools/micro/op_cost && tools/micro/op_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
typedef float float2_t __attribute__((ext_vector_type(2)));
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
// I(d, s): one instruction writing d, reading d / s (+ %8 / %9 / %10 = constants); eight independent destinations
#define OPS(I) I("%0", "%1") I("%1", "%2") I("%2", "%3") I("%3", "%4") I("%4", "%5") I("%5", "%6") I("%6", "%7") I("%7", "%0")
#define I_MOV(d, s) "v_mov_b32 " d ", %8\n\t"
#define I_ADD(d, s) "v_add_f32 " d ", " d ", %8\n\t"
#define I_FMA(d, s) "v_fma_f32 " d ", " d ", %8, %9\n\t"
#define I_CVT(d, s) "v_cvt_i32_f32 " d ", " d "\n\t"
#define I_CVTPK(d, s) "v_cvt_pk_i16_i32 " d ", " d ", %8\n\t"
#define I_LSHLADD(d, s) "v_lshl_add_u32 " d ", " d ", 2, %8\n\t"
#define I_SDWA(d, s) "v_add_u32_sdwa " d ", " d ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
#define I_DPPMOV(d, s) "v_mov_b32_dpp " d ", " s " row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
#define I_DPPCND(d, s) "v_cndmask_b32_dpp " d ", " s ", %8, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define I_DPPCNDR(d, s) "v_cndmask_b32_dpp " d ", " s ", %8, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
#define I_SWAP32(d, s) "v_permlane32_swap_b32 " d ", " s "\n\t"
#define I_SWAP16(d, s) "v_permlane16_swap_b32 " d ", " s "\n\t"
#define I_CMP(d, s) "v_cmp_lt_f32_e64 s[20:21], 0, " d "\n\t"
#define I_CNDS(d, s) "v_cndmask_b32_e64 " d ", " d ", %8, s[22:23]\n\t"
#define I_CNDNEG(d, s) "v_cndmask_b32_e64 " d ", " d ", -" d ", s[22:23]\n\t"
#define I_PERM(d, s) "v_perm_b32 " d ", " d ", %8, %9\n\t"
#define I_RCP(d, s) "v_rcp_f32 " d ", " d "\n\t"
#define I_BFI(d, s) "v_bfi_b32 " d ", %8, " d ", %9\n\t"
#define I_PKMUL(d, s) "v_pk_mul_f32 " d ", " d ", %8\n\t"
#define I_PKMULSEL(d, s) "v_pk_mul_f32 " d ", " d ", %8 op_sel:[1,1] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
#define I_PKADD(d, s) "v_pk_add_f32 " d ", " d ", %8 neg_lo:[0,1] neg_hi:[0,1]\n\t"
#define I_PKFMA(d, s) "v_pk_fma_f32 " d ", " d ", %8, %8\n\t"
#define I_MOV64(d, s) "v_mov_b64 " d ", %8\n\t"
#define I_PKMOV(d, s) "v_pk_mov_b32 " d ", " d ", %8 op_sel:[1,0]\n\t"

#define I_CNDVCC(d, s) "v_cndmask_b32_e32 " d ", " d ", %8, vcc\n\t"
#define I_CMPVCC(d, s) "v_cmp_lt_f32_e32 vcc, 0, " d "\n\t"
#define I_DPPQ(d, s) "v_mov_b32_dpp " d ", " s " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define I_DPPQBC(d, s) "v_mov_b32_dpp " d ", " s " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define I_SWZ(d, s) "ds_swizzle_b32 " d ", " s " offset:swizzle(QUAD_PERM,1,0,3,2)\n\t"
#define I_BPERM(d, s) "ds_bpermute_b32 " d ", %8, " s "\n\t"
#define I_MUL(d, s) "v_mul_f32_e32 " d ", " d ", %8\n\t"
#define I_CVTF(d, s) "v_cvt_f32_i32_e32 " d ", " d "\n\t"
#define I_TRUNC(d, s) "v_trunc_f32_e32 " d ", " d "\n\t"
#define I_LSHL(d, s) "v_lshlrev_b32_e32 " d ", 2, " d "\n\t"
#define I_ADDU(d, s) "v_add_u32_e32 " d ", " d ", %8\n\t"
#define I_AND(d, s) "v_and_b32_e32 " d ", " d ", %8\n\t"
#define I_ADD3(d, s) "v_add3_u32 " d ", " d ", %8, %9\n\t"
#define I_MAD24(d, s) "v_mad_u32_u24 " d ", " d ", %8, %9\n\t"
#define I_CNDDPP_S(d, s) "v_cndmask_b32_dpp " d ", " s ", %8, vcc row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define I_ADDDPP(d, s) "v_add_f32_dpp " d ", " s ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define KERNEL(NAME, I, T, C)                                                                                \
	__global__ void NAME(unsigned long long *out, float seed, int reps)                                      \
	{                                                                                                        \
		T a = T(seed), b = a, c = a, d = a, e = a, f = a, g = a, h = a, m = T(1.0f), z = T(0.0f);            \
		__syncthreads();                                                                                     \
		unsigned long long t0, t1;                                                                           \
		asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));                                     \
		for (int r = 0; r < reps; r++)                                                                       \
			asm volatile("s_mov_b64 s[22:23], 0x5555\n\ts_mov_b64 vcc, 0x3333\n\t" R64(OPS(I)) "s_waitcnt lgkmcnt(0)\n\t"                 \
					: "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)                   \
					: "v"(m), "v"(z), "v"(z) : "s20", "s21", "s22", "s23", "vcc");                             \
		asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));                                     \
		if (C((a)) + C((b)) + C((c)) + C((d)) + C((e)) + C((f)) + C((g)) + C((h)) == 12345.f)                                \
			out[1000] = 1;                                                                                   \
		if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) {                                                    \
			out[2 * (threadIdx.x >> 6)] = t0;                                                                \
			out[2 * (threadIdx.x >> 6) + 1] = t1;                                                            \
		}                                                                                                    \
	}
#define CF(x) (x)
#define C2(v_) ((v_)[0] + (v_)[1])
KERNEL(k_mov, I_MOV, float, CF)
KERNEL(k_add, I_ADD, float, CF)
KERNEL(k_fma, I_FMA, float, CF)
KERNEL(k_cvt, I_CVT, float, CF)
KERNEL(k_cvtpk, I_CVTPK, float, CF)
KERNEL(k_lshladd, I_LSHLADD, float, CF)
KERNEL(k_sdwa, I_SDWA, float, CF)
KERNEL(k_dppmov, I_DPPMOV, float, CF)
KERNEL(k_dppcnd, I_DPPCND, float, CF)
KERNEL(k_dppcndr, I_DPPCNDR, float, CF)
KERNEL(k_swap32, I_SWAP32, float, CF)
KERNEL(k_swap16, I_SWAP16, float, CF)
KERNEL(k_cmp, I_CMP, float, CF)
KERNEL(k_cnds, I_CNDS, float, CF)
KERNEL(k_cndneg, I_CNDNEG, float, CF)
KERNEL(k_perm, I_PERM, float, CF)
KERNEL(k_rcp, I_RCP, float, CF)
KERNEL(k_bfi, I_BFI, float, CF)
KERNEL(k_cndvcc, I_CNDVCC, float, CF)
KERNEL(k_cmpvcc, I_CMPVCC, float, CF)
KERNEL(k_dppq, I_DPPQ, float, CF)
KERNEL(k_dppqbc, I_DPPQBC, float, CF)
KERNEL(k_swz, I_SWZ, float, CF)
KERNEL(k_bperm, I_BPERM, float, CF)
KERNEL(k_mul, I_MUL, float, CF)
KERNEL(k_cvtf, I_CVTF, float, CF)
KERNEL(k_trunc, I_TRUNC, float, CF)
KERNEL(k_lshl, I_LSHL, float, CF)
KERNEL(k_addu, I_ADDU, float, CF)
KERNEL(k_and, I_AND, float, CF)
KERNEL(k_add3, I_ADD3, float, CF)
KERNEL(k_mad24, I_MAD24, float, CF)
KERNEL(k_cnddpps, I_CNDDPP_S, float, CF)
KERNEL(k_adddpp, I_ADDDPP, float, CF)
KERNEL(k_pkmul, I_PKMUL, float2_t, C2)
KERNEL(k_pkmulsel, I_PKMULSEL, float2_t, C2)
KERNEL(k_pkadd, I_PKADD, float2_t, C2)
KERNEL(k_pkfma, I_PKFMA, float2_t, C2)
KERNEL(k_mov64, I_MOV64, float2_t, C2)
KERNEL(k_pkmov, I_PKMOV, float2_t, C2)

typedef void (*kern_t)(unsigned long long *, float, int);
int main()
{
	unsigned long long *d, h[64];
	(void)hipMalloc(&d, 16384);
	struct { const char *name; kern_t k; } ks[] = {
		{"v_mov_b32", k_mov}, {"v_add_f32", k_add}, {"v_fma_f32", k_fma}, {"v_cvt_i32_f32", k_cvt}, {"v_cvt_pk_i16_i32", k_cvtpk},
		{"v_lshl_add_u32", k_lshladd}, {"v_add_u32_sdwa", k_sdwa}, {"v_mov_b32_dpp row_ror (tied)", k_dppmov},
		{"v_cndmask_b32_dpp quad_perm", k_dppcnd}, {"v_cndmask_b32_dpp row_ror", k_dppcndr}, {"v_permlane32_swap", k_swap32},
		{"v_permlane16_swap", k_swap16}, {"v_cmp_lt_f32 -> sgpr", k_cmp}, {"v_cndmask_b32 (sgpr mask)", k_cnds},
		{"v_cndmask_b32 (neg src)", k_cndneg}, {"v_perm_b32", k_perm}, {"v_rcp_f32", k_rcp}, {"v_bfi_b32", k_bfi},
		{"v_pk_mul_f32", k_pkmul}, {"v_pk_mul_f32 op_sel/neg", k_pkmulsel}, {"v_pk_add_f32 neg", k_pkadd}, {"v_pk_fma_f32", k_pkfma},
		{"v_mov_b64", k_mov64}, {"v_pk_mov_b32", k_pkmov},
		{"v_cndmask_b32_e32 (vcc)", k_cndvcc}, {"v_cmp_lt_f32_e32 -> vcc", k_cmpvcc}, {"v_mov_b32_dpp quad_perm (all lanes)", k_dppq},
		{"v_mov_b32_dpp quad_perm bound_ctrl", k_dppqbc}, {"ds_swizzle_b32", k_swz}, {"ds_bpermute_b32", k_bperm}, {"v_mul_f32_e32", k_mul},
		{"v_cvt_f32_i32", k_cvtf}, {"v_trunc_f32", k_trunc}, {"v_lshlrev_b32_e32", k_lshl}, {"v_add_u32_e32", k_addu}, {"v_and_b32_e32", k_and},
		{"v_add3_u32", k_add3}, {"v_mad_u32_u24", k_mad24}, {"v_cndmask_b32_dpp row_shr bc", k_cnddpps}, {"v_add_f32_dpp quad_perm", k_adddpp}};
	for (int waves : {16})
		for (auto &e : ks) {
			const int reps = 4;
			for (int rep = 0; rep < 3; rep++) {
				hipLaunchKernelGGL(e.k, dim3(1), dim3(64 * waves), 0, 0, d, 1.0f, reps);
				(void)hipDeviceSynchronize();
			}
			(void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
			unsigned long long lo = ~0ull, hi = 0;
			for (int w = 0; w < waves; w++) { lo = std::min(lo, h[2 * w]); hi = std::max(hi, h[2 * w + 1]); }
			const double n = 512.0 * reps;
			printf("%2d waves/CU %-30s %.2f cycles per instruction per SIMD (one wave alone %.2f)\n", waves, e.name,
					(double)(hi - lo) / (n * waves / 4), (double)(h[1] - h[0]) / n);
		}
	return 0;
}
