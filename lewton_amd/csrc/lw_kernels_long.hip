// Specialised long-block synthesis kernel for gfx950 (MI355X): n = 2048, window flags (1,1).
//
// One wave64 handles one "unit" (a coupled channel pair, or a single channel) of one packet from the
// entropy-decoded record to PCM:
//     residue load (coalesced float4) -> inverse coupling -> floor-1 curve -> floor x residue
//     -> IMDCT entirely in registers with three wave-private LDS transposes -> window/overlap-add
//     -> i16/f32 store (coalesced).
// A workgroup is 16 waves (one per CU, 4 waves per SIMD, <= 128 VGPRs) and works through a chunk of
// `rounds` x `per_round` consecutive items of the stream-sorted work list, one round at a time.  There is no
// s_barrier after the table image is staged: a packet's un-windowed right half reaches its successor through a
// 4 KB hand-over buffer in LDS guarded by a pair of LDS counters (published / consumed), so every wave runs
// ahead as far as its own data allows.  The second half of the waves issues its HBM loads only after staging
// the tables, so the first half's data is queued -- and arrives -- first: HBM traffic of one half overlaps the
// arithmetic of the other inside a single launch.
// LDS: 24 KB re-ordered twiddle/window tables + per wave 4 KB transpose scratch + 4 KB hand-over buffer.
// At chunk starts the previous right half comes from the stream's state slot, from a halo buffer filled by a
// RIGHT_ONLY pre-pass of this same kernel, or from the time-domain block of a generic-kernel predecessor.
//
// Register layouts of the 512 complex pairs p (u[2p], u[2p+1]) of imdct.rs's butterfly array, 8 per lane:
//   B: lane = p[5:0], reg = p[8:6]   step 2 and stages l = 0,1   (pair bits 8,7,6 are lane-local)
//   C: lane = (p[8:6], p[2:0]), reg = p[5:3]   stages l = 2,3,4
//   D: lane = p[8:3], reg = p[2:0]   fused last three stages (imdct.rs:234-288), lane-local
//   E: lane handles m' = 2*lane + c: bit-reverse gather (imdct.rs:490-528), step 7, step 8, overlap-add
// The numpy model tests/fast_model.py is the executable specification of these layouts; it is checked
// bit-for-bit against the oracle on the CPU.
//
// Arithmetic contract: identical to lw_kernels.hip -- same f32 operations on the same operands as the
// reference, compiled with -ffp-contract=off.  The floor curve uses trunc((t*dy +- 0.5) * (1/adx)), proven
// equal to the reference's integer render_line for every reachable segment (tests/test_fast_model.py).
#include <algorithm>
#include <cstddef>
#include <mutex>

#include "lw_fast.hpp"
#include "lw_kernels.hpp"

#define LW_WG (64 * LW_FAST_WAVES)
// wave priorities (s_setprio 0..3) of the three kinds of phases; measured on MI355X (tools/exp.sh)
#define LW_PRIO_FLOOR 2
#define LW_PRIO_IMDCT 0
#define LW_PRIO_FINISH 3
#define LW_FLOOR_FIRST_FROM 7 // first wave of a workgroup that builds its floor curve before it requests its residues (>= 4)
#define LW_IMDCT_PRIO_LATE 12
#define LW_PRIO_PACE 0 // while a wave waits for its turn in the load queue and requests its residues
#define LW_PRIO_READY 3 // inverse coupling + floor multiply of a wave whose floor curve was built ahead
#define LW_SCR_BYTES 4096u // per wave: transposes of one channel at a time / 2 x 1 KB floor segment tables
#define LW_PUB_BYTES 4096u // per wave: published right half [2 channels][2][64] float4
#define LW_LDS_BYTES (LWI_TOTAL + LW_FAST_WAVES * (LW_SCR_BYTES + LW_PUB_BYTES) + 3 * LW_FAST_WAVES * 4)

#include "lw_stamps.inc" // experiment hooks: empty macros unless built with -DLW_STAMPS

struct LwFastArgs {
	// hot scalars first: everything the first HBM loads depend on sits in the first kernel-argument cache lines
	const float *residue;      // batch arrays (lw_records.h)
	const uint16_t *floors;
	uint32_t n_items;
	uint32_t n_units;
	uint32_t per_round;        // packets per workgroup and round (<= LW_FAST_WAVES / n_units)
	uint32_t rounds;           // rounds per workgroup
	uint32_t dense;            // item k of the list is packet k and every packet block has the same size
	uint32_t late_from;        // pacing group size: wave w issues its first HBM loads when wave w - late_from has its data
	uint32_t ch, fstride;      // channels, u16 entries per channel in a floor block
	LwFastUnit waves[LW_FAST_WAVES]; // per wave: its unit and packet slot (one 8-byte scalar load at kernarg + 48 + 8 * wave)
	const uint8_t *image;      // LDS image in HBM (LWI_TOTAL bytes)
	const LwFastItem *items;   // work list in stream-sorted order
	float *state;              // state pool [slots][2][ch][n1/2]
	float *td;                 // time-domain blocks of the generic kernels
	float *halo;               // [slots][ch][512]
	float *edge;               // EDGE kernels: [packet][side][ch][64] raw edges of long blocks with short slopes (lw_fast.hpp)
	void *out;
	uint32_t state_stride, state_chan_stride;
	uint32_t edge_n;           // k_long10<EDGE>: values per raw edge = blocksize_0 / 4 (k_long: always LW_EDGE_VALUES)
	const float *tabA;         // k_long12: the block size's A table (header_cached.rs:64-99) in HBM, read as pairs by step 1
	const uint16_t *sid12;     // k_long12: static interval indices of the staged floors (LwL12Layout)
	uint32_t pre[LW_FAST_WAVES]; // k_long<..., PRE>: per wave, the coupling steps it evaluates itself (LwFastPlan::pre, lw_fast.hpp)
};
static_assert(offsetof(LwFastArgs, waves) == 48, "kernel reads waves[] through the kernarg segment pointer");

typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void lds_fence()
{
	asm volatile("" ::: "memory"); // wave-private LDS traffic is in order in hardware; stop compiler motion only
}

// LDS access by absolute byte address (the kernel has no static LDS, the dynamic segment starts at 0): saves the add of
// the relocatable segment base that hipcc otherwise keeps in every computed address
typedef __attribute__((address_space(3))) const float *lds_cfloat_ptr;
__device__ __forceinline__ float lds_abs_f32(uint32_t byte_addr)
{
	return *reinterpret_cast<lds_cfloat_ptr>((uintptr_t)byte_addr);
}

__device__ __forceinline__ float2_t lds2(const char *base, uint32_t byte_off)
{
	return *reinterpret_cast<const float2_t *>(base + byte_off);
}

__device__ __forceinline__ float4_t lds4(const char *base, uint32_t byte_off)
{
	return *reinterpret_cast<const float4_t *>(base + byte_off);
}

// ---------------------------------------------------------------------------------------------
// Packed f32 arithmetic.  Every multiply/add of the transform is issued as v_pk_mul_f32 / v_pk_add_f32 on
// register pairs, with op_sel / op_sel_hi choosing the half of each source that feeds the low / high result
// and neg_lo / neg_hi negating sources (exact).  Written as inline asm so that the operation tree is exactly
// the reference's (no contraction, no re-association) and no register shuffling is left to the vectoriser.
// tests/fast_model.py emulates these very modifier sets and is checked bit-for-bit against the oracle.
// ---------------------------------------------------------------------------------------------
#define LW_PK(name, op, mods)                                                              \
	__device__ __forceinline__ float2_t name(float2_t a, float2_t b)                       \
	{                                                                                      \
		float2_t d;                                                                        \
		asm(op " %0, %1, %2 " mods : "=v"(d) : "v"(a), "v"(b));                            \
		return d;                                                                          \
	}
LW_PK(pk_add, "v_pk_add_f32", "")                                                              // (a.lo+b.lo, a.hi+b.hi)
LW_PK(pk_sub, "v_pk_add_f32", "neg_lo:[0,1] neg_hi:[0,1]")                                     // (a.lo-b.lo, a.hi-b.hi)
LW_PK(pk_add_A2, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")                  // (a.lo-b.hi, a.hi+b.lo)
LW_PK(pk_add_A3, "v_pk_add_f32", "op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[1,0]")     // (a.hi-b.hi, b.lo-a.lo)
LW_PK(pk_add_A4, "v_pk_add_f32", "op_sel:[0,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]")     // (b.lo-a.lo, a.hi-b.hi)
LW_PK(pk_add_A5, "v_pk_add_f32", "op_sel:[1,0] op_sel_hi:[1,0] neg_lo:[0,1]")                  // (a.hi-b.lo, a.hi+b.lo)
LW_PK(pk_add_A6, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")                  // (a.lo+b.hi, a.hi-b.lo)
LW_PK(pk_add_A7, "v_pk_add_f32", "neg_hi:[0,1]")                                               // (a.lo+b.lo, a.hi-b.hi)
LW_PK(pk_add_A8, "v_pk_add_f32", "neg_lo:[0,1]")                                               // (a.lo-b.lo, a.hi+b.hi)
LW_PK(pk_add_A9, "v_pk_add_f32", "neg_lo:[0,1] neg_hi:[1,0]")                                  // (a.lo-b.lo, b.hi-a.hi)
LW_PK(pk_mul, "v_pk_mul_f32", "")                                                              // (a.lo*b.lo, a.hi*b.hi)
LW_PK(pk_mul_M1, "v_pk_mul_f32", "op_sel:[0,0] op_sel_hi:[1,0]")                               // (a.lo*b.lo, a.hi*b.lo)
LW_PK(pk_mul_M2, "v_pk_mul_f32", "op_sel:[1,1] op_sel_hi:[0,1] neg_hi:[0,1]")                  // (a.hi*b.hi, -a.lo*b.hi)
LW_PK(pk_mul_M3, "v_pk_mul_f32", "op_sel:[0,1] op_sel_hi:[0,0]")                               // (a.lo*b.hi, a.lo*b.lo)
LW_PK(pk_mul_M4, "v_pk_mul_f32", "op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]")                  // (a.lo*b.lo, -a.lo*b.hi)
LW_PK(pk_mul_M5, "v_pk_mul_f32", "op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]")     // (-a.hi*b.hi, -a.hi*b.lo)
LW_PK(pk_mul_M6, "v_pk_mul_f32", "op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0]")                  // (-a.hi*b.lo, a.hi*b.hi)
LW_PK(pk_mul_M7, "v_pk_mul_f32", "op_sel:[0,1] op_sel_hi:[1,1]")                               // (a.lo*b.hi, a.hi*b.hi)
LW_PK(pk_mul_M8, "v_pk_mul_f32", "op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[0,1]")                  // (-a.hi*b.lo, a.lo*b.lo)
LW_PK(pk_mul_M9, "v_pk_mul_f32", "op_sel:[1,1] op_sel_hi:[1,0] neg_hi:[1,0]")                  // (a.hi*b.hi, -a.hi*b.lo)
LW_PK(pk_mul_M10, "v_pk_mul_f32", "op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]")    // (-a.lo*b.lo, -a.lo*b.hi)
LW_PK(pk_mul_M12lo, "v_pk_mul_f32", "op_sel:[0,1] op_sel_hi:[0,0]")                            // (a.lo*b.hi, a.lo*b.lo)
LW_PK(pk_mul_M12hi, "v_pk_mul_f32", "op_sel:[1,1] op_sel_hi:[1,0]")                            // (a.hi*b.hi, a.hi*b.lo)

// imdct.rs:36-41 on pairs (e0, e1) = (u[hi-1], u[hi]); t = (t0, t1)
__device__ __forceinline__ void bfly2(float2_t &H, float2_t &L, float2_t t)
{
	const float2_t S = pk_add(H, L);
	const float2_t K = pk_sub(H, L);                  // (k01, k00)
	L = pk_add(pk_mul_M1(K, t), pk_mul_M2(K, t));     // (k01 t0 + k00 t1, k00 t0 - k01 t1)
	H = S;
}

// imdct.rs:548-560 (one half of a step-7 iteration): P = (D1, D0), Q = (E3, E2), C2 = (C0, C1)
__device__ __forceinline__ void step7_block(float2_t P, float2_t Q, float2_t C2, float2_t &Dn, float2_t &En)
{
	const float2_t Aa = pk_add_A7(P, Q);              // (a11, a02)
	const float2_t Bb = pk_add_A8(P, Q);              // (b3, b2)
	const float2_t Bv = pk_add(pk_mul_M7(Aa, C2), pk_mul_M8(Aa, C2)); // (b1, b0)
	Dn = pk_add(Bb, Bv);                              // (b3 + b1, b2 + b0)
	En = pk_add_A9(Bv, Bb);                           // (b1 - b3, b2 - b0)
}

// imdct.rs:619-620: Wv = (w1, w0), Bq = (Bc, Bs) -> (pa, pb)
__device__ __forceinline__ float2_t step8(float2_t Wv, float2_t Bq)
{
	return pk_add(pk_mul_M9(Wv, Bq), pk_mul_M10(Wv, Bq));
}

// ---------------------------------------------------------------------------------------------
// Grouped arithmetic blocks: the same packed operations as the single-instruction helpers above, but several
// independent chains per asm statement, interleaved so that dependent instructions are >= 4 issue slots apart.
// One statement = one scheduling unit for hipcc: no per-instruction hazard pads, no register shuffling between
// the operations of a butterfly.  Modifier sets (see LW_PK above): SUB, M1, M2, ...
// ---------------------------------------------------------------------------------------------
#define LW_M_SUB " neg_lo:[0,1] neg_hi:[0,1]"
#define LW_M_M1 " op_sel:[0,0] op_sel_hi:[1,0]"
#define LW_M_M2 " op_sel:[1,1] op_sel_hi:[0,1] neg_hi:[0,1]"
#define LW_M_M3 " op_sel:[0,1] op_sel_hi:[0,0]"
#define LW_M_M4 " op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]"
#define LW_M_M5 " op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]"
#define LW_M_M6 " op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0]"
#define LW_M_M7 " op_sel:[0,1] op_sel_hi:[1,1]"
#define LW_M_M8 " op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[0,1]"
#define LW_M_M9 " op_sel:[1,1] op_sel_hi:[1,0] neg_hi:[1,0]"
#define LW_M_M10 " op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]"
#define LW_M_M12LO " op_sel:[0,1] op_sel_hi:[0,0]"
#define LW_M_M12HI " op_sel:[1,1] op_sel_hi:[1,0]"
#define LW_M_A2 " op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"
#define LW_M_A3 " op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[1,0]"
#define LW_M_A4 " op_sel:[0,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]"
#define LW_M_A5 " op_sel:[1,0] op_sel_hi:[1,0] neg_lo:[0,1]"
#define LW_M_A6 " op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]"
#define LW_M_A7 " neg_hi:[0,1]"
#define LW_M_A8 " neg_lo:[0,1]"
#define LW_M_A9 " neg_lo:[0,1] neg_hi:[1,0]"
#define LW_ADD(d, a, b, m) "v_pk_add_f32 " d ", " a ", " b m "\n\t"
#define LW_MUL(d, a, b, m) "v_pk_mul_f32 " d ", " a ", " b m "\n\t"

// four butterflies (imdct.rs:36-41 / :94-99 / :161-166): (H_i, L_i, t_i) in place
__device__ __forceinline__ void bfly2x4(float2_t &H0, float2_t &L0, float2_t t0, float2_t &H1, float2_t &L1, float2_t t1,
		float2_t &H2, float2_t &L2, float2_t t2, float2_t &H3, float2_t &L3, float2_t t3)
{
	float2_t K0, K1, K2, K3;
	asm(LW_ADD("%8", "%0", "%1", LW_M_SUB) LW_ADD("%9", "%2", "%3", LW_M_SUB) LW_ADD("%10", "%4", "%5", LW_M_SUB) LW_ADD("%11", "%6", "%7", LW_M_SUB)
	    LW_ADD("%0", "%0", "%1", "") LW_ADD("%2", "%2", "%3", "") LW_ADD("%4", "%4", "%5", "") LW_ADD("%6", "%6", "%7", "")
	    LW_MUL("%1", "%8", "%12", LW_M_M1) LW_MUL("%3", "%9", "%13", LW_M_M1) LW_MUL("%5", "%10", "%14", LW_M_M1) LW_MUL("%7", "%11", "%15", LW_M_M1)
	    LW_MUL("%8", "%8", "%12", LW_M_M2) LW_MUL("%9", "%9", "%13", LW_M_M2) LW_MUL("%10", "%10", "%14", LW_M_M2) LW_MUL("%11", "%11", "%15", LW_M_M2)
	    LW_ADD("%1", "%1", "%8", "") LW_ADD("%3", "%3", "%9", "") LW_ADD("%5", "%5", "%10", "") LW_ADD("%7", "%7", "%11", "")
	    : "+v"(H0), "+v"(L0), "+v"(H1), "+v"(L1), "+v"(H2), "+v"(L2), "+v"(H3), "+v"(L3), "=&v"(K0), "=&v"(K1), "=&v"(K2), "=&v"(K3)
	    : "v"(t0), "v"(t1), "v"(t2), "v"(t3));
}

// step 1 (imdct.rs:337-371) of two float4 groups: X = (X0, X1, X2, X3) -> U (pair 511 - m, still on the mirror lane), P (pair m)
__device__ __forceinline__ void step1x2(float4_t Xa, float2_t aua, float2_t ala, float4_t Xb, float2_t aub, float2_t alb,
		float2_t &Ua, float2_t &Pa, float2_t &Ub, float2_t &Pb)
{
	float2_t Ta, Tb;
	const float2_t Xa0 = float2_t{Xa.x, Xa.y}, Xa1 = float2_t{Xa.z, Xa.w}, Xb0 = float2_t{Xb.x, Xb.y}, Xb1 = float2_t{Xb.z, Xb.w};
	asm(LW_MUL("%0", "%6", "%10", LW_M_M3) LW_MUL("%3", "%8", "%12", LW_M_M3)
	    LW_MUL("%2", "%7", "%10", LW_M_M4) LW_MUL("%5", "%9", "%12", LW_M_M4)
	    LW_MUL("%1", "%7", "%11", LW_M_M5) LW_MUL("%4", "%9", "%13", LW_M_M5)
	    LW_ADD("%0", "%0", "%2", "") LW_ADD("%3", "%3", "%5", "")
	    LW_MUL("%2", "%6", "%11", LW_M_M6) LW_MUL("%5", "%8", "%13", LW_M_M6)
	    LW_ADD("%1", "%1", "%2", "") LW_ADD("%4", "%4", "%5", "")
	    : "=&v"(Ua), "=&v"(Pa), "=&v"(Ta), "=&v"(Ub), "=&v"(Pb), "=&v"(Tb)
	    : "v"(Xa0), "v"(Xa1), "v"(Xb0), "v"(Xb1), "v"(aua), "v"(ala), "v"(aub), "v"(alb));
}

// fused last three stages (imdct.rs:234-288) of one 16-float group, 28 packed operations, no register moves:
// z4..z7 are updated in place, the new z0..z3 come out in fresh registers
__device__ __forceinline__ void stage_d_block(float2_t a2, float2_t (&z)[8])
{
	float2_t n0, n1, n2, n3, t0, t1, t2, t3;
	// %0-%3 = z4..z7 (in/out)   %4-%7 = n0..n3   %8-%11 = t0..t3   %12-%15 = z0..z3 (in)   %16 = a2
	asm(// imdct.rs:240-275: pairs (7,3) (6,2) (5,1) (4,0); t0..t3 = the new z3, z2, z1, z0
	    LW_ADD("%8", "%3", "%15", LW_M_SUB)     // t0 = z7 - z3
	    LW_ADD("%9", "%2", "%14", LW_M_SUB)     // t1 = z6 - z2 = (k11, k00)
	    LW_ADD("%10", "%13", "%1", LW_M_A3)     // t2 = A3(z1, z5)
	    LW_ADD("%11", "%12", "%0", LW_M_A4)     // t3 = A4(z0, z4) = (k11, k00)
	    LW_ADD("%3", "%3", "%15", "")           // z7 += z3
	    LW_ADD("%2", "%2", "%14", "")           // z6 += z2
	    LW_ADD("%1", "%1", "%13", "")           // z5 += z1
	    LW_ADD("%0", "%0", "%12", "")           // z4 += z0
	    LW_ADD("%9", "%9", "%9", LW_M_A2)
	    LW_ADD("%11", "%11", "%11", LW_M_A5)
	    // iter_54 (imdct.rs:202-232) on z[4..8)
	    LW_ADD("%7", "%3", "%1", "")            // A  = z7 + z5
	    LW_ADD("%6", "%2", "%0", "")            // Cc = z6 + z4
	    LW_MUL("%9", "%9", "%16", "")           // t1 = new z2
	    LW_MUL("%11", "%11", "%16", "")         // t3 = new z0
	    LW_ADD("%5", "%3", "%1", LW_M_SUB)      // Bm = z7 - z5
	    LW_ADD("%4", "%2", "%0", LW_M_SUB)      // Dm = z6 - z4
	    LW_ADD("%3", "%7", "%6", "")            // z7 = A + Cc
	    LW_ADD("%2", "%7", "%6", LW_M_SUB)      // z6 = A - Cc
	    LW_ADD("%1", "%5", "%4", LW_M_A2)       // z5 = A2(Bm, Dm)
	    LW_ADD("%0", "%5", "%4", LW_M_A6)       // z4 = A6(Bm, Dm)
	    // iter_54 on the new z[0..4) = (t3, t2, t1, t0)
	    LW_ADD("%5", "%8", "%10", LW_M_SUB)     // Bm = z3 - z1
	    LW_ADD("%4", "%9", "%11", LW_M_SUB)     // Dm = z2 - z0
	    LW_ADD("%8", "%8", "%10", "")           // A  = z3 + z1
	    LW_ADD("%9", "%9", "%11", "")           // Cc = z2 + z0
	    LW_ADD("%10", "%5", "%4", LW_M_A2)      // new z1 = A2(Bm, Dm)
	    LW_ADD("%4", "%5", "%4", LW_M_A6)       // new z0 = A6(Bm, Dm)
	    LW_ADD("%7", "%8", "%9", "")            // new z3 = A + Cc
	    LW_ADD("%6", "%8", "%9", LW_M_SUB)      // new z2 = A - Cc
	    : "+v"(z[4]), "+v"(z[5]), "+v"(z[6]), "+v"(z[7]), "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3),
	      "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
	    : "v"(z[0]), "v"(z[1]), "v"(z[2]), "v"(z[3]), "v"(a2));
	z[3] = n3;
	z[2] = n2;
	z[1] = t2;
	z[0] = n0;
}

// step 7 (imdct.rs:533-580) + step 8 (:589-658) of one m': (p511, pq, p255, pq256) -> R[0..4) = (pa, pb) at q = 511-2m', 510-2m', 1+2m', 2m'
__device__ __forceinline__ void step78_block(float2_t p511, float2_t pq, float2_t p255, float2_t pq256, float4_t Cq, float4_t Bl, float4_t Bh,
		float2_t (&R)[4])
{
	float2_t A1, B1, T1, A2, B2, T2;
	const float2_t C0 = float2_t{Cq.x, Cq.y}, C1 = float2_t{Cq.z, Cq.w}, Bl0 = float2_t{Bl.x, Bl.y}, Bl1 = float2_t{Bl.z, Bl.w};
	const float2_t Bh0 = float2_t{Bh.x, Bh.y}, Bh1 = float2_t{Bh.z, Bh.w};
	asm(LW_ADD("%4", "%10", "%11", LW_M_A7) LW_ADD("%7", "%12", "%13", LW_M_A7)       // Aa = A7(P, Q)
	    LW_ADD("%5", "%10", "%11", LW_M_A8) LW_ADD("%8", "%12", "%13", LW_M_A8)       // Bb = A8(P, Q)
	    LW_MUL("%6", "%4", "%14", LW_M_M7) LW_MUL("%9", "%7", "%15", LW_M_M7)         // T = M7(Aa, C)
	    LW_MUL("%4", "%4", "%14", LW_M_M8) LW_MUL("%7", "%7", "%15", LW_M_M8)         // Aa = M8(Aa, C)
	    LW_ADD("%6", "%6", "%4", "") LW_ADD("%9", "%9", "%7", "")                     // Bv = T + Aa
	    LW_ADD("%4", "%5", "%6", "") LW_ADD("%7", "%8", "%9", "")                     // Dn = Bb + Bv     (A1 = Dn1, A2 = Dn2)
	    LW_ADD("%5", "%6", "%5", LW_M_A9) LW_ADD("%8", "%9", "%8", LW_M_A9)           // En = A9(Bv, Bb)  (B1 = En1, B2 = En2)
	    LW_MUL("%6", "%4", "%16", LW_M_M9) LW_MUL("%9", "%7", "%17", LW_M_M9)         // step 8 of Dn1 / Dn2
	    LW_MUL("%0", "%4", "%16", LW_M_M10) LW_MUL("%1", "%7", "%17", LW_M_M10)
	    LW_ADD("%0", "%6", "%0", "") LW_ADD("%1", "%9", "%1", "")
	    LW_MUL("%6", "%8", "%18", LW_M_M9) LW_MUL("%9", "%5", "%19", LW_M_M9)         // step 8 of En2 / En1
	    LW_MUL("%2", "%8", "%18", LW_M_M10) LW_MUL("%3", "%5", "%19", LW_M_M10)
	    LW_ADD("%2", "%6", "%2", "") LW_ADD("%3", "%9", "%3", "")
	    : "=&v"(R[0]), "=&v"(R[1]), "=&v"(R[2]), "=&v"(R[3]), "=&v"(A1), "=&v"(B1), "=&v"(T1), "=&v"(A2), "=&v"(B2), "=&v"(T2)
	    : "v"(p511), "v"(pq), "v"(p255), "v"(pq256), "v"(C0), "v"(C1), "v"(Bl0), "v"(Bl1), "v"(Bh0), "v"(Bh1));
}

// window + overlap-add (audio.rs:1116-1118) of the four pairs of one c2, optionally scaled by 32768 (samples.rs:92-103)
template <bool SCALE>
__device__ __forceinline__ void ola_block(const float2_t (&Rc)[4], float2_t pp0, float2_t pp1, float4_t w0, float4_t w1, float2_t (&O)[4])
{
	float2_t T0, T1;
	const float2_t S0 = float2_t{w0.x, w0.y}, S1 = float2_t{w0.z, w0.w}, S2 = float2_t{w1.x, w1.y}, S3 = float2_t{w1.z, w1.w};
	const float2_t k = float2_t{32768.0f, 32768.0f};
	if (SCALE)
		asm(LW_MUL("%0", "%6", "%12", LW_M_M4) LW_MUL("%1", "%7", "%13", LW_M_M4) LW_MUL("%2", "%8", "%14", LW_M_M4) LW_MUL("%3", "%9", "%15", LW_M_M4)
		    LW_MUL("%4", "%10", "%12", LW_M_M12LO) LW_MUL("%5", "%10", "%13", LW_M_M12HI)
		    LW_ADD("%0", "%0", "%4", "") LW_ADD("%1", "%1", "%5", "")
		    LW_MUL("%4", "%11", "%14", LW_M_M12LO) LW_MUL("%5", "%11", "%15", LW_M_M12HI)
		    LW_ADD("%2", "%2", "%4", "") LW_ADD("%3", "%3", "%5", "")
		    LW_MUL("%0", "%0", "%16", "") LW_MUL("%1", "%1", "%16", "") LW_MUL("%2", "%2", "%16", "") LW_MUL("%3", "%3", "%16", "")
		    : "=&v"(O[0]), "=&v"(O[1]), "=&v"(O[2]), "=&v"(O[3]), "=&v"(T0), "=&v"(T1)
		    : "v"(Rc[0]), "v"(Rc[1]), "v"(Rc[2]), "v"(Rc[3]), "v"(pp0), "v"(pp1), "v"(S0), "v"(S1), "v"(S2), "v"(S3), "v"(k));
	else
		asm(LW_MUL("%0", "%6", "%12", LW_M_M4) LW_MUL("%1", "%7", "%13", LW_M_M4) LW_MUL("%2", "%8", "%14", LW_M_M4) LW_MUL("%3", "%9", "%15", LW_M_M4)
		    LW_MUL("%4", "%10", "%12", LW_M_M12LO) LW_MUL("%5", "%10", "%13", LW_M_M12HI)
		    LW_ADD("%0", "%0", "%4", "") LW_ADD("%1", "%1", "%5", "")
		    LW_MUL("%4", "%11", "%14", LW_M_M12LO) LW_MUL("%5", "%11", "%15", LW_M_M12HI)
		    LW_ADD("%2", "%2", "%4", "") LW_ADD("%3", "%3", "%5", "")
		    : "=&v"(O[0]), "=&v"(O[1]), "=&v"(O[2]), "=&v"(O[3]), "=&v"(T0), "=&v"(T1)
		    : "v"(Rc[0]), "v"(Rc[1]), "v"(Rc[2]), "v"(Rc[3]), "v"(pp0), "v"(pp1), "v"(S0), "v"(S1), "v"(S2), "v"(S3));
}

// inverse coupling (audio.rs:762-777) of four (magnitude, angle) pairs in place.  With c = (m > 0), d = (a > 0) the reference's
//   ap = c ? a : -a;  new_m = d ? m : m + ap;  new_a = d ? m - ap : m
// is v = m + ((c == d) ? -a : a);  (new_m, new_a) = d ? (m, v) : (v, m)   (x - y and x + (-y) are the same operation):
// 6 VALU + 1 SALU per pair, compare masks in SGPR pairs (no VCC serialisation, no hazard pads).
__device__ __forceinline__ void decouple4(float &m0, float &a0, float &m1, float &a1, float &m2, float &a2, float &m3, float &a3)
{
	float t0, t1, t2, t3;
	unsigned long long c0, c1, c2, c3, d0, d1, d2, d3;
	asm("v_cmp_lt_f32_e64 %12, 0, %0\n\tv_cmp_lt_f32_e64 %13, 0, %2\n\tv_cmp_lt_f32_e64 %14, 0, %4\n\tv_cmp_lt_f32_e64 %15, 0, %6\n\t"
	    "v_cmp_lt_f32_e64 %16, 0, %1\n\tv_cmp_lt_f32_e64 %17, 0, %3\n\tv_cmp_lt_f32_e64 %18, 0, %5\n\tv_cmp_lt_f32_e64 %19, 0, %7\n\t"
	    "s_xnor_b64 %12, %12, %16\n\ts_xnor_b64 %13, %13, %17\n\ts_xnor_b64 %14, %14, %18\n\ts_xnor_b64 %15, %15, %19\n\t"
	    "v_cndmask_b32_e64 %8, %1, -%1, %12\n\tv_cndmask_b32_e64 %9, %3, -%3, %13\n\t"
	    "v_cndmask_b32_e64 %10, %5, -%5, %14\n\tv_cndmask_b32_e64 %11, %7, -%7, %15\n\t"
	    "v_add_f32_e32 %8, %0, %8\n\tv_add_f32_e32 %9, %2, %9\n\tv_add_f32_e32 %10, %4, %10\n\tv_add_f32_e32 %11, %6, %11\n\t"
	    "v_cndmask_b32_e64 %1, %0, %8, %16\n\tv_cndmask_b32_e64 %3, %2, %9, %17\n\t"
	    "v_cndmask_b32_e64 %5, %4, %10, %18\n\tv_cndmask_b32_e64 %7, %6, %11, %19\n\t"
	    "v_cndmask_b32_e64 %0, %8, %0, %16\n\tv_cndmask_b32_e64 %2, %9, %2, %17\n\t"
	    "v_cndmask_b32_e64 %4, %10, %4, %18\n\tv_cndmask_b32_e64 %6, %11, %6, %19"
	    : "+v"(m0), "+v"(a0), "+v"(m1), "+v"(a1), "+v"(m2), "+v"(a2), "+v"(m3), "+v"(a3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3),
	      "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3), "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3));
}

// The same for the steps a k_long<..., PRE> wave runs under a switch on its program (apply_pre): `asm volatile`, so that the statement
// cannot be speculated -- with the plain form the compiler executed ALL twelve cases of the switch and selected the results
// (3 500 more VALU instructions per wave than the one case that runs)
__device__ __forceinline__ void decouple4_branchy(float &m0, float &a0, float &m1, float &a1, float &m2, float &a2, float &m3, float &a3)
{
	float t0, t1, t2, t3;
	unsigned long long c0, c1, c2, c3, d0, d1, d2, d3;
	asm volatile("v_cmp_lt_f32_e64 %12, 0, %0\n\tv_cmp_lt_f32_e64 %13, 0, %2\n\tv_cmp_lt_f32_e64 %14, 0, %4\n\tv_cmp_lt_f32_e64 %15, 0, %6\n\t"
	    "v_cmp_lt_f32_e64 %16, 0, %1\n\tv_cmp_lt_f32_e64 %17, 0, %3\n\tv_cmp_lt_f32_e64 %18, 0, %5\n\tv_cmp_lt_f32_e64 %19, 0, %7\n\t"
	    "s_xnor_b64 %12, %12, %16\n\ts_xnor_b64 %13, %13, %17\n\ts_xnor_b64 %14, %14, %18\n\ts_xnor_b64 %15, %15, %19\n\t"
	    "v_cndmask_b32_e64 %8, %1, -%1, %12\n\tv_cndmask_b32_e64 %9, %3, -%3, %13\n\t"
	    "v_cndmask_b32_e64 %10, %5, -%5, %14\n\tv_cndmask_b32_e64 %11, %7, -%7, %15\n\t"
	    "v_add_f32_e32 %8, %0, %8\n\tv_add_f32_e32 %9, %2, %9\n\tv_add_f32_e32 %10, %4, %10\n\tv_add_f32_e32 %11, %6, %11\n\t"
	    "v_cndmask_b32_e64 %1, %0, %8, %16\n\tv_cndmask_b32_e64 %3, %2, %9, %17\n\t"
	    "v_cndmask_b32_e64 %5, %4, %10, %18\n\tv_cndmask_b32_e64 %7, %6, %11, %19\n\t"
	    "v_cndmask_b32_e64 %0, %8, %0, %16\n\tv_cndmask_b32_e64 %2, %9, %2, %17\n\t"
	    "v_cndmask_b32_e64 %4, %10, %4, %18\n\tv_cndmask_b32_e64 %6, %11, %6, %19"
	    : "+v"(m0), "+v"(a0), "+v"(m1), "+v"(a1), "+v"(m2), "+v"(a2), "+v"(m3), "+v"(a3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3),
	      "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3), "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3));
}

// ---------------------------------------------------------------------------------------------
// Per-round register state
// ---------------------------------------------------------------------------------------------
// a work item in registers: every field a full dword (see load_item)
struct ItemRegs {
	uint32_t res_off, floor_off, out_off, src_arg;
	int32_t state_out;
	uint32_t halo_out, src_kind, flags;
	uint32_t pkt; // batch index of the packet: its entry in the edge buffer (EDGE instantiation only)
};

struct Pref {          // what one wave loads from HBM for one item
	float4_t r[2][4];  // residues: lane holds bins 4 (64 x + lane) + j of each channel
	uint32_t fe[2];    // floor-1 post entry of post `lane` of each channel (lanes >= F hold 0)
};


// The floor posts (58 bytes per channel) are requested BEFORE the residues: loads return in order, so a wave can wait for
// its floor record alone (s_waitcnt vmcnt(#residue loads)) and build the whole floor curve while the residues are in flight.
__device__ __forceinline__ void issue_floor_loads(const LwFastArgs &F, const ItemRegs &it, const LwFastUnit &un, uint32_t lane,
		Pref &p)
{
	const uint16_t *f0 = F.floors + it.floor_off + (uint32_t)un.ch_a * F.fstride;
	p.fe[0] = lane < un.F_a ? (uint32_t)f0[lane] : 0u;
	p.fe[1] = 0u;
	if (un.ch_b >= 0) {
		const uint16_t *f1 = F.floors + it.floor_off + (uint32_t)un.ch_b * F.fstride;
		p.fe[1] = lane < un.F_b ? (uint32_t)f1[lane] : 0u;
	}
}

// PAIR: the caller knows that the unit has two channels (every residue register is written: none stays live before the call)
// NT = false (k_long<..., PRE>): the vectors are read by several waves of the workgroup -- ordinary loads, the later readers meet them in L2
template <bool PAIR = false, bool NT = true>
__device__ __forceinline__ void issue_residue_loads(const LwFastArgs &F, const ItemRegs &it, const LwFastUnit &un,
		uint32_t lane, Pref &p)
{
	// non-temporal loads: the residues are read once -- streaming them past the caches instead of allocating 33 MB per launch
	// in L2 / MALL took the launch from 16.7 to 15.6 us (interleaved A/B, 2000 steps each)
	const float4_t *s0 = reinterpret_cast<const float4_t *>(F.residue + it.res_off + (uint32_t)un.ch_a * 1024u);
#pragma unroll
	for (int x = 0; x < 4; x++)
		p.r[0][x] = NT ? __builtin_nontemporal_load(&s0[64 * x + lane]) : s0[64 * x + lane];
	if (PAIR || un.ch_b >= 0) {
		const float4_t *s1 = reinterpret_cast<const float4_t *>(F.residue + it.res_off + (uint32_t)un.ch_b * 1024u);
#pragma unroll
		for (int x = 0; x < 4; x++)
			p.r[1][x] = NT ? __builtin_nontemporal_load(&s1[64 * x + lane]) : s1[64 * x + lane];
	}
}

// the ordinary order: residues first, the floor records behind them
template <bool NT = true>
__device__ __forceinline__ void issue_loads(const LwFastArgs &F, const ItemRegs &it, const LwFastUnit &un, uint32_t lane, Pref &p)
{
	issue_residue_loads<false, NT>(F, it, un, lane, p);
	issue_floor_loads(F, it, un, lane, p);
}

// ---- coupling steps inside the wave (k_long<..., PRE>; LwFastPlan::pre in lw_fast.hpp): up to two more raw channels of the packet ...
struct PreRegs {
	float4_t t[2][4]; // same lane layout as Pref::r
};

__device__ __forceinline__ void issue_pre_loads(const LwFastArgs &F, const ItemRegs &it, uint32_t prog, uint32_t lane, PreRegs &q)
{
#pragma unroll
	for (int k = 0; k < 2; k++) {
		const uint32_t c = (prog >> (8 * k)) & 0xffu;
		if (c != LW_PRE_NO_CH) { // (wave-uniform)
			const float4_t *s = reinterpret_cast<const float4_t *>(F.residue + it.res_off + c * 1024u);
#pragma unroll
			for (int x = 0; x < 4; x++)
				q.t[k][x] = s[64 * x + lane];
		} else {
#pragma unroll
			for (int x = 0; x < 4; x++)
				asm volatile("" : "=v"(q.t[k][x])); // (never read: defined, so that the previous round's values are not kept alive through the transform)
		}
	}
}

// ---- floor segment table of one channel (one 16-byte entry per static interval), built by lanes = posts:
//      {dy, c0, 1/adx, w}; the floor value of bin k is inverse_db[y(k)] with
//          y(k) = ((bits(fma(fma(k, dy, c0), 1/adx, w)) & 0x7fc) >> 2) - 1
//      i.e. two fused multiply-adds and one AND per bin, no conversion and no shift: w = 2^21 + 1 + y_base puts the sum into
//      [2^21, 2^22), where one ulp is 1/4, so the low mantissa bits ARE 4 (y + 1) + (two fraction bits) -- the byte offset of
//      the table entry (the - 4 goes into the ds_read's immediate offset).  render_line (audio.rs:503-524) is
//      y = y0 + trunc((k - x0) dy / adx); both signs are written as a FLOOR of something non-negative:
//          dy >= 0:  y = y0 + floor(((k - x0) dy + 1/2) / adx)                 c0 = 1/2 - x0 dy - adx/8,        y_base = y0
//          dy <  0:  y = y1 + floor(((x1 - k) |dy| + adx - 1/2) / adx)         c0 = 7 adx/8 - 1/2 - x1 dy,      y_base = y1
//      (the numerators are integers + 1/2, so the quotient's fraction lies in [1/(2 adx), 1 - 1/(2 adx)]; the - adx/8 moves
//      it to [-1/8 + 1/(2 adx), 7/8 - 1/(2 adx)], which rounds to a multiple of 1/4 in [0, 3/4]: the integer part is never
//      touched, whatever the tie rule.  Both inner sums are exact in f32; tests/test_fast_model.py checks every dy, every
//      adx <= 1024, every offset, with the reciprocal one ulp off either way, as v_rcp_f32 may be.)
#define LW_FLOOR_W0 2097153.0f // 2^21 + 1
#define LW_FLOOR_MASK 0x7fcu
__device__ __forceinline__ bool floor_table(const LwFastArgs &F, const char *img, char *sc, uint32_t lane, uint32_t e,
		uint32_t fslot, uint32_t Fp)
{
	const bool unused = __builtin_amdgcn_readfirstlane(e) == LW_FLOOR_UNUSED;
	const unsigned long long M = __ballot((e & LW_POST_ACTIVE) != 0);
	const unsigned long long lowmask = (2ull << lane) - 1ull;
	const unsigned long long below = M & lowmask, above = M & ~lowmask;
	const int lo = below ? 63 - __builtin_clzll(below) : 0;
	const int hi = above ? __builtin_ctzll(above) : lo;
	const int y = (int)(e & 0xffu);
	const float xs = *reinterpret_cast<const float *>(img + LWI_XSF + 4u * (64u * fslot + lane));
	const int ylo = __builtin_amdgcn_ds_bpermute(lo << 2, y);
	const int yhi = __builtin_amdgcn_ds_bpermute(hi << 2, y);
	const float xlo = __int_as_float(__builtin_amdgcn_ds_bpermute(lo << 2, __float_as_int(xs)));
	const float xhi = __int_as_float(__builtin_amdgcn_ds_bpermute(hi << 2, __float_as_int(xs)));
	const float dy = (float)(yhi - ylo); // 0 when there is no later active post (flat, audio.rs:546-548)
	const float adx = above ? xhi - xlo : 1.0f;
	const bool down = yhi < ylo;
	float4_t ent;
	ent.x = dy;
	ent.y = down ? (0.875f * adx - 0.5f) - xhi * dy : (0.5f - 0.125f * adx) - xlo * dy; // exact (22 bits at most)
	ent.z = above ? __builtin_amdgcn_rcpf(adx) : 1.0f;
	ent.w = (float)(down ? yhi : ylo) + LW_FLOOR_W0;
	if (lane < Fp)
		*reinterpret_cast<float4_t *>(sc + 16 * lane) = ent;
	return unused;
}

// floor value of bin kf through its interval entry (see floor_table)
__device__ __forceinline__ float floor_bin(float kf, float4_t ent)
{
	const float t = __builtin_fmaf(__builtin_fmaf(kf, ent.x, ent.y), ent.z, ent.w);
	return lds_abs_f32(LWI_INV_DB - 4u + (__float_as_uint(t) & LW_FLOOR_MASK));
}

// ---- floor value per bin; spectrum = floor * residue in place (audio.rs:1035-1037)
__device__ __forceinline__ void spectrum(const LwFastArgs &F, const char *img, const char *sc, uint32_t lane, uint32_t fslot,
		bool unused, float4_t (&r)[4])
{
	if (unused) {
#pragma unroll
		for (int x = 0; x < 4; x++)
			r[x] = r[x] * 0.0f; // zero floor (audio.rs:1021-1024)
		return;
	}
	const float kf0 = (float)(4 * (int)lane);
#pragma unroll
	for (int x = 0; x < 4; x++) {
		const uint2_t sw = *reinterpret_cast<const uint2_t *>(img + LWI_SID16 + 8u * ((fslot * 4 + x) * 64u + lane));
		const uint32_t s16[4] = {sw.x & 0xffffu, sw.x >> 16, sw.y & 0xffffu, sw.y >> 16};
		float4_t fl;
#pragma unroll
		for (int j = 0; j < 4; j++) {
			fl[j] = floor_bin(kf0 + (float)(256 * x + j), lds4(sc, s16[j]));
		}
		const float2_t lo2 = pk_mul(float2_t{fl.x, fl.y}, float2_t{r[x].x, r[x].y});
		const float2_t hi2 = pk_mul(float2_t{fl.z, fl.w}, float2_t{r[x].z, r[x].w});
		r[x] = float4_t{lo2.x, lo2.y, hi2.x, hi2.y};
	}
}

// ---- the same for both channels of a pair as ONE software pipeline over 8 steps of 4 bins (step b = 4 c + x):
//      the interval-entry gathers of step b+2 and the inverse-dB gathers of step b are in flight while step b+1's
//      indices are computed, so the three dependent LDS round trips per bin are paid once, not 24 times.
__device__ __forceinline__ void spectrum_pair(const char *img, const char *sc, uint32_t lane, uint32_t fslot_a, uint32_t fslot_b,
		float4_t (&r)[2][4])
{
	const float kf0 = (float)(4 * (int)lane);
	uint2_t sid[2][4];
#pragma unroll
	for (int x = 0; x < 4; x++) {
		sid[0][x] = *reinterpret_cast<const uint2_t *>(img + LWI_SID16 + 8u * ((fslot_a * 4 + x) * 64u + lane));
		sid[1][x] = *reinterpret_cast<const uint2_t *>(img + LWI_SID16 + 8u * ((fslot_b * 4 + x) * 64u + lane));
	}
	float4_t ent[2][4];
	float fl[2][4];
#define LW_SP_G(b)                                                                     \
	do {                                                                               \
		const uint2_t sw = sid[(b) >> 2][(b) & 3];                                     \
		const char *t = sc + 1024 * ((b) >> 2);                                        \
		ent[(b) & 1][0] = lds4(t, sw.x & 0xffffu);                                     \
		ent[(b) & 1][1] = lds4(t, sw.x >> 16);                                         \
		ent[(b) & 1][2] = lds4(t, sw.y & 0xffffu);                                     \
		ent[(b) & 1][3] = lds4(t, sw.y >> 16);                                         \
	} while (0)
#define LW_SP_XI(b)                                                                    \
	do {                                                                               \
		_Pragma("unroll") for (int j = 0; j < 4; j++) {                                \
			fl[(b) & 1][j] = floor_bin(kf0 + (float)(256 * ((b) & 3) + j), ent[(b) & 1][j]); \
		}                                                                              \
	} while (0)
#define LW_SP_M(b)                                                                     \
	do {                                                                               \
		float4_t &rr = r[(b) >> 2][(b) & 3];                                           \
		const float2_t lo2 = pk_mul(float2_t{fl[(b) & 1][0], fl[(b) & 1][1]}, float2_t{rr.x, rr.y}); \
		const float2_t hi2 = pk_mul(float2_t{fl[(b) & 1][2], fl[(b) & 1][3]}, float2_t{rr.z, rr.w}); \
		rr = float4_t{lo2.x, lo2.y, hi2.x, hi2.y};                                     \
	} while (0)
	LW_SP_G(0);
	LW_SP_G(1);
	__builtin_amdgcn_sched_barrier(0);
#pragma unroll
	for (int b = 0; b < 8; b++) {
		LW_SP_XI(b);
		if (b + 2 < 8)
			LW_SP_G(b + 2);
		if (b >= 1)
			LW_SP_M(b - 1);
		__builtin_amdgcn_sched_barrier(0);
	}
	LW_SP_M(7);
#undef LW_SP_G
#undef LW_SP_XI
#undef LW_SP_M
}

// ---- floor value per bin (audio.rs:552-554) of one channel: fl[x][j] = floor of bin 4 (64 x + lane) + j
__device__ __forceinline__ void floor_values(const char *img, const char *sc, uint32_t lane, uint32_t fslot, bool unused,
		float4_t (&fl)[4])
{
	if (unused) {
#pragma unroll
		for (int x = 0; x < 4; x++)
			fl[x] = float4_t{0.0f, 0.0f, 0.0f, 0.0f}; // zero floor (audio.rs:1021-1024)
		return;
	}
	const float kf0 = (float)(4 * (int)lane);
#pragma unroll
	for (int x = 0; x < 4; x++) {
		const uint2_t sw = *reinterpret_cast<const uint2_t *>(img + LWI_SID16 + 8u * ((fslot * 4 + x) * 64u + lane));
		const uint32_t s16[4] = {sw.x & 0xffffu, sw.x >> 16, sw.y & 0xffffu, sw.y >> 16};
#pragma unroll
		for (int j = 0; j < 4; j++) {
			fl[x][j] = floor_bin(kf0 + (float)(256 * x + j), lds4(sc, s16[j]));
		}
	}
}

// ---- the whole floor stage of one unit: segment tables (1 KB each in the wave's scratch) + floor value of every bin this
//      lane holds.  Needs only the 58-byte floor records, so it runs while the unit's residues are still on their way.
template <int NCH>
__device__ __forceinline__ void floor_phase(const LwFastArgs &F, const char *img, char *sc, uint32_t lane, const LwFastUnit &un,
		const uint32_t (&fe)[2], float4_t (&fl)[2][4])
{
	__builtin_amdgcn_s_setprio(LW_PRIO_FLOOR); // latency-bound phase (LDS round trips, few VALU): issue ahead of waves in the IMDCT
	const bool unused0 = floor_table(F, img, sc, lane, fe[0], un.floor_a, un.F_a);
	bool unused1 = false;
	if (NCH == 2)
		unused1 = floor_table(F, img, sc + 1024, lane, fe[1], un.floor_b, un.F_b);
	lds_fence();
	// (no software pipeline across the channels as in spectrum_pair: this runs while the wave would otherwise idle, and the
	// plain form needs half the registers)
	floor_values(img, sc, lane, un.floor_a, unused0, fl[0]);
	if (NCH == 2)
		floor_values(img, sc + 1024, lane, un.floor_b, unused1, fl[1]);
	lds_fence();
	__builtin_amdgcn_s_setprio(0);
}

// ---- IMDCT step 1 (imdct.rs:337-371) in the load layout, exchange with the mirror lane -> layout B;
//      step 2 (imdct.rs:385-430) and stages l = 0, 1 (imdct.rs:445-452); twiddles shared by the channels
template <int NCH>
__device__ __forceinline__ void stage_b(const LwFastArgs &F, const char *img, uint32_t lane, const float4_t (&r)[2][4],
		float2_t (&P)[2][8])
{
	const uint32_t mirror = (63u - lane) << 2;
	float2_t au[4], al[4];
#pragma unroll
	for (int x = 0; x < 4; x++) {
		const uint32_t m = 64u * x + lane;
		au[x] = lds2(img + LWI_APAIR, 8u * m);          // (A[2m], A[2m+1])
		al[x] = lds2(img + LWI_APAIR, 8u * (511u - m)); // (A[1022-2m], A[1023-2m])
	}
#pragma unroll
	for (int c = 0; c < NCH; c++) {
		float2_t U[4];
		step1x2(r[c][0], au[0], al[0], r[c][1], au[1], al[1], U[0], P[c][0], U[1], P[c][1]);
		step1x2(r[c][2], au[2], al[2], r[c][3], au[3], al[3], U[2], P[c][2], U[3], P[c][3]);
#pragma unroll
		for (int x = 0; x < 4; x++) { // pair 511 - m lives on the mirror lane
			P[c][7 - x].x = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(U[x].x)));
			P[c][7 - x].y = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(U[x].y)));
		}
	}
	float2_t s2[4], l0[2], l1;
#pragma unroll
	for (int x = 0; x < 4; x++)
		s2[x] = lds2(img + LWI_TW_S2, 8u * (64u * x + lane));
	l0[0] = lds2(img + LWI_TW_L0, 8u * lane);
	l0[1] = lds2(img + LWI_TW_L0, 8u * (64u + lane));
	l1 = lds2(img + LWI_TW_L1, 8u * lane);
#pragma unroll
	for (int c = 0; c < NCH; c++) {
		float2_t(&Q)[8] = P[c];
		bfly2x4(Q[4], Q[0], s2[0], Q[5], Q[1], s2[1], Q[6], Q[2], s2[2], Q[7], Q[3], s2[3]); // step 2
		bfly2x4(Q[2], Q[0], l0[0], Q[6], Q[4], l0[0], Q[3], Q[1], l0[1], Q[7], Q[5], l0[1]); // l = 0
		bfly2x4(Q[1], Q[0], l1, Q[3], Q[2], l1, Q[5], Q[4], l1, Q[7], Q[6], l1);             // l = 1
	}
}

// Exchange of one lane bit (1 or 0, i.e. inside a quad) with one register-index bit for eight (a, b) dword pairs:
// a' = bit ? b[partner lane] : a, b' = bit ? b : a[partner lane].  Per pair: two full-width v_mov_b32_dpp quad_perm into
// temporaries and two v_cndmask_b32_e64 under an SGPR-pair lane mask (4 x 4.3 cycles per SIMD).  NOT v_cndmask_b32_dpp: a
// VALU select whose mask is VCC -- the only form the DPP encoding has -- costs 16-23 cycles per instruction on gfx950,
// against 4.3 for the same select with the mask in an SGPR pair (tools/micro/op_cost.hip, profiles/r02_micro_op_cost.txt).
// The block works on four pairs at a time (eight temporaries); M = lanes whose bit is SET.
#define LW_XCHG4(QP, i0)                                                                                      \
	asm volatile("s_nop 1\n\t"                                                                                \
	             "v_mov_b32_dpp %8, %4 " QP " row_mask:0xf bank_mask:0xf\n\t"                                  \
	             "v_mov_b32_dpp %9, %5 " QP " row_mask:0xf bank_mask:0xf\n\t"                                  \
	             "v_mov_b32_dpp %10, %6 " QP " row_mask:0xf bank_mask:0xf\n\t"                                 \
	             "v_mov_b32_dpp %11, %7 " QP " row_mask:0xf bank_mask:0xf\n\t"                                 \
	             "v_mov_b32_dpp %12, %0 " QP " row_mask:0xf bank_mask:0xf\n\t"                                 \
	             "v_mov_b32_dpp %13, %1 " QP " row_mask:0xf bank_mask:0xf\n\t"                                 \
	             "v_mov_b32_dpp %14, %2 " QP " row_mask:0xf bank_mask:0xf\n\t"                                 \
	             "v_mov_b32_dpp %15, %3 " QP " row_mask:0xf bank_mask:0xf\n\t"                                 \
	             "v_cndmask_b32_e64 %0, %0, %8, %16\n\t"                                                       \
	             "v_cndmask_b32_e64 %1, %1, %9, %16\n\t"                                                       \
	             "v_cndmask_b32_e64 %2, %2, %10, %16\n\t"                                                      \
	             "v_cndmask_b32_e64 %3, %3, %11, %16\n\t"                                                      \
	             "v_cndmask_b32_e64 %4, %12, %4, %16\n\t"                                                      \
	             "v_cndmask_b32_e64 %5, %13, %5, %16\n\t"                                                      \
	             "v_cndmask_b32_e64 %6, %14, %6, %16\n\t"                                                      \
	             "v_cndmask_b32_e64 %7, %15, %7, %16"                                                           \
	             : "+v"(a[i0]), "+v"(a[i0 + 1]), "+v"(a[i0 + 2]), "+v"(a[i0 + 3]), "+v"(b[i0]), "+v"(b[i0 + 1]),   \
	               "+v"(b[i0 + 2]), "+v"(b[i0 + 3]), "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), \
	               "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])                                                        \
	             : "s"(M))

__device__ __forceinline__ void xq_bit1(float (&a)[8], float (&b)[8]) // partner = lane ^ 2
{
	const unsigned long long M = 0xccccccccccccccccull;
	float t[8];
	LW_XCHG4("quad_perm:[2,3,0,1]", 0);
	LW_XCHG4("quad_perm:[2,3,0,1]", 4);
}
__device__ __forceinline__ void xq_bit0(float (&a)[8], float (&b)[8]) // partner = lane ^ 1
{
	const unsigned long long M = 0xaaaaaaaaaaaaaaaaull;
	float t[8];
	LW_XCHG4("quad_perm:[1,0,3,2]", 0);
	LW_XCHG4("quad_perm:[1,0,3,2]", 4);
}

// ---- T2: layout B -> C is an 8 x 8 transpose between the register index (pair bits 8..6) and lane bits 5..3.
//      Three butterfly exchanges: lane bit 5 by v_permlane32_swap, bit 4 by v_permlane16_swap, bit 3 by DPP row_ror:8 with
//      bank masks.  40 VALU per channel instead of 8 ds_write_b64 + 8 ds_read_b64: the LDS write data path (one per CU,
//      ~7.6 cycles per ds_write_b64) is the scarcer resource (tools/exp.sh: doubling the transposes' writes costs 1.85 us).
__device__ __forceinline__ void t2_inreg(float2_t (&P)[8])
{
#define LW_SWAP32(a, b)                                                                       \
	do {                                                                                      \
		auto r_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false); \
		a = __uint_as_float(r_[0]);                                                           \
		b = __uint_as_float(r_[1]);                                                           \
	} while (0)
#define LW_SWAP16(a, b)                                                                       \
	do {                                                                                      \
		auto r_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false); \
		a = __uint_as_float(r_[0]);                                                           \
		b = __uint_as_float(r_[1]);                                                           \
	} while (0)
#define LW_SWAP8(a, b)                                                                        \
	do {                                                                                      \
		const int a_ = __float_as_int(a), b_ = __float_as_int(b);                             \
		a = __int_as_float(__builtin_amdgcn_update_dpp(a_, b_, 0x128, 0xf, 0xc, false)); /* lanes 8-15 of a row <- b[l ^ 8] */ \
		b = __int_as_float(__builtin_amdgcn_update_dpp(b_, a_, 0x128, 0xf, 0x3, false)); /* lanes 0-7  of a row <- a[l ^ 8] */ \
	} while (0)
#pragma unroll
	for (int x = 0; x < 4; x++) { // register bit 2 <-> lane bit 5
		LW_SWAP32(P[x].x, P[x + 4].x);
		LW_SWAP32(P[x].y, P[x + 4].y);
	}
#pragma unroll
	for (int i = 0; i < 4; i++) { // register bit 1 <-> lane bit 4
		const int x = (i & 1) | ((i & 2) << 1);
		LW_SWAP16(P[x].x, P[x + 2].x);
		LW_SWAP16(P[x].y, P[x + 2].y);
	}
#pragma unroll
	for (int x = 0; x < 8; x += 2) { // register bit 0 <-> lane bit 3 (tied DPP moves under bank masks + one copy per pair)
		LW_SWAP8(P[x].x, P[x + 1].x);
		LW_SWAP8(P[x].y, P[x + 1].y);
	}
#undef LW_SWAP32
#undef LW_SWAP16
#undef LW_SWAP8
}

// ---- stages l = 2, 3, 4 (imdct.rs:454-477)
template <int NCH>
__device__ __forceinline__ void stage_c(const LwFastArgs &F, const char *img, uint32_t lane, float2_t (&P)[2][8])
{
	const uint32_t lo3 = lane & 7u;
	float2_t t2[4], t3[2], t4;
#pragma unroll
	for (int yy = 0; yy < 4; yy++)
		t2[yy] = lds2(img + LWI_TW_L2, 8u * (8u * yy + lo3));
	t3[0] = lds2(img + LWI_TW_L3, 8u * lo3);
	t3[1] = lds2(img + LWI_TW_L3, 8u * (8u + lo3));
	t4 = lds2(img + LWI_TW_L4, 8u * lo3);
#pragma unroll
	for (int c = 0; c < NCH; c++) {
		float2_t(&Q)[8] = P[c];
		bfly2x4(Q[4], Q[0], t2[0], Q[5], Q[1], t2[1], Q[6], Q[2], t2[2], Q[7], Q[3], t2[3]); // l = 2
		bfly2x4(Q[2], Q[0], t3[0], Q[6], Q[4], t3[0], Q[3], Q[1], t3[1], Q[7], Q[5], t3[1]); // l = 3
		bfly2x4(Q[1], Q[0], t4, Q[3], Q[2], t4, Q[5], Q[4], t4, Q[7], Q[6], t4);             // l = 4
	}
}

// ---- T3: layout C -> D is an 8 x 8 transpose between the register index (pair bits 5..3) and lane bits 2..0.
//      Lane bit 2: v_mov_b32_dpp row_shr / row_shl:4 with bank masks (tied: one copy per pair).  Lane bits 1 and 0 (inside a
//      quad, where bank masks cannot select): LW_XCHG4, 4 instructions per exchanged dword pair.
__device__ __forceinline__ void t3_inreg(float2_t (&P)[8])
{
#define LW_X4(a, b)                                                                           \
	do {                                                                                      \
		const int a_ = __float_as_int(a), b_ = __float_as_int(b);                             \
		a = __int_as_float(__builtin_amdgcn_update_dpp(a_, b_, 0x114, 0xf, 0xa, false)); /* lanes 4-7, 12-15 <- b[l - 4] */ \
		b = __int_as_float(__builtin_amdgcn_update_dpp(b_, a_, 0x104, 0xf, 0x5, false)); /* lanes 0-3, 8-11  <- a[l + 4] */ \
	} while (0)
#pragma unroll
	for (int y = 0; y < 4; y++) { // register bit 2 <-> lane bit 2
		LW_X4(P[y].x, P[y + 4].x);
		LW_X4(P[y].y, P[y + 4].y);
	}
#undef LW_X4
	float a[8], b[8];
#pragma unroll
	for (int i = 0; i < 4; i++) { // register bit 1 <-> lane bit 1
		const int y = (i & 1) | ((i & 2) << 1);
		a[2 * i] = P[y].x, a[2 * i + 1] = P[y].y, b[2 * i] = P[y + 2].x, b[2 * i + 1] = P[y + 2].y;
	}
	xq_bit1(a, b);
#pragma unroll
	for (int i = 0; i < 4; i++) {
		const int y = (i & 1) | ((i & 2) << 1);
		P[y] = float2_t{a[2 * i], a[2 * i + 1]}, P[y + 2] = float2_t{b[2 * i], b[2 * i + 1]};
	}
#pragma unroll
	for (int i = 0; i < 4; i++) // register bit 0 <-> lane bit 0
		a[2 * i] = P[2 * i].x, a[2 * i + 1] = P[2 * i].y, b[2 * i] = P[2 * i + 1].x, b[2 * i + 1] = P[2 * i + 1].y;
	xq_bit0(a, b);
#pragma unroll
	for (int i = 0; i < 4; i++)
		P[2 * i] = float2_t{a[2 * i], a[2 * i + 1]}, P[2 * i + 1] = float2_t{b[2 * i], b[2 * i + 1]};
}

// ---- T4: layout D -> bit-reverse gather.  slot(p) = (p & ~127) | ((p & 3) << 5) | ((p >> 2) & 31)
__device__ __forceinline__ void t4_write(char *sc, uint32_t lane, const float2_t (&Z)[8])
{
	const uint32_t base = 128u * (lane >> 4) + 2u * (lane & 15u);
#pragma unroll
	for (int zz = 0; zz < 8; zz++) {
		const uint32_t slot = base + 32u * (zz & 3) + (zz >> 2);
		*reinterpret_cast<float2_t *>(sc + 8u * slot) = Z[zz];
	}
}

struct TwidE { // tables of layout E for one c2
	float4_t Cq, Bl, Bh;
};

// ---- bit-reverse gather (imdct.rs:490-528), step 7 (:533-580), step 8 (:589-658) of m' = 2 lane + c2
__device__ __forceinline__ void stage_e(const char *sc, uint32_t lane, int c2, const TwidE &tw, float2_t (&R)[4])
{
	// rho = rev6(lane); s0 = slot(2 rho): pairs 2v, 2v+256, 255-2v, 511-2v of m' = 2 lane + c2
	const uint32_t rho = __builtin_bitreverse32(lane) >> 26;
	const uint32_t s0 = ((rho & 1u) << 6) | (rho >> 1);
	const uint32_t sa = 128u * c2 + s0; // slot of pair 2v
	const uint32_t sb = 255u - sa;      // slot of pair 255 - 2v
	const float2_t pq = lds2(sc, 8u * sa), pq256 = lds2(sc, 8u * (sa + 256u));   // (E3,E2), (E1,E0)
	const float2_t p255 = lds2(sc, 8u * sb), p511 = lds2(sc, 8u * (sb + 256u));  // (D3,D2), (D1,D0)
	step78_block(p511, pq, p255, pq256, tw.Cq, tw.Bl, tw.Bh, R);
}

// ---- per-channel pieces of the stages with the twiddles passed in (shared by the two channels of a pair)
struct TwB {
	float2_t au[4], al[4], s2[4], l0[2], l1;
};
struct TwC {
	float2_t t2[4], t3[2], t4;
};

__device__ __forceinline__ void load_tw_b(const char *img, uint32_t lane, TwB &t)
{
#pragma unroll
	for (int x = 0; x < 4; x++) {
		const uint32_t m = 64u * x + lane;
		t.au[x] = lds2(img + LWI_APAIR, 8u * m);          // (A[2m], A[2m+1])
		t.al[x] = lds2(img + LWI_APAIR, 8u * (511u - m)); // (A[1022-2m], A[1023-2m])
		t.s2[x] = lds2(img + LWI_TW_S2, 8u * (64u * x + lane));
	}
	t.l0[0] = lds2(img + LWI_TW_L0, 8u * lane);
	t.l0[1] = lds2(img + LWI_TW_L0, 8u * (64u + lane));
	t.l1 = lds2(img + LWI_TW_L1, 8u * lane);
}

__device__ __forceinline__ void load_tw_c(const char *img, uint32_t lane, TwC &t)
{
	const uint32_t lo3 = lane & 7u;
#pragma unroll
	for (int yy = 0; yy < 4; yy++)
		t.t2[yy] = lds2(img + LWI_TW_L2, 8u * (8u * yy + lo3));
	t.t3[0] = lds2(img + LWI_TW_L3, 8u * lo3);
	t.t3[1] = lds2(img + LWI_TW_L3, 8u * (8u + lo3));
	t.t4 = lds2(img + LWI_TW_L4, 8u * lo3);
}

// step 1 + mirror exchange + step 2 + stages l = 0, 1 of one channel
__device__ __forceinline__ void stage_b1(const TwB &t, uint32_t lane, const float4_t (&r)[4], float2_t (&Q)[8])
{
	const uint32_t mirror = (63u - lane) << 2;
	float2_t U[4];
	step1x2(r[0], t.au[0], t.al[0], r[1], t.au[1], t.al[1], U[0], Q[0], U[1], Q[1]);
	step1x2(r[2], t.au[2], t.al[2], r[3], t.au[3], t.al[3], U[2], Q[2], U[3], Q[3]);
#pragma unroll
	for (int x = 0; x < 4; x++) { // pair 511 - m lives on the mirror lane
		Q[7 - x].x = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(U[x].x)));
		Q[7 - x].y = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(U[x].y)));
	}
	bfly2x4(Q[4], Q[0], t.s2[0], Q[5], Q[1], t.s2[1], Q[6], Q[2], t.s2[2], Q[7], Q[3], t.s2[3]); // step 2
	bfly2x4(Q[2], Q[0], t.l0[0], Q[6], Q[4], t.l0[0], Q[3], Q[1], t.l0[1], Q[7], Q[5], t.l0[1]); // l = 0
	bfly2x4(Q[1], Q[0], t.l1, Q[3], Q[2], t.l1, Q[5], Q[4], t.l1, Q[7], Q[6], t.l1);             // l = 1
}

__device__ __forceinline__ void stage_c1(const TwC &t, float2_t (&Q)[8])
{
	bfly2x4(Q[4], Q[0], t.t2[0], Q[5], Q[1], t.t2[1], Q[6], Q[2], t.t2[2], Q[7], Q[3], t.t2[3]); // l = 2
	bfly2x4(Q[2], Q[0], t.t3[0], Q[6], Q[4], t.t3[0], Q[3], Q[1], t.t3[1], Q[7], Q[5], t.t3[1]); // l = 3
	bfly2x4(Q[1], Q[0], t.t4, Q[3], Q[2], t.t4, Q[5], Q[4], t.t4, Q[7], Q[6], t.t4);             // l = 4
}

__device__ __forceinline__ void stage_e1(const char *img, const char *sc, uint32_t lane, float2_t (&Rc)[2][4])
{
#pragma unroll
	for (int c2 = 0; c2 < 2; c2++) {
		TwidE te;
		te.Cq = lds4(img + LWI_C4, 16u * (64u * c2 + lane));
		te.Bl = lds4(img + LWI_B_LO, 16u * (64u * c2 + lane));
		te.Bh = lds4(img + LWI_B_HI, 16u * (64u * c2 + lane));
		stage_e(sc, lane, c2, te, Rc[c2]);
	}
}

// ---- the IMDCT of a channel pair as one software pipeline through ONE 4 KB transpose buffer (LDS operations of a
//      wave execute in order): while channel 0's transposed data is on its way back from LDS, channel 1's
//      butterflies issue, and vice versa:
//        B0 W2_0 R2_0 | B1 W2_1 R2_1 | C0 W3_0 R3_0 | C1 W3_1 R3_1 | D0 W4_0 E0 | D1 W4_1 E1
//      (sched_barrier pins this order; without it the scheduler moves each consumer right behind its loads)
__device__ __forceinline__ void imdct_pair(const char *img, char *sc, uint32_t lane, const float4_t (&r)[2][4],
		float2_t (&R)[2][2][4])
{
	float2_t P0[8], P1[8];
	TwC tc;
	{
		TwB tb;
		load_tw_b(img, lane, tb);
		stage_b1(tb, lane, r[0], P0);
		stage_b1(tb, lane, r[1], P1);
		load_tw_c(img, lane, tc);
		t2_inreg(P0);
		t2_inreg(P1);
		__builtin_amdgcn_sched_barrier(0);
	}
	stage_c1(tc, P0);
	__builtin_amdgcn_sched_barrier(0);
	const float a2s = *reinterpret_cast<const float *>(img + LWI_A2);
	const float2_t a2 = float2_t{a2s, a2s};
	t3_inreg(P0);
	stage_c1(tc, P1);
	__builtin_amdgcn_sched_barrier(0);
	t3_inreg(P1);
	stage_d_block(a2, P0);
	__builtin_amdgcn_sched_barrier(0);
	t4_write(sc, lane, P0);
	__builtin_amdgcn_sched_barrier(0);
	stage_e1(img, sc, lane, R[0]);
	stage_d_block(a2, P1);
	__builtin_amdgcn_sched_barrier(0);
	t4_write(sc, lane, P1);
	__builtin_amdgcn_sched_barrier(0);
	stage_e1(img, sc, lane, R[1]);
	__builtin_amdgcn_sched_barrier(0);
}

// ---- inverse coupling (audio.rs:762-777, :990-1002) of a coupled pair on the raw residues
__device__ __forceinline__ void decouple_pair(Pref &pf)
{
#pragma unroll
	for (int x = 0; x < 4; x++) {
		float m[4] = {pf.r[0][x].x, pf.r[0][x].y, pf.r[0][x].z, pf.r[0][x].w};
		float a[4] = {pf.r[1][x].x, pf.r[1][x].y, pf.r[1][x].z, pf.r[1][x].w};
		decouple4(m[0], a[0], m[1], a[1], m[2], a[2], m[3], a[3]);
		pf.r[0][x] = float4_t{m[0], m[1], m[2], m[3]};
		pf.r[1][x] = float4_t{a[0], a[1], a[2], a[3]};
	}
}

// ... and up to three inverse-coupling steps (audio.rs:762-777) on the registers r0 / r1 (the unit's channels) and t0 / t1, each
// (magnitude, angle) -> (magnitude', angle') in place, before the unit's own step: op = magnitude register << 2 | angle register
__device__ __forceinline__ void couple_quad(float4_t &M, float4_t &A)
{
	float m[4] = {M.x, M.y, M.z, M.w};
	float a[4] = {A.x, A.y, A.z, A.w};
	decouple4_branchy(m[0], a[0], m[1], a[1], m[2], a[2], m[3], a[3]);
	M = float4_t{m[0], m[1], m[2], m[3]};
	A = float4_t{a[0], a[1], a[2], a[3]};
}

// (quarter by quarter of the vectors: the program runs on 16 registers at a time -- with whole vectors under the switch the
// compiler spilled 300-500 registers around it)
__device__ __forceinline__ void apply_pre(uint32_t prog, Pref &pf, PreRegs &q)
{
	const uint32_t n = prog >> 28;
#pragma unroll
	for (int x = 0; x < 4; x++) {
		float4_t v0 = pf.r[0][x], v1 = pf.r[1][x], v2 = q.t[0][x], v3 = q.t[1][x];
#pragma nounroll
		for (uint32_t i = 0; i < n; i++) { // (wave-uniform control: scalar branches)
			switch ((prog >> (16u + 4u * i)) & 0xfu) {
			case 0x1: couple_quad(v0, v1); break;
			case 0x2: couple_quad(v0, v2); break;
			case 0x3: couple_quad(v0, v3); break;
			case 0x4: couple_quad(v1, v0); break;
			case 0x6: couple_quad(v1, v2); break;
			case 0x7: couple_quad(v1, v3); break;
			case 0x8: couple_quad(v2, v0); break;
			case 0x9: couple_quad(v2, v1); break;
			case 0xb: couple_quad(v2, v3); break;
			case 0xc: couple_quad(v3, v0); break;
			case 0xd: couple_quad(v3, v1); break;
			case 0xe: couple_quad(v3, v2); break;
			default: break;
			}
		}
		pf.r[0][x] = v0;
		pf.r[1][x] = v1;
	}
}

// ---- from the landed residues to the spectrum, floor stage included (the path of a wave whose residues were requested
//      before it had time for the floor stage): segment tables, inverse coupling, floor x residue fused into the gathers
template <int NCH, bool SPLIT = false>
__device__ __forceinline__ void spectrum_fused(const LwFastArgs &F, const char *img, char *sc, uint32_t lane,
		const LwFastUnit &un, Pref &pf)
{
	__builtin_amdgcn_s_setprio(LW_PRIO_FLOOR); // latency-bound phase (LDS round trips, few VALU): issue ahead of waves in the IMDCT
	const bool unused0 = floor_table(F, img, sc, lane, pf.fe[0], un.floor_a, un.F_a);
	bool unused1 = false;
	if (NCH == 2)
		unused1 = floor_table(F, img, sc + 1024, lane, pf.fe[1], un.floor_b, un.F_b);
	lds_fence();
	if (NCH == 2 && un.coupled)
		decouple_pair(pf);
	if (SPLIT && NCH == 1 && un.coupled >= LW_UNIT_SPLIT_MAG) {
		// half of a coupled pair (sparse launches, lw_fast.hpp): both raw vectors are here, r[0] = this wave's channel, r[1] =
		// its partner; after the inverse coupling only this wave's channel goes on
		if (un.coupled == LW_UNIT_SPLIT_ANG) { // decouple_pair wants (magnitude, angle)
#pragma unroll
			for (int x = 0; x < 4; x++) {
				const float4_t t = pf.r[0][x];
				pf.r[0][x] = pf.r[1][x];
				pf.r[1][x] = t;
			}
			decouple_pair(pf);
#pragma unroll
			for (int x = 0; x < 4; x++)
				pf.r[0][x] = pf.r[1][x];
		} else {
			decouple_pair(pf);
		}
	}
	if (NCH == 2 && !unused0 && !unused1) {
		spectrum_pair(img, sc, lane, un.floor_a, un.floor_b, pf.r);
	} else {
		spectrum(F, img, sc, lane, un.floor_a, unused0, pf.r[0]);
		if (NCH == 2)
			spectrum(F, img, sc + 1024, lane, un.floor_b, unused1, pf.r[1]);
	}
	lds_fence();
}

// ---- the same with the floor values already in registers (floor_phase ran while the residues were in flight):
//      inverse coupling and one multiply per bin (audio.rs:1035-1037)
__device__ __forceinline__ void spectrum_ready(const LwFastUnit &un, Pref &pf, const float4_t (&fl)[2][4])
{
	if (un.coupled)
		decouple_pair(pf);
#pragma unroll
	for (int c = 0; c < 2; c++)
#pragma unroll
		for (int x = 0; x < 4; x++) {
			const float2_t lo2 = pk_mul(float2_t{fl[c][x].x, fl[c][x].y}, float2_t{pf.r[c][x].x, pf.r[c][x].y});
			const float2_t hi2 = pk_mul(float2_t{fl[c][x].z, fl[c][x].w}, float2_t{pf.r[c][x].z, pf.r[c][x].w});
			pf.r[c][x] = float4_t{lo2.x, lo2.y, hi2.x, hi2.y};
		}
}

// ---- IMDCT of the unit's spectrum up to the un-windowed halves (pa, pb) (wave-private).  Both channels of a pair advance
// through the stages together (shared twiddles); their transposes go through ONE 4 KB buffer one after the other (LDS
// operations of a wave execute in order).
// UP (k_mix): the long blocks' waves share their SIMDs with short blocks' waves that will wait for them: one priority level up
template <int NCH, int UP = 0>
__device__ __forceinline__ void long_imdct(const LwFastArgs &F, const char *img, char *sc, uint32_t lane, Pref &pf,
		float2_t (&R)[2][2][4])
{
	// the last waves to get their data (the launch ends when they do) run the IMDCT one priority level up
	if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) >= LW_IMDCT_PRIO_LATE)
		__builtin_amdgcn_s_setprio(LW_PRIO_IMDCT + 1 + UP);
	else
		__builtin_amdgcn_s_setprio(LW_PRIO_IMDCT + UP);
	if (NCH == 2) {
		imdct_pair(img, sc, lane, pf.r, R);
		return;
	}
	float2_t P[2][8];
	stage_b<NCH>(F, img, lane, pf.r, P);
#pragma unroll
	for (int c = 0; c < NCH; c++) // T2
		t2_inreg(P[c]);
	stage_c<NCH>(F, img, lane, P);
#pragma unroll
	for (int c = 0; c < NCH; c++) { // T3
		t3_inreg(P[c]);
	}
	{
		const float a2s = *reinterpret_cast<const float *>(img + LWI_A2);
		const float2_t a2 = float2_t{a2s, a2s};
#pragma unroll
		for (int c = 0; c < NCH; c++)
			stage_d_block(a2, P[c]);
	}
#pragma unroll
	for (int c = 0; c < NCH; c++) { // T4 + layout E
		t4_write(sc, lane, P[c]);
		lds_fence();
#pragma unroll
		for (int c2 = 0; c2 < 2; c2++) {
			TwidE te;
			te.Cq = lds4(img + LWI_C4, 16u * (64u * c2 + lane));
			te.Bl = lds4(img + LWI_B_LO, 16u * (64u * c2 + lane));
			te.Bh = lds4(img + LWI_B_HI, 16u * (64u * c2 + lane));
			stage_e(sc, lane, c2, te, R[c][c2]);
		}
		lds_fence();
	}
}

// ---- publish the un-windowed right half for the successor ([channel][c2][lane] float4)
template <int NCH>
__device__ __forceinline__ void publish(char *dst, uint32_t lane, const float2_t (&R)[2][2][4])
{
#pragma unroll
	for (int c = 0; c < 2; c++)
		if (c < NCH) {
#pragma unroll
			for (int c2 = 0; c2 < 2; c2++)
				*reinterpret_cast<float4_t *>(dst + 2048 * c + 16u * (64u * c2 + lane)) =
					float4_t{R[c][c2][0].y, R[c][c2][1].y, R[c][c2][2].y, R[c][c2][3].y};
		}
}

// pb at this lane's q positions: group 0 = q in [4 lane, +4), group 1 = q in [508 - 4 lane, +4) (ascending)
#define LW_PB_LO0(c) float4_t{R[c][0][3].y, R[c][0][2].y, R[c][1][3].y, R[c][1][2].y}
#define LW_PB_LO1(c) float4_t{R[c][1][1].y, R[c][1][0].y, R[c][0][1].y, R[c][0][0].y}

// previous right half of one channel as the overlap-add consumes it: pp[c2][0] = (k=0, k=1), pp[c2][1] = (k=2, k=3)
struct PrevHalf {
	float2_t pp[2][2];
};

__device__ __forceinline__ void prev_from_lds(const char *src, uint32_t lane, PrevHalf &h)
{
#pragma unroll
	for (int c2 = 0; c2 < 2; c2++) {
		const float4_t v = *reinterpret_cast<const float4_t *>(src + 16u * (64u * c2 + lane));
		h.pp[c2][0] = float2_t{v.x, v.y};
		h.pp[c2][1] = float2_t{v.z, v.w};
	}
}

__device__ __forceinline__ void prev_from_global(const float *g, uint32_t lane, PrevHalf &h)
{
	const float4_t g0 = *reinterpret_cast<const float4_t *>(g + 4u * lane);        // q = 4l .. 4l+3
	const float4_t g1 = *reinterpret_cast<const float4_t *>(g + 508u - 4u * lane); // q = 508-4l ..
	h.pp[0][1] = float2_t{g0.y, g0.x}; // (c2=0: k=2 -> q=4l+1, k=3 -> q=4l)
	h.pp[1][1] = float2_t{g0.w, g0.z}; // (c2=1: k=2 -> 4l+3, k=3 -> 4l+2)
	h.pp[1][0] = float2_t{g1.y, g1.x}; // (c2=1: k=0 -> 509-4l, k=1 -> 508-4l)
	h.pp[0][0] = float2_t{g1.w, g1.z}; // (c2=0: k=0 -> 511-4l, k=1 -> 510-4l)
}

// 8-byte PCM store as a write-through (sc1) store: the bytes leave the L2 while the kernel is still running instead of
// staying dirty until the end-of-kernel write-back (16.8 MB of dirty PCM cost ~2.7 us at every kernel boundary)
__device__ __forceinline__ void store_pcm8(void *p, uint32_t lo, uint32_t hi)
{
	__hip_atomic_store(reinterpret_cast<unsigned long long *>(p), ((unsigned long long)hi << 32) | lo, __ATOMIC_RELAXED,
			__HIP_MEMORY_SCOPE_AGENT);
}

// 16-byte write-through store (f32 PCM, stream state).  Inline asm because the builtin path offers sc1 only up to 8
// bytes; the trailing s_nop keeps hipcc from overwriting the data registers before the store has read them.
__device__ __forceinline__ void store16_wt(void *p, float4_t v)
{
	asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// kernel-internal output format: LW_OUT_I16_INTERLEAVED of a 2-channel stream whose channels form ONE unit (a coupled pair)
#define LW_OUT_I16_ITL_STEREO 3

// ---- interleaved stereo (InterleavedSamples<i16>, samples.rs:48-78; `read_dec_packet_itl`): one channel's window +
//      overlap-add + conversion, packed two samples per dword: D[g][h] = positions (p_g + 2h, p_g + 2h + 1), with
//      p_0 = 4l, p_1 = 508 - 4l, p_2 = 512 + 4l, p_3 = 1020 - 4l
__device__ __forceinline__ void ola_pack_i16(const char *img, uint32_t lane, const float2_t (&Rc)[2][4], const PrevHalf &h,
		uint32_t (&D)[4][2])
{
	float2_t O[2][4];
#pragma unroll
	for (int c2 = 0; c2 < 2; c2++) {
		const float4_t w0 = lds4(img + LWI_WIN, 32u * (64u * c2 + lane));
		const float4_t w1 = lds4(img + LWI_WIN, 32u * (64u * c2 + lane) + 16u);
		ola_block<true>(Rc[c2], h.pp[c2][0], h.pp[c2][1], w0, w1, O[c2]);
	}
	typedef short short2_t __attribute__((ext_vector_type(2)));
	union {
		short2_t s;
		uint32_t u;
	} a;
#define LW_PKI(d, v0, v1) a.s = __builtin_amdgcn_cvt_pk_i16((int)(v0), (int)(v1)), d = a.u
	LW_PKI(D[0][0], O[0][3].x, O[0][2].x);
	LW_PKI(D[0][1], O[1][3].x, O[1][2].x);
	LW_PKI(D[1][0], O[1][1].x, O[1][0].x);
	LW_PKI(D[1][1], O[0][1].x, O[0][0].x);
	LW_PKI(D[2][0], O[0][0].y, O[0][1].y);
	LW_PKI(D[2][1], O[1][0].y, O[1][1].y);
	LW_PKI(D[3][0], O[1][2].y, O[1][3].y);
	LW_PKI(D[3][1], O[0][2].y, O[0][3].y);
#undef LW_PKI
}

// (L, R) pairs of four consecutive positions = 16 contiguous bytes per group: 2 v_perm_b32 per dword pair, 4 stores
__device__ __forceinline__ void store_interleaved2(const LwFastArgs &F, uint32_t lane, uint32_t out_off,
		const uint32_t (&L)[4][2], const uint32_t (&R)[4][2])
{
	int16_t *o = reinterpret_cast<int16_t *>(F.out) + out_off;
	const uint32_t pos[4] = {4u * lane, 508u - 4u * lane, 512u + 4u * lane, 1020u - 4u * lane};
#pragma unroll
	for (int g = 0; g < 4; g++) {
		const uint32_t d0 = __builtin_amdgcn_perm(R[g][0], L[g][0], 0x05040100u); // (L[p], R[p])
		const uint32_t d1 = __builtin_amdgcn_perm(R[g][0], L[g][0], 0x07060302u); // (L[p+1], R[p+1])
		const uint32_t d2 = __builtin_amdgcn_perm(R[g][1], L[g][1], 0x05040100u);
		const uint32_t d3 = __builtin_amdgcn_perm(R[g][1], L[g][1], 0x07060302u);
		store16_wt(o + 2u * pos[g], float4_t{__uint_as_float(d0), __uint_as_float(d1), __uint_as_float(d2), __uint_as_float(d3)});
	}
}

// ---- window + overlap-add (audio.rs:1116-1118), sample conversion (samples.rs:92-103), stores of one channel
// (mch = samples per channel of the packet's output block: 1024 unless the block has a short right slope, EDGE kernels)
template <int FMT>
__device__ __forceinline__ void ola_store(const LwFastArgs &F, const char *img, uint32_t lane, int chn,
		uint32_t out_off, const float2_t (&Rc)[2][4], const PrevHalf &h, uint32_t mch = 1024u)
{
	// (out[q], out[1023-q]) = (pa s[q] + pb' s[r], -pa s[r] + pb' s[q]), times 32768 for the i16 formats
	float2_t O[2][4];
#pragma unroll
	for (int c2 = 0; c2 < 2; c2++) {
		const float4_t w0 = lds4(img + LWI_WIN, 32u * (64u * c2 + lane));
		const float4_t w1 = lds4(img + LWI_WIN, 32u * (64u * c2 + lane) + 16u);
		ola_block<FMT != LW_OUT_F32_PLANAR>(Rc[c2], h.pp[c2][0], h.pp[c2][1], w0, w1, O[c2]);
	}
	// positions: [4l..4l+3] = .x of (0,3) (0,2) (1,3) (1,2); [508-4l..] = .x of (1,1) (1,0) (0,1) (0,0)
	//            [512+4l..] = .y of (0,0) (0,1) (1,0) (1,1); [1020-4l..] = .y of (1,2) (1,3) (0,2) (0,3)
	const uint32_t p0 = 4u * lane, p1 = 508u - 4u * lane, p2 = 512u + 4u * lane, p3 = 1020u - 4u * lane;
	if (FMT == LW_OUT_F32_PLANAR) {
		float *o = reinterpret_cast<float *>(F.out) + out_off + (uint32_t)chn * mch;
		store16_wt(o + p0, float4_t{O[0][3].x, O[0][2].x, O[1][3].x, O[1][2].x});
		store16_wt(o + p1, float4_t{O[1][1].x, O[1][0].x, O[0][1].x, O[0][0].x});
		store16_wt(o + p2, float4_t{O[0][0].y, O[0][1].y, O[1][0].y, O[1][1].y});
		store16_wt(o + p3, float4_t{O[1][2].y, O[1][3].y, O[0][2].y, O[0][3].y});
	} else {
		// samples.rs:92-103: x*32768, truncate toward zero (v_cvt_i32_f32: saturating, NaN -> 0), clamp to
		// i16 by the saturating pack v_cvt_pk_i16_i32 -- equal to the reference's compare/clamp/`as i16`
		int iq[2][4], im[2][4];
#pragma unroll
		for (int c2 = 0; c2 < 2; c2++)
#pragma unroll
			for (int k = 0; k < 4; k++) {
				iq[c2][k] = (int)O[c2][k].x;
				im[c2][k] = (int)O[c2][k].y;
			}
		if (FMT == LW_OUT_I16_PLANAR) {
			typedef short short2_t __attribute__((ext_vector_type(2)));
			union {
				short2_t s;
				uint32_t u;
			} a, b;
			int16_t *o = reinterpret_cast<int16_t *>(F.out) + out_off + (uint32_t)chn * mch;
			{
			a.s = __builtin_amdgcn_cvt_pk_i16(iq[0][3], iq[0][2]);
			b.s = __builtin_amdgcn_cvt_pk_i16(iq[1][3], iq[1][2]);
			store_pcm8(o + p0, a.u, b.u);
			a.s = __builtin_amdgcn_cvt_pk_i16(iq[1][1], iq[1][0]);
			b.s = __builtin_amdgcn_cvt_pk_i16(iq[0][1], iq[0][0]);
			store_pcm8(o + p1, a.u, b.u);
			a.s = __builtin_amdgcn_cvt_pk_i16(im[0][0], im[0][1]);
			b.s = __builtin_amdgcn_cvt_pk_i16(im[1][0], im[1][1]);
			store_pcm8(o + p2, a.u, b.u);
			a.s = __builtin_amdgcn_cvt_pk_i16(im[1][2], im[1][3]);
			b.s = __builtin_amdgcn_cvt_pk_i16(im[0][2], im[0][3]);
			store_pcm8(o + p3, a.u, b.u);
			}
		} else {
			// any channel count: 2-byte stores at stride ch (the stereo case never gets here, see store_interleaved2)
			typedef short short2_t __attribute__((ext_vector_type(2)));
			int16_t *o = reinterpret_cast<int16_t *>(F.out) + out_off + (uint32_t)chn;
			const uint32_t pos[4] = {p0, p1, p2, p3};
			const int lo[4][2] = {{iq[0][3], iq[1][3]}, {iq[1][1], iq[0][1]}, {im[0][0], im[1][0]}, {im[1][2], im[0][2]}};
			const int hi[4][2] = {{iq[0][2], iq[1][2]}, {iq[1][0], iq[0][0]}, {im[0][1], im[1][1]}, {im[1][3], im[0][3]}};
#pragma unroll
			for (int g = 0; g < 4; g++) {
				uint32_t off = pos[g] * F.ch; // one running offset per group keeps the address registers few
				asm volatile("" : "+v"(off));
#pragma unroll
				for (int h2 = 0; h2 < 2; h2++) {
					const short2_t pk = __builtin_amdgcn_cvt_pk_i16(lo[g][h2], hi[g][h2]); // saturating = the reference's clamp
					o[off] = pk.x;
					off += F.ch;
					o[off] = pk.y;
					off += F.ch;
				}
			}
		}
	}
}

// ---- raw left half of one channel to a td block: q ascending, then mirrored with the sign flipped (imdct.rs:589-658)
#define LW_PA_LO0(c) float4_t{R[c][0][3].x, R[c][0][2].x, R[c][1][3].x, R[c][1][2].x}
#define LW_PA_LO1(c) float4_t{R[c][1][1].x, R[c][1][0].x, R[c][0][1].x, R[c][0][0].x}
__device__ __forceinline__ void store16_wt(void *p, float4_t v);
__device__ __forceinline__ void store_left_half(float *dst, uint32_t lane, float4_t lo0, float4_t lo1)
{
	const float4_t hi0 = float4_t{-lo1.w, -lo1.z, -lo1.y, -lo1.x}; // 1023-q for q = 511-4l .. 508-4l
	const float4_t hi1 = float4_t{-lo0.w, -lo0.z, -lo0.y, -lo0.x};
	store16_wt(dst + 4u * lane, lo0);
	store16_wt(dst + 508u - 4u * lane, lo1);
	store16_wt(dst + 512u + 4u * lane, hi0);
	store16_wt(dst + 1020u - 4u * lane, hi1);
}

// ---- raw right half of one channel to a [1024]-float block (state slot / td block): q ascending, then mirrored
__device__ __forceinline__ void store_right_half(float *dst, uint32_t lane, float4_t lo0, float4_t lo1)
{
	const float4_t hi0 = float4_t{lo1.w, lo1.z, lo1.y, lo1.x}; // 1023-q for q = 511-4l .. 508-4l
	const float4_t hi1 = float4_t{lo0.w, lo0.z, lo0.y, lo0.x};
	store16_wt(dst + 4u * lane, lo0);
	store16_wt(dst + 508u - 4u * lane, lo1);
	store16_wt(dst + 512u + 4u * lane, hi0);
	store16_wt(dst + 1020u - 4u * lane, hi1);
}

// ---- four consecutive un-windowed samples of one channel (audio.rs:1119: positions past the overlap are copied), converted
//      and stored at sample index `rel` of the packet's output block (EDGE kernels: long blocks next to short ones)
template <int FMT>
__device__ __forceinline__ void store_quad(const LwFastArgs &F, int chn, uint32_t out_off, uint32_t mch, uint32_t rel, float4_t v, bool valid)
{
	if (!valid)
		return;
	if (FMT == LW_OUT_F32_PLANAR) {
		store16_wt(reinterpret_cast<float *>(F.out) + out_off + (uint32_t)chn * mch + rel, v);
		return;
	}
	typedef short short2_t __attribute__((ext_vector_type(2)));
	const float2_t k = float2_t{32768.0f, 32768.0f};
	const float2_t lo = pk_mul(float2_t{v.x, v.y}, k), hi = pk_mul(float2_t{v.z, v.w}, k); // samples.rs:92-103
	union {
		short2_t s;
		uint32_t u;
	} a, b;
	a.s = __builtin_amdgcn_cvt_pk_i16((int)lo.x, (int)lo.y);
	b.s = __builtin_amdgcn_cvt_pk_i16((int)hi.x, (int)hi.y);
	int16_t *o = reinterpret_cast<int16_t *>(F.out) + out_off;
	if (FMT == LW_OUT_I16_PLANAR) {
		store_pcm8(o + (uint32_t)chn * mch + rel, a.u, b.u);
	} else {
		uint32_t off = rel * F.ch + (uint32_t)chn;
		o[off] = a.s.x;
		o[off + F.ch] = a.s.y;
		o[off + 2u * F.ch] = b.s.x;
		o[off + 3u * F.ch] = b.s.y;
	}
}

// One work item = one 32-byte scalar load (the vector-memory path would park it in eight VGPRs per item).  In registers
// every field is a full dword: byte-sized struct members make hipcc copy the item byte by byte when it is carried from
// one round to the next.
typedef uint32_t u32x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ ItemRegs load_item(const LwFastItem *items, uint32_t idx)
{
	const LwFastItem *p = items + __builtin_amdgcn_readfirstlane(idx);
	u32x8_t v;
	asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
	ItemRegs it;
	it.res_off = v[0];
	it.floor_off = v[1];
	it.out_off = v[2];
	it.src_arg = v[3];
	it.state_out = (int32_t)v[4];
	it.halo_out = v[5];
	it.src_kind = v[6] & 0xffu;
	it.flags = (v[6] >> 16) & 0xffu;
	it.pkt = v[7];
	return it;
}

// item k of a dense list: packet k of a batch whose packets all have the same block sizes (no item load needed
// before the HBM loads can be issued)
__device__ __forceinline__ void dense_offsets(const LwFastArgs &F, uint32_t item, ItemRegs &it)
{
	it.res_off = item * F.ch * 1024u;
	it.floor_off = item * F.ch * F.fstride;
}

// LDS counters of the hand-over protocol (one producer, one consumer per counter), addressed by their LDS byte offset
// (the kernel has no static LDS, so the dynamic segment starts at 0).  Written as ds_ instructions: through a generic
// `volatile` pointer hipcc emits flat loads with system-scope cache bits and waits for every outstanding HBM access.
#define LW_CNT_BASE (LWI_TOTAL + LW_FAST_WAVES * (LW_SCR_BYTES + LW_PUB_BYTES))
#define LW_CNT_PUB(w) (LW_CNT_BASE + 4u * (w))
#define LW_CNT_ACK(w) (LW_CNT_BASE + 4u * (LW_FAST_WAVES + (w)))
#define LW_CNT_LANDED(w) (LW_CNT_BASE + 4u * (2 * LW_FAST_WAVES + (w)))

__device__ __forceinline__ uint32_t lds_load_u32(uint32_t byte_addr)
{
	uint32_t v;
	asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(byte_addr) : "memory");
	return v;
}

__device__ __forceinline__ void lds_store_u32(uint32_t byte_addr, uint32_t v)
{
	asm volatile("ds_write_b32 %0, %1" ::"v"(byte_addr), "v"(v) : "memory");
}

__device__ __forceinline__ void lds_wait_ge(uint32_t byte_addr, uint32_t need)
{
	while (__builtin_amdgcn_readfirstlane(lds_load_u32(byte_addr)) < need)
		__builtin_amdgcn_s_sleep(1);
}

// TD: the batch contains LW_IF_TDONLY items (long blocks next to short ones): a separate instantiation, so that the
// all-(1,1) batches keep the kernel without that branch (its presence alone cost 2 % of the headline launch time)
// EDGE: the batch's short blocks run through k_short (lw_fast.hpp): long blocks with a short slope on either side
// (LW_IF_EDGE_L / LW_IF_EDGE_R) do everything but the 128-sample overlap with the short neighbour here -- the samples past a
// short slope are copied un-windowed (audio.rs:1119), the raw edges pa(448..511) / pb(448..511) go to the edge buffer.
// SPLIT: sparse launches run one channel per wave (LW_UNIT_SPLIT_*, lw_fast.hpp); a separate instantiation, so that the dense
// launches keep their register allocation.
template <int FMT, bool RIGHT_ONLY, bool TD = false, bool EDGE = false, bool SPLIT = false, bool PRE = false>
__global__ void __launch_bounds__(LW_WG) k_long(LwFastArgs F)
{
	constexpr bool MIXF = false, mix_drop_flags = false;
	uint32_t *const edge_flags = nullptr;
#include "lw_long_body.inc"
}

// ---------------------------------------------------------------------------------------------
// k_short<L>: blocks of 32 L = 256 / 512 / 1024 points -- the short blocks of a stream, and the long blocks k_long does not cover
// ---------------------------------------------------------------------------------------------
// One wave64 = one workgroup = 64 / L "slots" (L lanes each) x one unit (a coupled channel pair, or one channel): blocks from
// their entropy records to PCM, no barrier, no communication with other waves.  The transform is k_long's, cut down to 8 L
// complex pairs per block (layouts B' / C' / D' / E' of lw_fast.hpp; tests/short_model.py is the executable specification):
// step 1 on the coalesced load layout, exchange with the mirror lane of the L-lane group, step 2 and stages l = 0, 1
// register-local, (L >= 16) one or two register <-> lane exchanges and stages l = 2 (, 3), ONE 8 x 8 register <-> lane
// transpose (t3_inreg), the fused last three stages register-local, the bit-reverse gather through LDS, steps 7 and 8,
// window / overlap-add, conversion, stores.  A slot's previous right part comes from the previous slot of the same wave
// (through LDS: consecutive blocks of a stream sit in consecutive slots), from the stream's state slot, from the edge
// buffer k_long fills for long blocks with a short right slope, or from the time-domain block of a predecessor the generic
// kernels (or k_long<TD>) transformed.  With 256-point short blocks next to k_long, a slot whose successor is a long block with
// a short LEFT slope also does that block's first 128 samples (its raw left edge pa(448..511) comes from the edge buffer;
// k_long runs before this kernel), so neither kernel waits for the other inside a launch and no time-domain block makes a
// round trip through HBM.
// Floor curve: as in k_long (interval entries {dy, 0.5 sgn(dy) - x0 dy, 1/adx, 4 y0} per post, y(k) = y0 + trunc((k dy +
// c0) / adx) -- tests/test_fast_model.py), with the L lanes of a slot building the entries of each channel: the active-post
// mask of a slot is L bits of a wave-wide ballot per group of L posts.
#define LW_SHORT_SPLIT_BELOW 1024u // waves (tasks x units) below which a launch splits channel pairs over two waves

template <int L>
struct LwBlkLds { // compile-time facts of the LDS layout: [table image][gather][right parts][records][interval entries]
	static constexpr uint32_t SLOTS = 64u / L;
	static constexpr uint32_t POSTS = LW_BLK_MAX_POSTS(L);          // most posts per channel of a block
	static constexpr uint32_t PT = POSTS / L;                      // posts per lane
};

// LDS of a wave behind the image.  Every area sits at a compile-time offset (ds offsets fold into the instructions: sizing the
// floor areas for the stream's real post count at run time saved 3 KB per wave and cost 8 % at 16 384 packets per launch).
// ONE: the areas of one channel and one pass only (k_mix: the waves finish one channel each, 13 of them share a workgroup's LDS)
template <int L, bool ONE = false>
struct LwBlkWave {
	static constexpr uint32_t SLOTS = 64u / L, POSTS = LW_BLK_MAX_POSTS(L), CH = ONE ? 1u : 2u;
	static constexpr uint32_t SCR = 0;                              // bit-reverse gather of one channel: [slot][8 L] pairs
	static constexpr uint32_t PUB = 4096u;                          // right parts [pass parity][slot][channel][c2][l] float4
	static constexpr uint32_t PUB_PASS = SLOTS * CH * 2u * L * 16u; // ... of one pass (4 KB; ONE: 2 KB)
	static constexpr uint32_t REC = PUB + (ONE ? 1u : 2u) * PUB_PASS; // floor records [slot][channel][POSTS] u16
	static constexpr uint32_t TAB = REC + SLOTS * CH * POSTS * 2u;  // interval entries [slot][channel][POSTS] x 16 bytes
	static constexpr uint32_t BYTES = TAB + SLOTS * CH * POSTS * 16u;
};
static_assert(LwBlkWave<8>::PUB_PASS == 4096u && LwBlkWave<32>::PUB_PASS == 4096u, "right parts of one pass: 4 KB");

struct LwShortArgs {
	const float *residue;
	const uint16_t *floors;
	const LwShortSlot *slots;
	const uint8_t *image;
	float *state, *td, *edge;
	void *out;
	uint32_t n_units, ch, fstride, state_stride, state_chan_stride;
	uint32_t n_waves;  // tasks x units = workgroups of one wave
	uint32_t passes; // a wave works through `passes` x 64 / L consecutive slots, 64 / L at a time (the last slot of a pass hands its
	                 // right part to the first slot of the next through LDS: a recomputed predecessor only at the start of a wave)
	LwFastUnit units[LW_FAST_WAVES];
};

// the 4 + 4 values a lane holds of a half block q = 0 .. 8L-1 (pa or pb): lo = q in [4l, 4l+4), hi = q in [8L-4-4l, 8L-4l)
struct Half8 {
	float4_t lo, hi;
};

template <int L>
__device__ __forceinline__ Half8 load_half8(const float *src, uint32_t l)
{
	Half8 h;
	h.lo = *reinterpret_cast<const float4_t *>(src + 4u * l);
	h.hi = *reinterpret_cast<const float4_t *>(src + (8u * L - 4u) - 4u * l);
	return h;
}

// previous right part as the overlap-add consumes it (see prev_from_global)
__device__ __forceinline__ void prev_from_half8(const Half8 &g, PrevHalf &h)
{
	h.pp[0][1] = float2_t{g.lo.y, g.lo.x};
	h.pp[1][1] = float2_t{g.lo.w, g.lo.z};
	h.pp[1][0] = float2_t{g.hi.y, g.hi.x};
	h.pp[0][0] = float2_t{g.hi.w, g.hi.z};
}

// interval entries of one channel of this lane's slot: posts i = l + L t (see floor_table for the entry)
template <int L>
__device__ __forceinline__ bool short_floor_table(const char *img, char *rec, char *tab, uint32_t g, uint32_t l,
		const uint32_t (&e)[LwBlkLds<L>::PT], uint32_t fslot, uint32_t Fp, bool has_floor)
{
	constexpr int PT = LwBlkLds<L>::PT;
	unsigned long long mask = 0;
#pragma unroll
	for (int t = 0; t < PT; t++) {
		const unsigned long long bal = __ballot(has_floor && (e[t] & LW_POST_ACTIVE) != 0);
		mask |= ((bal >> (L * g)) & ((1ull << L) - 1ull)) << (L * t);
		const uint32_t i = l + L * (uint32_t)t;
		if (i < Fp)
			*reinterpret_cast<uint16_t *>(rec + 2u * i) = (uint16_t)e[t];
	}
	const unsigned long long ub = __ballot(l == 0 && e[0] == LW_FLOOR_UNUSED);
	const bool unused = !has_floor || ((ub >> (L * g)) & 1ull) != 0;
	lds_fence();
#pragma unroll
	for (int t = 0; t < PT; t++) {
		const uint32_t i = l + L * (uint32_t)t;
		const unsigned long long lowmask = (2ull << i) - 1ull; // (i = 63: 2 << 63 wraps to 0, the mask becomes all ones)
		const unsigned long long below = mask & lowmask, above = mask & ~lowmask;
		const int lo = below ? 63 - __builtin_clzll(below) : 0;
		const int hi = above ? __builtin_ctzll(above) : lo;
		const int ylo = (int)(*reinterpret_cast<const uint16_t *>(rec + 2 * lo) & 0xffu);
		const int yhi = (int)(*reinterpret_cast<const uint16_t *>(rec + 2 * hi) & 0xffu);
		const float xlo = *reinterpret_cast<const float *>(img + LwBlkLayout<L>::XSF + 4u * (64u * fslot + (uint32_t)lo));
		const float xhi = *reinterpret_cast<const float *>(img + LwBlkLayout<L>::XSF + 4u * (64u * fslot + (uint32_t)hi));
		const float dy = (float)(yhi - ylo); // 0 when there is no later active post (flat, audio.rs:546-548)
		float4_t ent;
		ent.x = dy;
		ent.y = __builtin_copysignf(0.5f, dy) - xlo * dy; // exact
		ent.z = above ? __builtin_amdgcn_rcpf(xhi - xlo) : 1.0f;
		ent.w = __int_as_float(ylo << 2);
		if (i < Fp)
			*reinterpret_cast<float4_t *>(tab + 16u * i) = ent;
	}
	return unused;
}

// floor x residue of one channel of this lane's slot, in place (audio.rs:1035-1037); bins 4 (L x + l) + j
template <int L>
__device__ __forceinline__ void short_spectrum(const char *img, const char *tab, uint32_t l, uint32_t fslot, bool unused, float4_t (&r)[4])
{
	const float kf0 = (float)(4 * (int)l);
#pragma unroll
	for (int x = 0; x < 4; x++) {
		const uint2_t sw = *reinterpret_cast<const uint2_t *>(img + LwBlkLayout<L>::SID16 + 8u * ((fslot * 4 + x) * L + l));
		const uint32_t s16[4] = {sw.x & 0xffffu, sw.x >> 16, sw.y & 0xffffu, sw.y >> 16};
		float4_t fl;
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const float4_t ent = lds4(tab, s16[j]);
			const float z = __builtin_fmaf(kf0 + (float)(4 * L * x + j), ent.x, ent.y); // exact: |k*dy| < 2^17
			const int q = (int)(z * ent.z);
			const uint32_t idx = (uint32_t)((q << 2) + __float_as_int(ent.w));
			fl[j] = *reinterpret_cast<const float *>(img + LwBlkLayout<L>::INV_DB + idx);
		}
		if (unused)
			fl = float4_t{0.0f, 0.0f, 0.0f, 0.0f}; // zero floor (audio.rs:1021-1024): 0.0 x residue keeps the residue's sign
		const float2_t lo2 = pk_mul(float2_t{fl.x, fl.y}, float2_t{r[x].x, r[x].y});
		const float2_t hi2 = pk_mul(float2_t{fl.z, fl.w}, float2_t{r[x].z, r[x].w});
		r[x] = float4_t{lo2.x, lo2.y, hi2.x, hi2.y};
	}
}

// exchange of register-index bits with lane bits 4 / 3 for the eight pairs of a lane (the last one / two steps of t2_inreg)
template <int L>
__device__ __forceinline__ void blk_t2(float2_t (&P)[8])
{
	if (L == 32) {
#pragma unroll
		for (int i = 0; i < 4; i++) { // register bit 1 <-> lane bit 4
			const int x = (i & 1) | ((i & 2) << 1);
			auto r0 = __builtin_amdgcn_permlane16_swap(__float_as_uint(P[x].x), __float_as_uint(P[x + 2].x), false, false);
			P[x].x = __uint_as_float(r0[0]);
			P[x + 2].x = __uint_as_float(r0[1]);
			auto r1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(P[x].y), __float_as_uint(P[x + 2].y), false, false);
			P[x].y = __uint_as_float(r1[0]);
			P[x + 2].y = __uint_as_float(r1[1]);
		}
	}
	if (L >= 16) {
#pragma unroll
		for (int x = 0; x < 8; x += 2) { // register bit 0 <-> lane bit 3 (tied DPP moves under bank masks)
#pragma unroll
			for (int k = 0; k < 2; k++) {
				const int a_ = __float_as_int(k ? P[x].y : P[x].x), b_ = __float_as_int(k ? P[x + 1].y : P[x + 1].x);
				const float na = __int_as_float(__builtin_amdgcn_update_dpp(a_, b_, 0x128, 0xf, 0xc, false)); // lanes 8-15 of a row <- b[l ^ 8]
				const float nb = __int_as_float(__builtin_amdgcn_update_dpp(b_, a_, 0x128, 0xf, 0x3, false)); // lanes 0-7  of a row <- a[l ^ 8]
				if (k) {
					P[x].y = na;
					P[x + 1].y = nb;
				} else {
					P[x].x = na;
					P[x + 1].x = nb;
				}
			}
		}
	}
}

// Where pair p of block g lies in a wave's bit-reverse gather area (8-byte slots; a bijection of (g, p) onto 0 .. 511, linear over
// GF(2): slot(g, p ^ q) = slot(g, p) ^ slot(0, q)).  The gather is written in layout D' (ds_write_b64: a wave's lanes are served in four
// groups of 16 on 32 banks; a group's lanes differ in the pair bits above the register's three) and read at the pairs 2v, 2v + P/2,
// P/2 - 1 - 2v, P - 1 - 2v, v = bit-reversed m' (ds_read_b64: two groups of 32 lanes on 64 banks).  In the plain order slot = P g + p
// a group's writes fall on two to four bank pairs (8-way conflicts) and its reads on half of the banks (2- to 4-way): 36-49 % of
// the block kernel's LDS cycles were bank conflicts (profiles/r03_pmc_mixed.json).  These maps put the bits that vary inside a
// write group into the slot's low four bits and those that vary inside a read group into its low five, xor-ing in the bits
// the other access holds fixed: no conflict on either side (tests/test_short_model.py simulates the banks).
template <int L>
__host__ __device__ constexpr uint32_t blk_slot(uint32_t g, uint32_t p)
{
#define LW_PB(k) ((p >> (k)) & 1u)
	return L == 8 ? (LW_PB(3) | (g & 1u) << 1 | (LW_PB(4) ^ LW_PB(1)) << 2 | (LW_PB(5) ^ LW_PB(2)) << 3 | ((g >> 1) & 1u) << 4 | LW_PB(1) << 5 |
	                 LW_PB(2) << 6 | LW_PB(0) << 7 | ((g >> 2) & 1u) << 8)
	     : L == 16 ? (LW_PB(3) | LW_PB(4) << 1 | (LW_PB(5) ^ LW_PB(1)) << 2 | (LW_PB(6) ^ LW_PB(2)) << 3 | (g & 1u) << 4 | LW_PB(1) << 5 |
	                  LW_PB(2) << 6 | LW_PB(0) << 7 | ((g >> 1) & 1u) << 8)
	               : (LW_PB(3) | LW_PB(4) << 1 | LW_PB(5) << 2 | (LW_PB(1) ^ LW_PB(7)) << 3 | LW_PB(2) << 4 | LW_PB(0) << 5 | LW_PB(6) << 6 |
	                  LW_PB(7) << 7 | (g & 1u) << 8);
#undef LW_PB
}

// the transform of one channel of the wave's blocks: spectrum r (load layout) -> R[c2][k] = (pa, pb) at
// q = 8L-1 - 2m', 8L-2 - 2m', 1 + 2m', 2m' for m' = 2 l + c2 (imdct.rs:291-659)
template <int L>
__device__ __forceinline__ void short_imdct(const char *img, char *scr, uint32_t g, uint32_t l, const float4_t (&r)[4], float2_t (&R)[2][4])
{
	typedef LwBlkLayout<L> Y;
	constexpr uint32_t P = 8u * L;
	float2_t au[4], al[4], s2[4], l0[2], l1;
#pragma unroll
	for (int x = 0; x < 4; x++) {
		const uint32_t m = L * x + l;
		au[x] = lds2(img + Y::APAIR, 8u * m);           // (A[2m], A[2m+1])
		al[x] = lds2(img + Y::APAIR, 8u * (P - 1u - m)); // (A[n/2-2-2m], A[n/2-1-2m])
		s2[x] = lds2(img + Y::TW_S2, 8u * m);
	}
	l0[0] = lds2(img + Y::TW_L0, 8u * l);
	l0[1] = lds2(img + Y::TW_L0, 8u * (L + l));
	l1 = lds2(img + Y::TW_L1, 8u * l);
	float2_t Q[8], U[4];
	step1x2(r[0], au[0], al[0], r[1], au[1], al[1], U[0], Q[0], U[1], Q[1]);
	step1x2(r[2], au[2], al[2], r[3], au[3], al[3], U[2], Q[2], U[3], Q[3]);
#pragma unroll
	for (int x = 0; x < 4; x++) { // pair 8L-1 - m lives on the mirror lane of the L-lane group
		if (L == 32) {
			const uint32_t mirror = (threadIdx.x ^ 31u) << 2;
			Q[7 - x].x = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(U[x].x)));
			Q[7 - x].y = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(U[x].y)));
		} else { // DPP row_half_mirror (8 lanes) / row_mirror (16 lanes)
			Q[7 - x].x = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(U[x].x), L == 8 ? 0x141 : 0x140, 0xf, 0xf, false));
			Q[7 - x].y = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(U[x].y), L == 8 ? 0x141 : 0x140, 0xf, 0xf, false));
		}
	}
	bfly2x4(Q[4], Q[0], s2[0], Q[5], Q[1], s2[1], Q[6], Q[2], s2[2], Q[7], Q[3], s2[3]); // step 2 (imdct.rs:385-430)
	bfly2x4(Q[2], Q[0], l0[0], Q[6], Q[4], l0[0], Q[3], Q[1], l0[1], Q[7], Q[5], l0[1]); // l = 0
	bfly2x4(Q[1], Q[0], l1, Q[3], Q[2], l1, Q[5], Q[4], l1, Q[7], Q[6], l1);             // l = 1
	if (L >= 16) { // stages l = 2 (, 3) on pair bits that were lane bits 4 / 3 (imdct.rs:454-477)
		blk_t2<L>(Q);
		const uint32_t lo3 = l & 7u;
		if (L == 32) {
			const float2_t c2a = lds2(img + Y::TW_C2, 8u * lo3), c2b = lds2(img + Y::TW_C2, 8u * (8u + lo3));
			const float2_t c3 = lds2(img + Y::TW_C3, 8u * lo3);
			bfly2x4(Q[2], Q[0], c2a, Q[6], Q[4], c2a, Q[3], Q[1], c2b, Q[7], Q[5], c2b); // l = 2
			bfly2x4(Q[1], Q[0], c3, Q[3], Q[2], c3, Q[5], Q[4], c3, Q[7], Q[6], c3);     // l = 3
		} else {
			const float2_t c2 = lds2(img + Y::TW_C2, 8u * lo3);
			bfly2x4(Q[1], Q[0], c2, Q[3], Q[2], c2, Q[5], Q[4], c2, Q[7], Q[6], c2);     // l = 2
		}
	}
	t3_inreg(Q);                                                                         // -> D': register = p[2:0]
	const float a2s = *reinterpret_cast<const float *>(img + Y::A2);
	stage_d_block(float2_t{a2s, a2s}, Q);                                                // imdct.rs:234-288
	// pair index bits above the register's three, from the lane bits (the exchanges above permuted them)
	uint32_t hi;
	if (L == 8)
		hi = l;                                                                          // (p5, p4, p3)
	else if (L == 16)
		hi = ((l >> 2) & 1u) << 3 | ((l >> 1) & 1u) << 2 | ((l >> 3) & 1u) << 1 | (l & 1u); // (p6, p5, p4, p3) = lane bits (2, 1, 3, 0)
	else
		hi = ((l >> 2) & 1u) << 4 | ((l >> 4) & 1u) << 3 | ((l >> 3) & 1u) << 2 | ((l >> 1) & 1u) << 1 | (l & 1u); // (p7 .. p3) = lane bits (2, 4, 3, 1, 0)
	// (slots through blk_slot: the pair index is lane bits xor constants, and the map is linear, so every address is one of two lane
	// terms xor a compile-time constant)
	const uint32_t wbase = 8u * blk_slot<L>(g, 8u * hi);
#pragma unroll
	for (int zz = 0; zz < 8; zz++)
		*reinterpret_cast<float2_t *>(scr + (wbase ^ (8u * blk_slot<L>(0, zz)))) = Q[zz];
	lds_fence();
	constexpr int VB = (L == 8 ? 4 : L == 16 ? 5 : 6); // bits of m' = 2 l + c2
	const uint32_t rbase = 8u * blk_slot<L>(g, 2u * (__builtin_bitreverse32(2u * l) >> (32 - VB))); // pair 2v of c2 = 0
#pragma unroll
	for (int c2 = 0; c2 < 2; c2++) { // bit-reverse gather (imdct.rs:490-528), step 7 (:533-580), step 8 (:589-658)
		// v = rev(2 l + c2) = rev(2 l) ^ c2 << (VB - 1); the pairs 2v, 2v + P/2, P/2 - 1 - 2v, P - 1 - 2v = 2v ^ 0, P/2, P/2 - 1, P - 1
		const uint32_t kc = (uint32_t)c2 << VB;
		const float2_t pq = lds2(scr, rbase ^ (8u * blk_slot<L>(0, kc))), pqh = lds2(scr, rbase ^ (8u * blk_slot<L>(0, kc ^ (P / 2))));
		const float2_t ph = lds2(scr, rbase ^ (8u * blk_slot<L>(0, kc ^ (P / 2 - 1u)))), pf = lds2(scr, rbase ^ (8u * blk_slot<L>(0, kc ^ (P - 1u))));
		const float4_t Cq = lds4(img + Y::C4, 16u * (L * c2 + l));
		const float4_t Bl = lds4(img + Y::B_LO, 16u * (L * c2 + l)), Bh = lds4(img + Y::B_HI, 16u * (L * c2 + l));
		step78_block(pf, pq, ph, pqh, Cq, Bl, Bh, R[c2]);
	}
	lds_fence();
}

// window + overlap-add (audio.rs:1116-1118) of the 16 L samples of one channel of this lane's slot, conversion
// (samples.rs:92-103), stores.  R[c2][k].x = raw left half at q_k, h = the previous right part; `o` = element 0 of the
// channel's samples; stride = distance of consecutive samples (1: planar, ch: interleaved)
template <int FMT, int L>
__device__ __forceinline__ void short_ola_store(const char *img, uint32_t l, void *out, uint32_t elem0, uint32_t stride,
		const float2_t (&Rc)[2][4], const PrevHalf &h)
{
	float2_t O[2][4];
#pragma unroll
	for (int c2 = 0; c2 < 2; c2++) {
		const float4_t w0 = lds4(img + LwBlkLayout<L>::WIN, 32u * (L * c2 + l));
		const float4_t w1 = lds4(img + LwBlkLayout<L>::WIN, 32u * (L * c2 + l) + 16u);
		ola_block<FMT != LW_OUT_F32_PLANAR>(Rc[c2], h.pp[c2][0], h.pp[c2][1], w0, w1, O[c2]);
	}
	// positions: [4l..4l+3] = .x of (0,3) (0,2) (1,3) (1,2); [8L-4-4l..] = .x of (1,1) (1,0) (0,1) (0,0)
	//            [8L+4l..] = .y of (0,0) (0,1) (1,0) (1,1); [16L-4-4l..] = .y of (1,2) (1,3) (0,2) (0,3)
	const uint32_t pos[4] = {4u * l, 8u * L - 4u - 4u * l, 8u * L + 4u * l, 16u * L - 4u - 4u * l};
	const float v[4][4] = {{O[0][3].x, O[0][2].x, O[1][3].x, O[1][2].x}, {O[1][1].x, O[1][0].x, O[0][1].x, O[0][0].x},
		{O[0][0].y, O[0][1].y, O[1][0].y, O[1][1].y}, {O[1][2].y, O[1][3].y, O[0][2].y, O[0][3].y}};
#pragma unroll
	for (int q = 0; q < 4; q++) {
		if (FMT == LW_OUT_F32_PLANAR) {
			store16_wt(reinterpret_cast<float *>(out) + elem0 + pos[q], float4_t{v[q][0], v[q][1], v[q][2], v[q][3]});
		} else {
			// samples.rs:92-103: x * 32768 (done in ola_block), truncate toward zero (v_cvt_i32_f32: saturating, NaN -> 0), clamp
			// to i16 by the saturating pack
			typedef short short2_t __attribute__((ext_vector_type(2)));
			union {
				short2_t s;
				uint32_t u;
			} a, b;
			a.s = __builtin_amdgcn_cvt_pk_i16((int)v[q][0], (int)v[q][1]);
			b.s = __builtin_amdgcn_cvt_pk_i16((int)v[q][2], (int)v[q][3]);
			int16_t *o = reinterpret_cast<int16_t *>(out) + elem0;
			if (FMT == LW_OUT_I16_PLANAR) {
				store_pcm8(o + pos[q], a.u, b.u);
			} else {
				uint32_t off = pos[q] * stride;
				o[off] = a.s.x;
				o[off + stride] = a.s.y;
				o[off + 2u * stride] = b.s.x;
				o[off + 3u * stride] = b.s.y;
			}
		}
	}
}

// raw right part (16 L floats: pb(q) for q ascending, then mirrored) of one channel to a state slot / td block
template <int L>
__device__ __forceinline__ void short_store_right(float *dst, uint32_t l, const float2_t (&Rc)[2][4])
{
	const float4_t lo0 = float4_t{Rc[0][3].y, Rc[0][2].y, Rc[1][3].y, Rc[1][2].y}; // q = 4l ..
	const float4_t lo1 = float4_t{Rc[1][1].y, Rc[1][0].y, Rc[0][1].y, Rc[0][0].y}; // q = 8L - 4 - 4l ..
	store16_wt(dst + 4u * l, lo0);
	store16_wt(dst + 8u * L - 4u - 4u * l, lo1);
	store16_wt(dst + 8u * L + 4u * l, float4_t{lo1.w, lo1.z, lo1.y, lo1.x});
	store16_wt(dst + 16u * L - 4u - 4u * l, float4_t{lo0.w, lo0.z, lo0.y, lo0.x});
}

// 16-byte load that sees what another workgroup's wave has written through (sc1) in this very launch (k_mix: the raw edges)
__device__ __forceinline__ Half8 load_half8_coherent(const float *src, uint32_t l, uint32_t L8)
{
	Half8 h;
	asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
	             : "=&v"(h.lo), "=&v"(h.hi) : "v"(src + 4u * l), "v"(src + (L8 - 4u) - 4u * l) : "memory");
	return h;
}

// the batch's device error word lives in host memory (lw_batch_device_status reads it once the launch has completed)
__device__ __forceinline__ void raise_device_error(uint32_t *err)
{
	__hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ uint32_t load_flag_coherent(const uint32_t *p)
{
	uint32_t v;
	asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
	return v;
}

template <int FMT, int L>
__global__ void __launch_bounds__(64) k_short(LwShortArgs F)
{
	__shared__ __attribute__((aligned(16))) char smem_all[LwBlkLayout<L>::TOTAL + LwBlkWave<L>::BYTES];
	char *const smem = smem_all + LwBlkLayout<L>::TOTAL; // the wave's working areas
	const uint32_t wid = blockIdx.x, lane = threadIdx.x;     // this wave's (task, unit)
	constexpr int ONLY = -1;
	constexpr bool STAGE = true;
	uint32_t *const edge_flags = nullptr, *const mix_err = nullptr;
	constexpr uint32_t mix_spin = 0;
#include "lw_short_body.inc"
}

// ---------------------------------------------------------------------------------------------
// k_mix: a mixed short / long batch in ONE launch
// ---------------------------------------------------------------------------------------------
// BASELINE configs[2] at 4096 packets is two sparse launches -- k_long<EDGE, SPLIT> for the long blocks, k_short<8> behind it for the
// short ones -- and each lasts about as long as ONE wave's dependent chain plus 3 us of start-up: 8.3 + 7.2 us for 18 MB.  The
// short blocks need two things from the long ones, the raw edges either side of a run of short blocks; everything else they do
// (records, floor curves, transforms) is independent.  And the sparse long-block launch leaves waves idle: five packets per
// workgroup, one channel per wave, are ten of its sixteen.  k_mix is k_long's launch (the same grid, the same body, textually)
// in which the last LW_MIX_SHORT_WAVES waves of every workgroup, once the table image is staged, turn to k_short's work: wave
// 12 + i of workgroup g is k_short wave 4 g + i (one channel per wave, known at compile time: 108 registers; the compact areas of
// LwBlkWave<8, true> laid into the LDS of the idle waves 10..15, the short blocks' 5 KB image behind the workgroup's own
// areas).  A long block's wave sets a flag in HBM behind each raw edge it has written through; a short wave runs up to its
// overlap-add and only then waits for the flags of its two edges.  No deadlock: long-block waves wait for nobody, and every
// workgroup of the grid is resident (the planner never makes more workgroups than CUs for a sparse launch, one per CU).
static_assert(2u * LwBlkWave<8, true>::BYTES <= 6u * LW_SCR_BYTES && 2u * LwBlkWave<8, true>::BYTES <= 6u * LW_PUB_BYTES,
		"two short waves' areas fit the transpose buffers of the six idle waves, two more their hand-over buffers");
#define LW_MIX_IMG_OFF ((LW_LDS_BYTES + 15u) & ~15u) // the short blocks' image behind k_long's own LDS
#define LW_MIX_LDS_BYTES (LW_MIX_IMG_OFF + LwBlkLayout<8>::TOTAL)
static_assert(LW_MIX_LDS_BYTES <= 160u * 1024u, "k_long's areas and the short blocks' image fit a CU's LDS");

#define LW_MIX_SPIN (1u << 20) // polls of a short block's wave for its edge flags before it gives the batch up (~1 s)
struct LwMixArgs {
	uint32_t *flags;     // [packet][side][ch] of the batch, zero between launches (every flag is cleared by its one reader)
	uint32_t *err;       // the batch's device error word (host memory, mapped): set by a short block's wave whose flags never came
	uint32_t spin;       // polls before giving up (LW_MIX_SPIN; the test hook shortens it)
	uint32_t drop_flags; // test hook (lw_debug_batch_break_mix): the long blocks' waves never signal
};

template <int FMT>
__device__ __forceinline__ void mix_short_role(const LwShortArgs &F, const LwMixArgs &M, char *smem_dyn, uint32_t wave, uint32_t lane)
{
	uint32_t *const edge_flags = M.flags, *const mix_err = M.err;
	const uint32_t mix_spin = M.spin;
	constexpr int L = 8;
	const uint32_t i = wave - (LW_FAST_WAVES - LW_MIX_SHORT_WAVES);
	const uint32_t wid = blockIdx.x * LW_MIX_SHORT_WAVES + i; // this wave's (task, unit half)
	if (wid >= F.n_waves)
		return;
	char *const smem_all = smem_dyn + LW_MIX_IMG_OFF; // (staged by the workgroup's last four waves together with k_long's image)
	// waves 10..15 of a k_mix workgroup have no long-block work: their transpose buffers (24 KB in a row) and their hand-over
	// buffers (24 KB in a row) hold two short waves' areas each
	char *const smem = smem_dyn + LWI_TOTAL + (i < 2u ? 0u : LW_FAST_WAVES * LW_SCR_BYTES) + LW_MIX_LONG_WAVES * LW_SCR_BYTES +
		(i & 1u) * LwBlkWave<L, true>::BYTES;
	constexpr bool STAGE = false;
	if (__builtin_amdgcn_readfirstlane((uint32_t)F.units[wid % F.n_units].slot) == 0u) {
		constexpr int ONLY = 0;
#include "lw_short_body.inc"
	} else {
		constexpr int ONLY = 1;
#include "lw_short_body.inc"
	}
}

template <int FMT>
__global__ void __launch_bounds__(LW_WG) k_mix(LwFastArgs F, LwShortArgs FS, LwMixArgs M)
{
	constexpr bool RIGHT_ONLY = false, TD = false, EDGE = true, SPLIT = true, MIXF = true, PRE = false;
	uint32_t *const edge_flags = M.flags;
	const bool mix_drop_flags = M.drop_flags != 0;
#define LW_LONG_BODY_AFTER_STAGE                                                  \
	if (wave >= LW_FAST_WAVES - LW_MIX_SHORT_WAVES) {                            \
		mix_short_role<FMT>(FS, M, smem, wave, lane_id);                          \
		return;                                                                   \
	}
	// the short blocks' image (5 KB = 320 x 16 bytes) goes up with k_long's, by the threads of the last four waves
	static_assert(LwBlkLayout<8>::TOTAL / 16u <= 256u + 64u, "one 16-byte row per thread of four waves, a second one for the first 64");
#define LW_LONG_BODY_STAGE_LOAD                                                                       \
	const bool s_on = t >= 512u, s_two = s_on && t - 512u < LwBlkLayout<8>::TOTAL / 16u - 256u;       \
	const uint4 *s_src = reinterpret_cast<const uint4 *>(FS.image) + (s_on ? t - 512u : 0u);          \
	uint4 s_v0 = make_uint4(0, 0, 0, 0), s_v1 = s_v0;                                                 \
	if (s_on)                                                                                         \
		s_v0 = s_src[0];                                                                              \
	if (s_two)                                                                                        \
		s_v1 = s_src[256];
#define LW_LONG_BODY_STAGE_STORE                                                                      \
	uint4 *s_dst = reinterpret_cast<uint4 *>(smem + LW_MIX_IMG_OFF) + (s_on ? t - 512u : 0u);         \
	if (s_on)                                                                                         \
		s_dst[0] = s_v0;                                                                              \
	if (s_two)                                                                                        \
		s_dst[256] = s_v1;
#include "lw_long_body.inc"
#undef LW_LONG_BODY_AFTER_STAGE
#undef LW_LONG_BODY_STAGE_LOAD
#undef LW_LONG_BODY_STAGE_STORE
}

template <int L>
static hipError_t launch_short(LwShortArgs &F, int fmt, hipStream_t st)
{
	// one wave per workgroup with 17-30 KB of LDS (below the 64 KB that need no attribute): the waves a CU holds at a time
	// are bounded by LDS for L = 16 / 32 and by registers for L = 8
	const dim3 grid(F.n_waves), block(64);
	if (fmt == LW_OUT_I16_PLANAR)
		return lw_launch_k(k_short<LW_OUT_I16_PLANAR, L>, grid, block, 0, st, F);
	if (fmt == LW_OUT_I16_INTERLEAVED)
		return lw_launch_k(k_short<LW_OUT_I16_INTERLEAVED, L>, grid, block, 0, st, F);
	return lw_launch_k(k_short<LW_OUT_F32_PLANAR, L>, grid, block, 0, st, F);
}

// the kernel arguments of a k_short launch (also the short role of k_mix)
static void short_prepare(const LwDevTables &T, const LwBatchDev &B, const LwShortLaunch &L, void *out, LwShortArgs &F)
{
	F.residue = B.residue;
	F.floors = B.floors;
	F.slots = L.d_slots;
	F.image = L.d_image;
	F.state = B.state;
	F.td = B.td;
	F.edge = L.d_edge;
	F.out = out;
	F.ch = T.ch;
	F.fstride = T.fstride;
	F.state_stride = T.state_stride;
	F.state_chan_stride = T.state_chan_stride;
	F.passes = L.passes ? L.passes : 1u;
	// a launch of few waves splits channel pairs over two waves each (see the kernel)
	const bool split = (size_t)L.n_tasks * L.n_units <= LW_SHORT_SPLIT_BELOW && 2 * L.n_units <= LW_FAST_WAVES;
	uint32_t nu = 0;
	for (uint32_t u = 0; u < L.n_units && u < LW_FAST_WAVES; u++) {
		if (split && L.units[u].ch_b >= 0) {
			F.units[nu] = L.units[u];
			F.units[nu++].slot = 0;
			F.units[nu] = L.units[u];
			F.units[nu++].slot = 1;
		} else {
			F.units[nu] = L.units[u];
			F.units[nu++].slot = 0xFF;
		}
	}
	F.n_units = nu;
	F.n_waves = L.n_tasks * nu;
}

hipError_t lw_launch_short(const LwDevTables &T, const LwBatchDev &B, const LwShortLaunch &L, void *out, int fmt, hipStream_t st)
{
	if (L.n_tasks == 0)
		return hipSuccess;
	LwShortArgs F{};
	short_prepare(T, B, L, out, F);
	if (L.lanes == 8)
		return launch_short<8>(F, fmt, st);
	if (L.lanes == 16)
		return launch_short<16>(F, fmt, st);
	if (L.lanes == 32)
		return launch_short<32>(F, fmt, st);
	return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------------
// per-device attributes, the halo pre-pass, and the kernel arguments + grid of the main pass (also the long role of k_mix)
static hipError_t long_prepare(const LwDevTables &T, const LwBatchDev &B, const LwFastLaunch &L, void *out, hipStream_t st, LwFastArgs &F,
		uint32_t &grid)
{
	F.image = L.d_image;
	F.residue = B.residue;
	F.floors = B.floors;
	F.state = B.state;
	F.td = B.td;
	F.ch = T.ch;
	F.fstride = T.fstride;
	F.state_stride = T.state_stride;
	F.state_chan_stride = T.state_chan_stride;
	F.n_units = L.n_units;
	F.halo = L.d_halo;
	F.edge = L.d_edge;
	F.out = out;
	const size_t lds = LW_LDS_BYTES + LW_STAMP_LDS_EXTRA;
	for (uint32_t w = 0; w < LW_FAST_WAVES; w++) {
		F.waves[w] = L.units[w % L.n_units];
		F.waves[w].slot = (uint8_t)(w / L.n_units);
		F.pre[w] = L.pre_on ? L.pre[w % L.n_units] : 0u;
	}
	F.late_from = 0xFFFFFFFFu;
	// k_long needs its 152 KB of dynamic LDS opted in once per device (LwPerDeviceOnce, lw_kernels.hpp)
	static LwPerDeviceOnce once;
	const hipError_t attr_err = once.run([] {
		const void *fns[] = {(const void *)k_long<LW_OUT_I16_PLANAR, false>, (const void *)k_long<LW_OUT_I16_INTERLEAVED, false>,
			(const void *)k_long<LW_OUT_I16_ITL_STEREO, false>, (const void *)k_long<LW_OUT_F32_PLANAR, false>,
			(const void *)k_long<LW_OUT_I16_PLANAR, true>, (const void *)k_long<LW_OUT_I16_PLANAR, false, true>,
			(const void *)k_long<LW_OUT_I16_INTERLEAVED, false, true>, (const void *)k_long<LW_OUT_I16_ITL_STEREO, false, true>,
			(const void *)k_long<LW_OUT_F32_PLANAR, false, true>, (const void *)k_long<LW_OUT_I16_PLANAR, false, false, true>,
			(const void *)k_long<LW_OUT_I16_INTERLEAVED, false, false, true>, (const void *)k_long<LW_OUT_F32_PLANAR, false, false, true>,
			(const void *)k_long<LW_OUT_I16_PLANAR, true, false, false, true>,
			(const void *)k_long<LW_OUT_I16_PLANAR, false, false, false, true>, (const void *)k_long<LW_OUT_I16_INTERLEAVED, false, false, false, true>,
			(const void *)k_long<LW_OUT_F32_PLANAR, false, false, false, true>, (const void *)k_long<LW_OUT_I16_PLANAR, false, false, true, true>,
			(const void *)k_long<LW_OUT_I16_INTERLEAVED, false, false, true, true>, (const void *)k_long<LW_OUT_F32_PLANAR, false, false, true, true>,
			(const void *)k_mix<LW_OUT_I16_PLANAR>, (const void *)k_mix<LW_OUT_I16_INTERLEAVED>, (const void *)k_mix<LW_OUT_F32_PLANAR>,
			// PRE (coupling steps inside the waves): halo pre-pass, plain, TD and EDGE forms in the three sample formats
			(const void *)k_long<LW_OUT_I16_PLANAR, true, false, false, false, true>,
			(const void *)k_long<LW_OUT_I16_PLANAR, false, false, false, false, true>, (const void *)k_long<LW_OUT_I16_INTERLEAVED, false, false, false, false, true>,
			(const void *)k_long<LW_OUT_F32_PLANAR, false, false, false, false, true>,
			(const void *)k_long<LW_OUT_I16_PLANAR, false, true, false, false, true>, (const void *)k_long<LW_OUT_I16_INTERLEAVED, false, true, false, false, true>,
			(const void *)k_long<LW_OUT_F32_PLANAR, false, true, false, false, true>,
			(const void *)k_long<LW_OUT_I16_PLANAR, false, false, true, false, true>, (const void *)k_long<LW_OUT_I16_INTERLEAVED, false, false, true, false, true>,
			(const void *)k_long<LW_OUT_F32_PLANAR, false, false, true, false, true>};
		for (const void *f : fns) {
			const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
			if (e != hipSuccess)
				return e;
		}
		return hipSuccess;
	});
	if (attr_err != hipSuccess)
		return attr_err; // the caller reports this launch; the next one tries again
	if (L.n_halo_items) {
		F.items = L.d_halo_items;
		F.n_items = L.n_halo_items;
		F.per_round = 1; // spread the few halo packets over the whole chip: one packet per workgroup
		F.rounds = 1;
		const hipError_t e = L.pre_on
			? lw_launch_k(k_long<LW_OUT_I16_PLANAR, true, false, false, false, true>, dim3(L.n_halo_items), dim3(LW_WG), lds, st, F)
			: L.split
			? lw_launch_k(k_long<LW_OUT_I16_PLANAR, true, false, false, true>, dim3(L.n_halo_items), dim3(LW_WG), lds, st, F)
			: lw_launch_k(k_long<LW_OUT_I16_PLANAR, true>, dim3(L.n_halo_items), dim3(LW_WG), lds, st, F);
		if (e != hipSuccess)
			return e;
	}
	grid = 0;
	if (L.n_items) {
		F.items = L.d_items;
		F.n_items = L.n_items;
		F.per_round = L.per_round;
		F.rounds = L.rounds;
		F.dense = L.dense;
		const uint32_t chunk = L.per_round * L.rounds;
		grid = (L.n_items + chunk - 1) / chunk;
		F.late_from = L.late_from;
	}
	return hipSuccess;
}

hipError_t lw_launch_long(const LwDevTables &T, const LwBatchDev &B, const LwFastLaunch &L, void *out, int fmt, hipStream_t st)
{
	LwFastArgs F{};
	uint32_t grid = 0;
	const hipError_t pe = long_prepare(T, B, L, out, st, F, grid);
	if (pe != hipSuccess)
		return pe;
	const size_t lds = LW_LDS_BYTES + LW_STAMP_LDS_EXTRA;
	if (L.n_items) {
#define LW_LAUNCH_MAIN(F_)                                                                                     \
	do {                                                                                                      \
		if (L.pre_on) { /* coupling steps inside the waves (never with split units, never the stereo frame stores) */ \
			constexpr int FP = F_ == LW_OUT_I16_ITL_STEREO ? LW_OUT_I16_INTERLEAVED : F_;                      \
			if (L.edge_mode)                                                                                  \
				return lw_launch_k(k_long<FP, false, false, true, false, true>, dim3(grid), dim3(LW_WG), lds, st, F); \
			if (L.has_tdonly)                                                                                 \
				return lw_launch_k(k_long<FP, false, true, false, false, true>, dim3(grid), dim3(LW_WG), lds, st, F); \
			return lw_launch_k(k_long<FP, false, false, false, false, true>, dim3(grid), dim3(LW_WG), lds, st, F); \
		}                                                                                                     \
		if (L.split && !L.has_tdonly) { /* sparse launch: one channel per wave (generic interleaved stores: a wave has one channel) */ \
			if (L.edge_mode)                                                                                  \
				return lw_launch_k(k_long<F_ == LW_OUT_I16_ITL_STEREO ? LW_OUT_I16_INTERLEAVED : F_, false, false, true, true>,  \
						dim3(grid), dim3(LW_WG), lds, st, F);                                                  \
			return lw_launch_k(k_long<F_ == LW_OUT_I16_ITL_STEREO ? LW_OUT_I16_INTERLEAVED : F_, false, false, false, true>, \
					dim3(grid), dim3(LW_WG), lds, st, F);                                                      \
		}                                                                                                     \
		if (L.edge_mode)                                                                                      \
			return lw_launch_k(k_long<F_ == LW_OUT_I16_ITL_STEREO ? LW_OUT_I16_INTERLEAVED : F_, false, false, true>, dim3(grid), \
					dim3(LW_WG), lds, st, F);                                                                  \
		if (L.has_tdonly)                                                                                     \
			return lw_launch_k(k_long<F_, false, true>, dim3(grid), dim3(LW_WG), lds, st, F);                  \
		return lw_launch_k(k_long<F_, false, false>, dim3(grid), dim3(LW_WG), lds, st, F);                    \
	} while (0)
		if (fmt == LW_OUT_I16_PLANAR)
			LW_LAUNCH_MAIN(LW_OUT_I16_PLANAR);
		else if (fmt == LW_OUT_I16_INTERLEAVED && F.ch == 2 && L.n_units == 1 && L.units[0].ch_b >= 0)
			LW_LAUNCH_MAIN(LW_OUT_I16_ITL_STEREO);
		else if (fmt == LW_OUT_I16_INTERLEAVED)
			LW_LAUNCH_MAIN(LW_OUT_I16_INTERLEAVED);
		else
			LW_LAUNCH_MAIN(LW_OUT_F32_PLANAR);
#undef LW_LAUNCH_MAIN
	}
	return hipSuccess;
}

// ---- k_mix: can this pair of launches run as one, and the launch itself
// (the long blocks' launch is the sparse one-channel-per-wave form with raw edges, the short blocks' is 256-point blocks in one pass
// with every unit a channel pair split over two waves, and the whole grid is resident at once: one workgroup per CU)
bool lw_mix_applicable(const LwFastLaunch &LL, const LwShortLaunch &LS, int n_cus)
{
	if (!LL.n_items || !LL.split || !LL.edge_mode || LL.has_tdonly || LL.rounds != 1 || !LS.n_tasks || LS.lanes != 8 || LS.passes > 1)
		return false;
	if ((size_t)LS.n_tasks * LS.n_units > LW_SHORT_SPLIT_BELOW || 2 * LS.n_units > LW_FAST_WAVES)
		return false;
	for (uint32_t u = 0; u < LS.n_units; u++)
		if (LS.units[u].ch_b < 0)
			return false;
	if (LL.per_round * LL.n_units > LW_MIX_LONG_WAVES) // (the long blocks' waves: 0 .. LW_MIX_LONG_WAVES - 1)
		return false;
	const uint32_t chunk = LL.per_round * LL.rounds;
	const size_t n_wg = (LL.n_items + chunk - 1) / chunk;
	return n_wg <= (size_t)std::max(1, n_cus) && (size_t)LS.n_tasks * LS.n_units * 2 <= n_wg * LW_MIX_SHORT_WAVES;
}

// ONE mixed grid (k_mix, k_mix10) per device at a time.  Its short blocks' waves wait for long blocks' waves of other workgroups, which
// is safe because the whole grid is resident -- two such grids on one device (two decoders, two rings, logical shards; different
// streams) could each hold the CUs the other one's missing workgroups are waiting for.  Every launch therefore waits for the previous
// mixed launch of this device, whatever its stream, through one event per device.  (Not while a stream is being captured: a graph's
// launches are ordered by the graph, and an event from outside a capture cannot be waited for inside.)
struct LwMixOrder {
	struct PerDev {
		hipEvent_t done = nullptr;
		bool recorded = false;
	};
	static std::mutex &mu()
	{
		static std::mutex m;
		return m;
	}
	static PerDev *devs()
	{
		static PerDev d[64];
		return d;
	}
	std::unique_lock<std::mutex> lock;
	int dev = 0;
	bool ordered = false;
	hipError_t begin(hipStream_t st)
	{
		hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
		ordered = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && hipStreamIsCapturing(st, &cap) == hipSuccess &&
			cap == hipStreamCaptureStatusNone;
		if (!ordered)
			return hipSuccess;
		lock = std::unique_lock<std::mutex>(mu());
		PerDev &pd = devs()[dev];
		if (!pd.done && hipEventCreateWithFlags(&pd.done, hipEventDisableTiming) != hipSuccess)
			return hipErrorOutOfMemory;
		if (pd.recorded)
			return hipStreamWaitEvent(st, pd.done, 0);
		return hipSuccess;
	}
	hipError_t end(hipError_t e, hipStream_t st) // after the launch: its completion is what the next mixed launch of the device waits for
	{
		if (e == hipSuccess && ordered) {
			PerDev &pd = devs()[dev];
			e = hipEventRecord(pd.done, st);
			pd.recorded = e == hipSuccess;
		}
		return e;
	}
};

hipError_t lw_launch_mix(const LwDevTables &T, const LwBatchDev &B, const LwFastLaunch &LL, const LwShortLaunch &LS, uint32_t *d_flags,
		uint32_t *d_err, uint32_t spin, bool drop_flags, void *out, int fmt, hipStream_t st)
{
	LwFastArgs F{};
	uint32_t grid = 0;
	const hipError_t pe = long_prepare(T, B, LL, out, st, F, grid);
	if (pe != hipSuccess)
		return pe;
	LwShortArgs FS{};
	short_prepare(T, B, LS, out, FS);
	LwMixArgs M{};
	M.flags = d_flags;
	M.err = d_err;
	M.spin = spin ? spin : LW_MIX_SPIN;
	M.drop_flags = drop_flags ? 1u : 0u;
	const size_t lds = LW_MIX_LDS_BYTES + LW_STAMP_LDS_EXTRA;
	LwMixOrder order;
	const hipError_t oe = order.begin(st);
	if (oe != hipSuccess)
		return oe;
	auto launched = [&](hipError_t e) { return order.end(e, st); };
	if (fmt == LW_OUT_I16_PLANAR)
		return launched(lw_launch_k(k_mix<LW_OUT_I16_PLANAR>, dim3(grid), dim3(LW_WG), lds, st, F, FS, M));
	if (fmt == LW_OUT_I16_INTERLEAVED)
		return launched(lw_launch_k(k_mix<LW_OUT_I16_INTERLEAVED>, dim3(grid), dim3(LW_WG), lds, st, F, FS, M));
	return launched(lw_launch_k(k_mix<LW_OUT_F32_PLANAR>, dim3(grid), dim3(LW_WG), lds, st, F, FS, M));
}

// ---------------------------------------------------------------------------------------------
// k_long10: this file's design for blocksize_1 = 10 (n = 1024)
// ---------------------------------------------------------------------------------------------
#include "lw_long10.inc"

// ---------------------------------------------------------------------------------------------
// k_long12: this file's design for blocksize_1 = 12 (n = 4096), one wave per channel
// ---------------------------------------------------------------------------------------------
#include "lw_long12.inc"
