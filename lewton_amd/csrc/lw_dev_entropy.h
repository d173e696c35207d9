// Entropy stage ON THE DEVICE (product code): the bit-serial half of read_audio_packet_generic (audio.rs:921-986 --
// floor-1 decode :215-251 with the amplitude unwrap :391-435, residue decode :587-760) restated so that it compiles for the
// GPU (lw_kernels_entropy.hip: one WAVE per packet) and, unchanged, for the host, where the CPU suite runs it packet by
// packet against the host entropy stage (lw_entropy.cpp) on intact, truncated and mutated packets.
//
// Why: the host stage costs 5.4-6 us per stereo long-block packet and core; a GPU box grants its container 16 CPUs, so the
// staging ring tops out at 2.7-2.8 M packets/s while the synthesis kernels take 16.5 us per 4096 packets.
// Huffman decoding is serial inside a packet but packets are independent: thousands of waves decode thousands of packets
// side by side, and only the packets themselves (~0.5 KB instead of 8.3 KB of records) cross PCIe.
//
// Everything here is plain data and pointers: the setup header is flattened once into one image (lw_dev_entropy.cpp,
// lw::DevEntropyImage) that lives in HBM.  The function produces exactly what the host stage writes into a batch's staging
// -- floor records [ch][fstride] u16 and residue vectors [ch][n/2] f32 before inverse coupling -- so the synthesis kernels
// run unchanged behind it.  Eligible setups only (lw::dev_entropy_build says why not): floor type 1, residue books whose
// dimension divides the partition size, at most 16 channels.
// For those the packet status is decided by the prologue alone (the host reads it: mode number, window flags), so the
// host's planning pass needs nothing back from the device.
#pragma once

#include "lw_records.h"

#include <stdint.h>

#if defined(__HIP__) // compiled as HIP (lw_kernels_entropy.hip); plain C++ translation units get the host version only
#define LW_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define LW_HD inline
#endif

// On the device the packet's working set lives in the wave's LDS: residue accumulator, posts, classification digits
// (address space 3: ds_ instructions, no flat addressing), and the additions of one codeword's vector are spread over the
// lanes (element d on lane d).  On the host the same names are plain pointers and a loop.
//   LW_ENT_LANES(d, cnt)  device: EVERY lane, no masking (a lane >= cnt works on a harmless stand-in -- its own dump slot
//                         behind the accumulators); host: d < cnt
//   LW_LV / LW_L          a value per lane: a register on the device, an array on the host
// There is NO lane-masked region in the device code (no `if (lane < n)`): the packet's control flow is wave-uniform and has
// to look it to the compiler -- behind one divergent branch it treats what the paths merge as divergent, and from there the
// bit reader and every loop around it moved to vector registers and exec masks (tests/test_entropy_kernel_scalar.py counts
// the exec-mask manipulations of the built kernel).
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) float *LwEntAcc;
typedef __attribute__((address_space(3))) uint32_t *LwEntPosts;
typedef __attribute__((address_space(3))) uint16_t *LwEntDigits;
#define LW_ENT_LANE() __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))
#define LW_ENT_LANES(d, cnt) for (uint32_t d = LW_ENT_LANE(), once_ = 1; once_; once_ = 0)
#define LW_LV(type, name) type name
#define LW_L(name) name
typedef __attribute__((address_space(3))) uint8_t *LwEntFlags;
// posts on lanes: post i = h * 64 + lane for h = 0 (and h = 1 when there are 65 posts); LW_PV / LW_P: a value per post
#define LW_ENT_POSTS(i, h, F) _Pragma("unroll") for (uint32_t h = 0; h < 2u; h++) if (h * 64u < (F)) for (uint32_t i = h * 64u + LW_ENT_LANE(), once_ = 1; once_; once_ = 0)
#define LW_PV(type, name) type name[2]
#define LW_P(name) name[h]
#else
typedef float *LwEntAcc;
typedef uint32_t *LwEntPosts;
typedef uint16_t *LwEntDigits;
#define LW_ENT_LANES(d, cnt) for (uint32_t d = 0; d < (cnt); d++)
#define LW_LV(type, name) type name[64]
#define LW_L(name) name[d]
typedef uint8_t *LwEntFlags;
#define LW_ENT_POSTS(i, h, F) for (uint32_t h = 0; h < 1u; h++) for (uint32_t i = 0; i < (F); i++)
#define LW_PV(type, name) type name[LW_MAX_POSTS]
#define LW_P(name) name[i]
#endif
// The image and the packets are read-only for the kernel: on the device they are addressed through the CONSTANT address
// space, so every wave-uniform look-up is a scalar load whatever the compiler can or cannot prove about stores in between
// (behind lw_ent_run's asm statement, which clobbers memory, it can prove nothing: as plain global pointers every look-up of
// the kernel became a vector load and the whole bit reader moved to vector registers).
#if defined(__HIP_DEVICE_COMPILE__)
#define LW_K __attribute__((address_space(4)))
#else
#define LW_K
#endif
#define LW_ENT_DUMP_FLOATS 64u // device: one dump slot per lane behind the accumulators
#define LW_ENT_MAX_LDS (160u * 1024u - 64u) // dynamic LDS k_entropy may ask for (hipFuncAttributeMaxDynamicSharedMemorySize)
#if defined(__HIP_DEVICE_COMPILE__)
#define LW_ENT_ROW_UNIT 4u // a lane's place in a codeword's vector row is kept as a byte offset
// a lane's accumulator is kept as its LDS byte address
#define LW_ENT_AT(out, idx) ((uint32_t)(uintptr_t)(out) + 4u * (idx))
#define LW_ENT_ACC(out, at) (*(LwEntAcc)(uintptr_t)(at))
// a wave-uniform value read from LDS: into a scalar register, so that what is derived from it stays on the scalar unit
#define LW_ENT_SCALAR(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
// (a 64-bit value or a pointer, as an asm statement's scalar operand)
#define LW_ENT_SCALAR64(x) (((uint64_t)LW_ENT_SCALAR((uint64_t)(x) >> 32) << 32) | LW_ENT_SCALAR((uint32_t)(uint64_t)(x)))
__device__ inline __attribute__((always_inline)) uint32_t lw_ent_mad24(uint32_t entry, uint32_t scale, uint32_t add)
{
	uint32_t r; // (entry & 0xffffff) * scale + add; entry is wave-uniform
	asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "s"(entry), "v"(scale), "v"(add));
	return r;
}
#else
#define LW_ENT_SCALAR(x) (x)
#define LW_ENT_ROW_UNIT 1u
#define LW_ENT_AT(out, idx) (idx)
#define LW_ENT_ACC(out, at) (out)[at]
#endif

#define LW_ENT_MAX_CH 16 // (a submap's channel map travels as 16 four-bit entries of one 64-bit scalar: LW_ENT_CMAP)
#define LW_ENT_CMAP(cmap, v) ((uint32_t)(((cmap) >> (4u * (v))) & 0xfu))
#define LW_ENT_MAX_CLASSES 64
#define LW_ENT_MAX_COUPLING 16
#define LW_ENT_LINK 0x80000000u

// (the fields of the two records the decode loops fetch are whole dwords: a wave-uniform dword is a scalar load, a byte or
// a halfword at an odd offset is a vector load)
struct alignas(16) LwEntBook { // 16 bytes, read with one load
	uint32_t lut_off;  // first-level table (2^lut_bits entries) in the image's u32 pool; sub-tables follow at offsets relative to it
	uint32_t vq_off;   // entries * dims floats in the image's f32 pool
	uint32_t shape;    // lut_bits | dims << 8 | (uint16_t)single << 16; single >= 0: single-entry book, any one bit decodes this
	                   // entry (huffman_tree.rs:202-217); -2: empty book; -1: an ordinary book
	uint32_t nodes_off; // binary tree (2 ints per node) in the image's i32 pool for codes beyond the two table levels; ~0u: none
};
#define LW_ENT_SHAPE(lut_bits, dims, single) ((uint32_t)(lut_bits) | (uint32_t)(dims) << 8 | (uint32_t)(uint16_t)(single) << 16)

struct LwEntFloor { // (dwords throughout: see LwEntBook)
	uint32_t multiplier, range_bits, n_part, F, range;
	uint32_t n_levels;               // depth of the posts' dependence on their neighbours (LW_ENT_POST_LEVEL)
	uint32_t part[32];               // per partition: class_dim | subclass bits << 8 | master book << 16 | class << 24
	int32_t sub_books[16 * 8];       // [class][subclass value]
	// per post i in header order: its neighbours lo | hi << 8 (audio.rs:391-435), level << 16: 0 for posts 0 and 1, else
	// 1 + the larger level of its neighbours (posts of one level do not depend on each other: one lane each, level by
	// level); << 24: the header index of the i-th post in ASCENDING-x order (the order of the record)
	uint32_t post[LW_MAX_POSTS];
	uint32_t dx[LW_MAX_POSTS];
	uint32_t magic_lo[LW_MAX_POSTS], magic_hi[LW_MAX_POSTS];
};

struct LwEntResidue {
	uint32_t type, classifications, classbook, cpc;
	uint32_t begin, end, psize;
	uint32_t digits_off; // u16 [classbook entries][cpc] in the image's digit pool (see LwEntTables::digits), or 0xFFFFFFFF: by division
	uint32_t runs_off;   // LwEntRun [classifications][8 passes] in the image's run pool
	uint32_t used_any;   // passes some class of this residue uses at all
	uint8_t vals_used[LW_ENT_MAX_CLASSES];
};

// One (residue, class, pass): everything a partition's run of codewords needs, fetched with ONE 32-byte load (through the
// residue's class table and the book table it took four dependent accesses and a division per run)
struct alignas(16) LwEntRun {
	uint32_t lut_off, vq_off; // as LwEntBook, but in BYTES
	uint32_t shape, nodes_off; // as LwEntBook
	uint32_t count; // codewords per partition: type 0 psize / dims, types 1/2 ceil(psize / dims) (audio.rs:587-618)
	uint32_t step;  // element stride inside a codeword's vector: type 0 psize / dims, types 1/2 1 (audio.rs:587-618)
	uint32_t adv;   // elements between the starts of consecutive codewords: type 0 1, types 1/2 dims
	uint32_t fast;  // bit k (1..8): an ordinary book with count >= 1 whose dimension is a multiple of k and (types 1/2)
	                // divides the partition size: the runs lw_ent_range and lw_ent_run take
};

struct LwEntMode {
	uint8_t blockflag, n_coupling, n_submaps, pad;
	uint8_t floor_of_ch[LW_ENT_MAX_CH];
	uint8_t mux[LW_ENT_MAX_CH];           // submap of every channel
	uint8_t submap_residue[16];
	uint8_t mag[LW_ENT_MAX_COUPLING], ang[LW_ENT_MAX_COUPLING];
};

// Resolved view of the image (device pointers on the GPU, host pointers in the CPU harness)
struct LwEntTables {
	const LW_K LwEntBook *books;
	const LW_K LwEntFloor *floors;
	const LW_K LwEntResidue *residues;
	const LW_K LwEntMode *modes;
	const LW_K uint32_t *lut;
	const LW_K float *vq;
	const LW_K uint16_t *digits; // per classbook entry and digit: class | vals_used[class] << 8
	const LW_K LwEntRun *runs;
	const LW_K int32_t *nodes;
	uint32_t ch, fstride;
	uint32_t ws_bytes;  // per-packet scratch: posts (4 * LW_MAX_POSTS rounded up) + classification digits (u16)
	uint32_t res_floats; // largest residue block of a packet: ch * blocksize_1 / 2
	uint32_t general;    // 0: every mapping has one submap with all channels (the usual case; its own kernel instantiation)
};

// One packet of a device-entropy batch
struct LwEntPacket { // 16 bytes
	uint32_t word_off; // packet bytes in the batch's packet pool (u32 words; followed by >= 3 zero words)
	uint32_t len;      // bytes
	uint8_t start_bit; // first bit after the prologue (audio.rs:921-938, read by the host)
	uint8_t pad[3];
	uint32_t pad1;
};

// posts scratch: u32 y[LW_MAX_POSTS] + one dump entry, then u8 step-2 flags [LW_MAX_POSTS] + one dump entry
#define LW_ENT_Y_DUMP LW_MAX_POSTS
#define LW_ENT_FLAGS_OFF (4u * (LW_MAX_POSTS + 1u))
#define LW_ENT_POSTS_BYTES ((LW_ENT_FLAGS_OFF + LW_MAX_POSTS + 1u + 15u) & ~15u)

// A codebook as the decode loops hold it: everything in registers, fetched with ONE load of the 16-byte table entry.  (Read
// field by field through a pointer, every codeword paid 3-4 dependent table accesses before its own look-up: the compiler
// may not keep byte-typed fields in registers across the residue stores.)
struct LwEntBookRegs {
	const LW_K uint32_t *lut;
	const LW_K float *vq;
	const LW_K int32_t *nodes; // tree for the (rare) codes longer than the two table levels, or null
	uint32_t lut_mask, lut_bits, dims;
	int32_t single;
};

LW_HD LwEntBookRegs lw_ent_book(const LwEntTables &T, uint32_t bi)
{
	const LW_K LwEntBook &b = T.books[LW_ENT_SCALAR(bi)]; // (a book number may come out of a byte table: a vector load)
	LwEntBookRegs r;
	r.lut = T.lut + b.lut_off;
	r.vq = T.vq + b.vq_off;
	const uint32_t shape = b.shape;
	r.lut_bits = shape & 0xffu;
	r.lut_mask = (1u << r.lut_bits) - 1u;
	r.dims = (shape >> 8) & 0xffu;
	r.single = (int16_t)(shape >> 16);
	r.nodes = b.nodes_off != 0xFFFFFFFFu ? T.nodes + b.nodes_off : nullptr;
	return r;
}

// LSb-first reader with the bit window in registers: `win` holds the next `have` bits (>= 32 after peek()), `nxt` the word
// after them, requested one refill ahead -- the packet bytes are read sequentially whatever the code lengths, so their
// loads are never on a codeword's dependency chain.  The pool keeps >= 3 zero words behind every packet.
// `left` = bits of the packet not yet consumed: the reference's bounds checks (bitpacking.rs:291-297, huffman_tree.rs:362-381)
// are comparisons against it.
// Table entries as the image holds them: (len << 24) | entry for a code of the table level; LW_ENT_LINK | (bits << 24) | offset for
// a prefix with a second-level table; LW_ENT_WALK for "not in the tables" (a longer code: the tree; the host's tables have
// length 0 there) -- so that ONE sign test separates the ordinary entries from the rest.
#define LW_ENT_WALK 0xFF000000u
#define LW_ENT_SPECIAL(e) ((int32_t)(e) < 0)
struct LwEntReader {
	const LW_K uint32_t *w;
	uint32_t nbits, left;
	uint64_t win;
	uint32_t have, wo, nxt; // wo: byte offset of the word after nxt

	LW_HD void init(const LW_K uint32_t *words, uint32_t len_bytes, uint32_t start_bit)
	{
		w = words;
		nbits = len_bytes * 8u;
		left = nbits - start_bit;
		const uint32_t i = start_bit >> 5, s = start_bit & 31u;
		win = (uint64_t)(w[i] >> s);
		have = 32u - s;
		nxt = w[i + 1];
		wo = (i + 2) * 4u;
	}
	// the next 32 bits, zero past the end of the packet
	LW_HD uint32_t peek()
	{
		if (have < 32u) {
			win |= (uint64_t)nxt << have;
			have += 32u;
			nxt = *(const LW_K uint32_t *)((const LW_K char *)w + wo);
			wo += 4u;
		}
		return (uint32_t)win;
	}
	LW_HD void skip(uint32_t n) // n <= 32, n <= left, after peek()
	{
		win >>= n;
		have -= n;
		left -= n;
	}
	// bitpacking.rs:291-297: a fixed-width read that does not fit fails without consuming anything; n <= 32
	LW_HD bool read(uint32_t n, uint32_t &v)
	{
		if (n == 0) {
			v = 0;
			return true;
		}
		if (n > left)
			return false;
		const uint32_t x = peek();
		v = n >= 32 ? x : (x & ((1u << n) - 1u));
		skip(n);
		return true;
	}
	// huffman_tree.rs:362-381 through the two table levels: a code that runs past the end consumes the rest and fails
	// (after that every read fails on its bounds check; the window is not looked at again).
	// In two halves, so that a caller can put other work between the table load and its first use: probe() starts the
	// first-level look-up of an ordinary book, finish() does everything else.
	LW_HD uint32_t probe(const LW_K uint32_t *lut, uint32_t lut_mask)
	{
		return lut[peek() & lut_mask];
	}
	LW_HD bool finish(const LW_K uint32_t *lut, uint32_t lut_bits, const LW_K int32_t *nodes, uint32_t e, uint32_t &sym)
	{
		if (LW_ENT_SPECIAL(e)) {
			if (e < LW_ENT_WALK) // a link
				e = lut[(e & 0xffffffu) + (((uint32_t)win >> lut_bits) & ((1u << ((e >> 24) & 0x7fu)) - 1u))];
			if (LW_ENT_SPECIAL(e)) // (one call site: walk() is inlined wherever finish() is)
				return walk(nodes, sym);
		}
		const uint32_t len = e >> 24;
		if (len > left) {
			left = 0;
			return false;
		}
		skip(len);
		sym = e; // (len << 24) | entry: the callers use the low 24 bits
		return true;
	}
	LW_HD bool code(const LwEntBookRegs &b, uint32_t &sym)
	{
		if (b.single >= 0) {
			if (left < 1)
				return false;
			(void)peek();
			skip(1);
			sym = (uint32_t)b.single;
			return true;
		}
		if (b.single == -2) { // empty book: the reference panics; like the host stage, the packet ends here
			left = 0;
			return false;
		}
		const uint32_t e = probe(b.lut, b.lut_mask);
		if (!finish(b.lut, b.lut_bits, b.nodes, e, sym))
			return false;
		sym &= 0xffffffu;
		return true;
	}
	// a code beyond the table levels (19+ bits: one in 2^18 codewords of a real encoder's books): bit by bit through the tree,
	// straight from the packet words
	LW_HD bool walk(const LW_K int32_t *nodes, uint32_t &sym)
	{
		if (!nodes) {
			left = 0;
			return false;
		}
		int32_t node = 0;
		const uint32_t pos = nbits - left;
		uint32_t p = pos;
		for (;;) {
			if (p >= nbits) {
				left = 0;
				return false;
			}
			const uint32_t bit = (w[p >> 5] >> (p & 31u)) & 1u;
			p++;
			const int32_t c = nodes[2 * node + (int32_t)bit];
			if (c == (int32_t)0x80000000) {
				left = 0;
				return false;
			}
			if (c < 0) {
				sym = (uint32_t)~c;
				uint32_t n = p - pos;
				while (n) { // (more than 32 bits are possible)
					const uint32_t s = n < 32u ? n : 32u;
					(void)peek();
					skip(s);
					n -= s;
				}
				return true;
			}
			node = c;
		}
	}
};

// audio.rs:215-251; y = the packet's scratch.  false = unused floor (FloorSpecialCase::Unused)
LW_HD bool lw_ent_floor_decode(const LwEntTables &T, const LW_K LwEntFloor &fl, LwEntReader &r, LwEntPosts y)
{
	uint32_t nonzero;
	if (!r.read(1, nonzero) || !nonzero)
		return false;
	uint32_t k = 0, v;
	const uint32_t range_bits = fl.range_bits, n_part = fl.n_part;
	if (!r.read(range_bits, v))
		return false;
	y[k++] = v;
	if (!r.read(range_bits, v))
		return false;
	y[k++] = v;
	for (uint32_t p = 0; p < n_part; p++) {
		const uint32_t pt = fl.part[p];
		const uint32_t cdim = pt & 0xffu, cbits = (pt >> 8) & 0xffu, c = pt >> 24;
		const uint32_t csub = (1u << cbits) - 1u;
		uint32_t cval = 0;
		if (cbits && !r.code(lw_ent_book(T, (pt >> 16) & 0xffu), cval))
			return false;
		for (uint32_t d = 0; d < cdim; d++) {
			const int32_t book = fl.sub_books[c * 8u + (cval & csub)];
			cval >>= cbits;
			v = 0;
			if (book >= 0 && !r.code(lw_ent_book(T, (uint32_t)book), v))
				return false;
			y[k++] = v;
		}
	}
	return true;
}

// audio.rs:354-367 with wrapping u32 arithmetic; the division by the header constant adx is a multiplication by its
// precomputed 2^64 / adx + 1 (exact for every 32-bit dividend)
LW_HD uint32_t lw_ent_render_point(uint32_t y0, uint32_t y1, uint32_t dx, uint32_t magic_lo, uint32_t magic_hi)
{
	const int32_t dy = (int32_t)(y1 - y0);
	const uint32_t ady = dy < 0 ? 0u - (uint32_t)dy : (uint32_t)dy;
	const uint32_t num = ady * dx;
	// (num * magic) >> 64 in 64-bit pieces: adx >= 2 (a post lies strictly between its neighbours), so magic <= 2^63 + 1
	const uint64_t hi = (uint64_t)num * magic_hi, lo = (uint64_t)num * magic_lo;
	const uint32_t off = (uint32_t)((hi + (lo >> 32)) >> 32);
	return dy < 0 ? y0 - off : y0 + off;
}

// audio.rs:391-435 -> device record: per post in ascending-x order, (final_y * multiplier) | active flag.  One lane per
// post, level by level: a post's neighbours lie on lower levels, the posts of one level are independent of each other
// (the serial loop over the posts cost the device more than their decoding).  y is updated in place.
LW_HD void lw_ent_floor_record(const LW_K LwEntFloor &fl, LwEntPosts y, uint16_t *rec)
{
	const uint32_t F = fl.F, range = fl.range, n_levels = fl.n_levels, mult = fl.multiplier;
	LwEntFlags flag = (LwEntFlags)y + LW_ENT_FLAGS_OFF;
	LW_PV(uint32_t, pk);
	LW_PV(uint32_t, dxv);
	LW_PV(uint32_t, mlo);
	LW_PV(uint32_t, mhi);
	LW_PV(int32_t, val);
	LW_ENT_POSTS(i, h, F) {
		const uint32_t ii = i < F ? i : F - 1u; // (device: the lanes beyond the last post repeat its work; no lane is masked off)
		LW_P(pk) = fl.post[ii];
		LW_P(dxv) = fl.dx[ii];
		LW_P(mlo) = fl.magic_lo[ii];
		LW_P(mhi) = fl.magic_hi[ii];
		LW_P(val) = (int32_t)y[ii];
		flag[i < F ? i : LW_ENT_Y_DUMP] = i < 2u ? 1u : 0u;
	}
	for (uint32_t lvl = 1; lvl <= n_levels; lvl++) {
		LW_ENT_POSTS(i, h, F) {
			const uint32_t p = LW_P(pk);
			const uint32_t lo = p & 0xffu, hi = (p >> 8) & 0xffu;
			const bool mine = i < F && ((p >> 16) & 0xffu) == lvl;
			const int32_t predicted = (int32_t)lw_ent_render_point(y[lo], y[hi], LW_P(dxv), LW_P(mlo), LW_P(mhi));
			const int32_t v = LW_P(val);
			const int32_t highroom = (int32_t)(range - (uint32_t)predicted);
			const int32_t lowroom = predicted;
			const int32_t room = (int32_t)((uint32_t)(highroom < lowroom ? highroom : lowroom) * 2u);
			uint32_t fy = (uint32_t)predicted;
			if (v > 0) {
				if (v >= room) {
					fy = highroom > lowroom ? (uint32_t)predicted + (uint32_t)v - (uint32_t)lowroom
					                        : (uint32_t)predicted - (uint32_t)v + (uint32_t)highroom - 1u;
				} else {
					const int32_t t = (v % 2 == 1) ? (int32_t)(0u - (uint32_t)v - 1u) : v;
					fy = (uint32_t)predicted + (uint32_t)(t >> 1);
				}
			}
			y[mine ? i : LW_ENT_Y_DUMP] = fy;
			// step 2 flags of the post and of both neighbours (a lane without a mark to set writes the dump entry)
			const bool marks = mine && v > 0;
			flag[marks ? lo : LW_ENT_Y_DUMP] = 1u;
			flag[marks ? hi : LW_ENT_Y_DUMP] = 1u;
			flag[marks ? i : LW_ENT_Y_DUMP] = 1u;
		}
	}
	LW_ENT_POSTS(i, h, F) { // i: position in ascending-x order
		const uint32_t src = LW_P(pk) >> 24;
		const uint32_t fy = y[src] < range - 1u ? y[src] : range - 1u; // :431-433
		const uint32_t on = flag[src];
		rec[i < F ? i : F - 1u] = (uint16_t)(((fy * mult) & 0xffu) | (on ? LW_POST_ACTIVE : 0u));
	}
}

#if defined(__HIP_DEVICE_COMPILE__)
// ---- The steady state of the residue decode, by hand (gfx950).  One CU's ONE scalar unit serves its 16 resident packets, so
// the scalar instructions per codeword pace this kernel: here 10 + 4 branches / waits, 4 vector instructions, one table
// look-up, one row load, one LDS read-modify-write -- what the compiler makes of the same loop in C++ is ~50 + 19.
//   entry = lut[window & mask] | add the vector of the codeword before the last one | link -> second-level entry |
//   bounds (left -= len borrows: the code runs past the end) | window >>= len | bits -= len borrows: refill | row load
// Fixed registers inside the statements (clobbers; everything else is an operand):
//   s[64:65] bit window, s[66:67] scratch pair, s[68:75] the run record, s[76:77] table, s[78:79] rows, s80 mask, s81 table bits
//   v53 scratch, v54 dims * 4, v55 the lane's byte in a row, v56 its step per codeword, v57 digit address, v58 its accumulator
//   address for the next codeword, v59 scratch, v60/v61 and v62/v63 the two pending vectors (address, value)
// %[have] holds the window's bit count MINUS 32 inside the statements (its borrow is the refill test).
// The vectors are added TWO codewords behind the decoder (two pending slots used in turn: the step exists twice): a row load
// has two look-ups to arrive in; one behind, every codeword waited ~350 cycles for its predecessor's row.  Additions to one
// accumulator keep their order (the slots are a FIFO; within a run no two codewords touch the same element).  Every statement
// starts and ends with nothing pending and no load outstanding.
#define LW_ENT_ASM_REFILL /* the window has < 32 bits: %[have] (biased) has wrapped below zero */                      \
	"s_add_u32 %[have], %[have], 32\n"                                                                                \
	"s_mov_b32 s66, %[nxt]\n"                                                                                         \
	"s_mov_b32 s67, 0\n"                                                                                              \
	"s_lshl_b64 s[66:67], s[66:67], %[have]\n"                                                                        \
	"s_or_b64 s[64:65], s[64:65], s[66:67]\n"                                                                         \
	"s_load_dword %[nxt], %[w], %[wo]\n"                                                                              \
	"s_add_u32 %[wo], %[wo], 4\n"
#define LW_ENT_ASM_ENTER /* the window into its pair, the bit count biased, >= 32 bits in the window */                 \
	"s_mov_b64 s[64:65], %[win]\n"                                                                                    \
	"s_sub_u32 %[have], %[have], 32\n"                                                                                \
	"s_cbranch_scc0 10f\n" LW_ENT_ASM_REFILL "10:\n"                                                                  \
	"v_mov_b32 v60, %[dump]\n"                                                                                        \
	"v_mov_b32 v61, 0\n"                                                                                              \
	"v_mov_b32 v62, %[dump]\n"                                                                                        \
	"v_mov_b32 v63, 0\n"
// One codeword through pending slot P (accumulator address) / PV (value).  NEXT: the other slot's step; PH_DONE: the slot
// the next run starts with; PH_EXIT: drain order when leaving from here (this slot is empty then: the other one first);
// LUT .. INC: where the run's table, rows, mask, table bits, dims * 4, row byte and step live (fixed registers in
// lw_ent_range, operands in lw_ent_run: there every register the statement clobbers is one the C++ around it has to spill).
#define LW_ENT_ASM_STEP(P, PV, NEXT, PH_DONE, PH_EXIT, LUT, VQ, MASK, BITS, VD4, ROW, INC)                                                                \
	"s_and_b32 %[t0], s64, " MASK "\n"                                                                                     \
	"s_lshl_b32 %[t0], %[t0], 2\n"                                                                                    \
	"s_load_dword %[e], " LUT ", %[t0]\n"                                                                            \
	"ds_read_b32 v59, v" P "\n"                                                                                       \
	"s_waitcnt vmcnt(1) lgkmcnt(0)\n"                                                                                 \
	"v_add_f32 v59, v59, v" PV "\n"                                                                                   \
	"ds_write_b32 v" P ", v59\n"                                                                                      \
	"s_cmp_lt_i32 %[e], 0\n" /* a link to a second-level table, or "not in the tables" */                            \
	"s_cbranch_scc1 " P "4f\n" P "3:\n"                                                                               \
	"s_lshr_b32 %[t0], %[e], 24\n"                                                                                    \
	"s_sub_u32 %[left], %[left], %[t0]\n"                                                                             \
	"s_cbranch_scc1 " P "6f\n"                                                                                        \
	"s_lshr_b64 s[64:65], s[64:65], %[t0]\n"                                                                          \
	"s_sub_u32 %[have], %[have], %[t0]\n"                                                                             \
	"s_cbranch_scc0 " P "2f\n" LW_ENT_ASM_REFILL P "2:\n"                                                             \
	"v_mad_u32_u24 v59, %[e], " VD4 ", " ROW "\n"                                                                             \
	"global_load_dword v" PV ", v59, " VQ "\n"                                                                      \
	"v_mov_b32 v" P ", v58\n"                                                                                         \
	"v_add_u32 v58, v58, " INC "\n"                                                                                       \
	"s_add_u32 %[neg], %[neg], 1\n"                                                                                   \
	"s_cbranch_scc0 " NEXT "\n"                                                                                       \
	"s_mov_b32 %[ph], " PH_DONE "\n"                                                                                  \
	"s_branch 28f\n" P "4:\n"                                                                                         \
	"s_bfe_u32 %[t0], %[e], 0x70018\n"                                                                                \
	"s_cmp_eq_u32 %[t0], 0x7f\n"                                                                                      \
	"s_cbranch_scc1 " P "5f\n"                                                                                        \
	"s_lshr_b32 %[t1], s64, " BITS "\n"                                                                                    \
	"s_bfm_b32 %[t0], %[t0], 0\n"                                                                                     \
	"s_and_b32 %[t1], %[t1], %[t0]\n"                                                                                 \
	"s_and_b32 %[t0], %[e], 0xffffff\n"                                                                               \
	"s_add_u32 %[t0], %[t0], %[t1]\n"                                                                                 \
	"s_lshl_b32 %[t0], %[t0], 2\n"                                                                                    \
	"s_load_dword %[e], " LUT ", %[t0]\n"                                                                            \
	"s_waitcnt lgkmcnt(0)\n"                                                                                          \
	"s_cmp_lt_i32 %[e], 0\n"                                                                                          \
	"s_cbranch_scc0 " P "3b\n" P "5:\n"                                                                               \
	"s_mov_b32 %[st], 1\n"                                                                                            \
	"v_mov_b32 v" P ", %[dump]\n"                                                                                     \
	"s_mov_b32 %[ph], " PH_EXIT "\n"                                                                                  \
	"s_branch 70f\n" P "6:\n"                                                                                         \
	"s_mov_b32 %[st], 2\n"                                                                                            \
	"v_mov_b32 v" P ", %[dump]\n"                                                                                     \
	"s_mov_b32 %[ph], " PH_EXIT "\n"                                                                                  \
	"s_branch 70f\n"
#define LW_ENT_ASM_STEP_V(...) LW_ENT_ASM_STEP(__VA_ARGS__) // (the register list of a caller is one macro: expanded first)
#define LW_ENT_ASM_FLUSH(P, PV)                                                                                       \
	"ds_read_b32 v59, v" P "\n"                                                                                       \
	"s_waitcnt lgkmcnt(0)\n"                                                                                          \
	"v_add_f32 v59, v59, v" PV "\n"                                                                                   \
	"ds_write_b32 v" P ", v59\n"
#define LW_ENT_ASM_LEAVE /* the older pending vector first (ph: the slot the next codeword would have used) */          \
	"70:\n"                                                                                                           \
	"s_waitcnt vmcnt(0)\n"                                                                                            \
	"s_cmp_eq_u32 %[ph], 0\n"                                                                                         \
	"s_cbranch_scc0 72f\n" LW_ENT_ASM_FLUSH("60", "61") LW_ENT_ASM_FLUSH("62", "63") "s_branch 73f\n"                \
	"72:\n" LW_ENT_ASM_FLUSH("62", "63") LW_ENT_ASM_FLUSH("60", "61") "73:\n"                                         \
	"s_waitcnt lgkmcnt(0)\n"                                                                                          \
	"s_add_u32 %[have], %[have], 32\n"                                                                                \
	"s_mov_b64 %[win], s[64:65]\n"
#define LW_ENT_ASM_CLOBBERS                                                                                           \
	"s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "v53",  \
		"v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "vcc", "scc", "memory"

#define LW_ENT_RUN_REGS "%[lut]", "%[vq]", "%[mask]", "%[bits]", "%[vd4]", "%[row]", "%[inc]"
// The run of ONE partition (the C++ path behind a return of lw_ent_range, and the vectors it does not take).
// Returns 0: `todo` codewords done (todo = 0); 1: the codeword at the head of the window needs the tree (its table entry in
// `e`), nothing of it consumed; 2: the codeword runs past the end of the packet.  `at` is not changed: the caller advances it by inc per codeword.
__device__ inline __attribute__((always_inline)) uint32_t lw_ent_run(LwEntReader &r, const LW_K uint32_t *lut, const LW_K float *vq,
		uint32_t lut_mask, uint32_t lut_bits, uint32_t &todo, uint32_t &e, uint32_t at, uint32_t inc, uint32_t row, uint32_t vdims4,
		uint32_t dump)
{
	uint32_t st, t0, t1, ph = 0;
	uint32_t neg = LW_ENT_SCALAR(0u - todo); // (wave-uniform like everything scalar here; the compiler keeps this one in a vector register)
	asm volatile(LW_ENT_ASM_ENTER
	             "v_mov_b32 v58, %[at]\n"
	             "79:\n" LW_ENT_ASM_STEP_V("60", "61", "80f", "1", "1", LW_ENT_RUN_REGS) "80:\n" LW_ENT_ASM_STEP_V("62", "63", "79b", "0", "0", LW_ENT_RUN_REGS)
	             "28:\n"
	             "s_mov_b32 %[st], 0\n" LW_ENT_ASM_LEAVE
	             : [win] "+s"(r.win), [have] "+s"(r.have), [left] "+s"(r.left), [nxt] "+s"(r.nxt), [wo] "+s"(r.wo), [neg] "+s"(neg),
	               [ph] "+s"(ph), [e] "=&s"(e), [st] "=&s"(st), [t0] "=&s"(t0), [t1] "=&s"(t1)
	             : [w] "s"(r.w), [lut] "s"(lut), [vq] "s"(vq), [mask] "s"(lut_mask), [bits] "s"(lut_bits), [at] "v"(at), [inc] "v"(inc),
	               [row] "v"(row), [vd4] "v"(vdims4), [dump] "v"(dump)
	             : "s64", "s65", "s66", "s67", "v58", "v59", "v60", "v61", "v62", "v63", "scc", "memory");
	todo = 0u - neg;
	return st;
}

// ---- lw_ent_range's statement in pieces (DEINT 0 / 2 differ in the PLACE piece)
#define LW_ENT_RANGE_ENTER LW_ENT_ASM_ENTER                                                                           \
	"v_mov_b32 v57, %[cls]\n"                                                                                         \
	"20:\n" /* ---- the next partition */
// pass 0 of a single vector: the class word of a group of cpc partitions, decoded where the group starts (gc = partitions of
// the current group still to visit); its digits go from the image's digit table to the LDS digits (lane q < cpc each one)
#define LW_ENT_RANGE_CLASSWORD                                                                                        \
	"s_cmp_lg_u32 %[gc], 0\n"                                                                                         \
	"s_cbranch_scc1 29f\n"                                                                                            \
	"s_and_b32 %[t0], s64, %[cbmask]\n"                                                                               \
	"s_lshl_b32 %[t0], %[t0], 2\n"                                                                                    \
	"s_load_dword %[e], %[cblut], %[t0]\n"                                                                            \
	"s_waitcnt lgkmcnt(0)\n"                                                                                          \
	"s_cmp_lt_i32 %[e], 0\n"                                                                                          \
	"s_cbranch_scc1 24f\n"                                                                                            \
	"23:\n"                                                                                                           \
	"s_lshr_b32 %[t0], %[e], 24\n"                                                                                    \
	"s_sub_u32 %[left], %[left], %[t0]\n"                                                                             \
	"s_cbranch_scc1 26f\n"                                                                                            \
	"s_lshr_b64 s[64:65], s[64:65], %[t0]\n"                                                                          \
	"s_sub_u32 %[have], %[have], %[t0]\n"                                                                             \
	"s_cbranch_scc0 22f\n" LW_ENT_ASM_REFILL "22:\n"                                                                 \
	"s_and_b32 %[t0], %[e], 0xffffff\n"                                                                               \
	"s_mul_i32 %[t0], %[t0], %[cpc]\n"                                                                                \
	"v_lshl_add_u32 v59, %[t0], 1, %[q2]\n"                                                                           \
	"global_load_ushort v53, v59, %[digits]\n"                                                                        \
	"v_add_u32 v59, v57, %[q2]\n"                                                                                     \
	"s_waitcnt vmcnt(0)\n"                                                                                            \
	"ds_write_b16 v59, v53\n"                                                                                         \
	"s_mov_b32 %[gc], %[cpc]\n"                                                                                       \
	"s_branch 29f\n"                                                                                                  \
	"24:\n"                                                                                                           \
	"s_bfe_u32 %[t0], %[e], 0x70018\n"                                                                                \
	"s_cmp_eq_u32 %[t0], 0x7f\n"                                                                                      \
	"s_cbranch_scc1 25f\n"                                                                                            \
	"s_lshr_b32 %[t1], s64, %[cbbits]\n"                                                                              \
	"s_bfm_b32 %[t0], %[t0], 0\n"                                                                                     \
	"s_and_b32 %[t1], %[t1], %[t0]\n"                                                                                 \
	"s_and_b32 %[t0], %[e], 0xffffff\n"                                                                               \
	"s_add_u32 %[t0], %[t0], %[t1]\n"                                                                                 \
	"s_lshl_b32 %[t0], %[t0], 2\n"                                                                                    \
	"s_load_dword %[e], %[cblut], %[t0]\n"                                                                            \
	"s_waitcnt lgkmcnt(0)\n"                                                                                          \
	"s_cmp_lt_i32 %[e], 0\n"                                                                                          \
	"s_cbranch_scc0 23b\n"                                                                                            \
	"25:\n"                                                                                                           \
	"s_mov_b32 %[st], 4\n"                                                                                            \
	"s_branch 70f\n"                                                                                                  \
	"26:\n"                                                                                                           \
	"s_mov_b32 %[st], 2\n"                                                                                            \
	"s_branch 70f\n"                                                                                                  \
	"29:\n"                                                                                                           \
	"s_sub_u32 %[gc], %[gc], 1\n"
#define LW_ENT_RANGE_VISIT /* the partition's class digit | the passes of that class << 8 */                           \
	"ds_read_u16 v59, v57\n"                                                                                          \
	"v_add_u32 v57, 2, v57\n"                                                                                         \
	"s_waitcnt lgkmcnt(0)\n"                                                                                          \
	"v_readfirstlane_b32 %[t0], v59\n"                                                                                \
	"s_and_b32 %[t1], %[t0], %[passmask]\n"                                                                           \
	"s_cbranch_scc0 28f\n"                                                                                            \
	"s_and_b32 %[t0], %[t0], 0xff\n"                                                                                  \
	"s_lshl_b32 %[t0], %[t0], 8\n"                                                                                    \
	"s_add_u32 %[t0], %[t0], %[pass32]\n"                                                                             \
	"s_load_dwordx8 s[68:75], %[runs], %[t0]\n" /* lut, vq (byte offsets), shape, nodes, count, step, adv, fast */    \
	"s_waitcnt lgkmcnt(0)\n"                                                                                          \
	"s_bitcmp1_b32 s75, %[dper]\n"                                                                                    \
	"s_cbranch_scc0 33f\n"                                                                                            \
	"s_add_u32 s76, %[lutlo], s68\n"                                                                                  \
	"s_addc_u32 s77, %[luthi], 0\n"                                                                                   \
	"s_add_u32 s78, %[vqlo], s69\n"                                                                                   \
	"s_addc_u32 s79, %[vqhi], 0\n"                                                                                    \
	"s_and_b32 s81, s70, 0xff\n"                                                                                      \
	"s_bfm_b32 s80, s81, 0\n"                                                                                         \
	"s_bfe_u32 %[t0], s70, 0x80008\n" /* dims */                                                                      \
	"s_sub_u32 %[neg], 0, s72\n"                                                                                      \
	"v_cmp_gt_u32 vcc, %[t0], %[lane]\n" /* this lane holds an element of the codewords' vectors */                   \
	"s_lshl_b32 %[t1], %[t0], 2\n"                                                                                    \
	"v_mov_b32 v54, %[t1]\n"                                                                                          \
	"v_lshlrev_b32 v59, 2, %[lane]\n"                                                                                 \
	"v_cndmask_b32 v55, 0, v59, vcc\n"                                                                                \
	"v_mov_b32 v53, %[el]\n"                                                                                          \
	"v_mad_u32_u24 v59, %[lane], s73, v53\n"
// the lane's accumulator address (v59) and its step from one codeword to the next in bytes (t1)
#define LW_ENT_RANGE_PLACE0                                                                                           \
	"v_lshl_add_u32 v59, v59, 2, %[accbase]\n"                                                                        \
	"s_lshl_b32 %[t1], s74, 2\n"
#define LW_ENT_RANGE_PLACE2                                                                                           \
	"v_and_b32 v53, 1, v59\n"                                                                                         \
	"v_lshrrev_b32 v59, 1, v59\n"                                                                                     \
	"v_mul_u32_u24 v53, %[half4], v53\n"                                                                              \
	"v_lshl_add_u32 v59, v59, 2, v53\n"                                                                               \
	"v_add_u32 v59, %[accbase], v59\n"                                                                                \
	"s_lshl_b32 %[t1], %[t0], 1\n"
#define LW_ENT_RANGE_REGS "s[76:77]", "s[78:79]", "s80", "s81", "v54", "v55", "v56"
#define LW_ENT_RANGE_TAIL                                                                                             \
	"v_cndmask_b32 v58, %[dump], v59, vcc\n"                                                                          \
	"v_mov_b32 v59, %[t1]\n"                                                                                          \
	"v_cndmask_b32 v56, 0, v59, vcc\n"                                                                                \
	"s_cmp_eq_u32 %[ph], 0\n"                                                                                         \
	"s_cbranch_scc0 80f\n"                                                                                            \
	"79:\n" LW_ENT_ASM_STEP_V("60", "61", "80f", "1", "1", LW_ENT_RANGE_REGS) "80:\n" LW_ENT_ASM_STEP_V("62", "63", "79b", "0", "0", LW_ENT_RANGE_REGS) \
	"28:\n" /* ---- on to the next partition */                                                                       \
	"s_add_u32 %[el], %[el], %[psize]\n"                                                                              \
	"s_sub_u32 %[n], %[n], 1\n"                                                                                       \
	"s_cmp_lg_u32 %[n], 0\n"                                                                                          \
	"s_cbranch_scc1 20b\n"                                                                                            \
	"s_mov_b32 %[st], 0\n"                                                                                            \
	"s_branch 70f\n"                                                                                                  \
	"33:\n"                                                                                                           \
	"s_mov_b32 %[st], 3\n" LW_ENT_ASM_LEAVE
#define LW_ENT_RANGE_OPERANDS                                                                                         \
	: [win] "+s"(r.win), [have] "+s"(r.have), [left] "+s"(r.left), [nxt] "+s"(r.nxt), [wo] "+s"(r.wo), [neg] "+s"(neg),  \
	  [n] "+s"(n), [el] "+s"(el), [ph] "+s"(ph), [e] "=&s"(e), [st] "=&s"(st), [t0] "=&s"(t0), [t1] "=&s"(t1)          \
	: [w] "s"(r.w), [runs] "s"(runs), [lutlo] "s"(lutlo), [luthi] "s"(luthi), [vqlo] "s"(vqlo), [vqhi] "s"(vqhi),        \
	  [pass32] "s"(pass32), [passmask] "s"(passmask), [dper] "s"(dper), [psize] "s"(psize), [cls] "s"(cls_at),           \
	  [accbase] "s"(acc_base), [half4] "s"(half4), [lane] "v"(lane), [dump] "v"(dump)                                    \
	: LW_ENT_ASM_CLOBBERS

// A RANGE of partitions of one vector in one pass: per partition the class digit (LDS), the test whether the class has a book
// in this pass, the (residue, class, pass) record, the lanes' accumulator addresses, then the run of codewords -- with the two
// pending vectors carried from one partition into the next (nothing is drained between runs).  In C++ a partition cost ~100
// scalar instructions before its first codeword and a drain behind its last one.
//   DEINT 0: element el + lane * step of the accumulators (el includes the channel's base); 2: the interleaved vector of two
//   channels: element a = el + lane goes to (a & 1) * half + (a >> 1).
// n: partitions to visit; on return the ones not visited yet (the one a return code names included).  Returns 0: all
// visited; 1: a codeword of partition (first + visited) needs the tree -- `todo` of its codewords are left, that one
// included; 2: a codeword runs past the end of the packet; 3: that partition is not one for this loop (single-entry book,
// a dimension that does not divide the partition or is not a multiple of the channel count): not started.
template <int DEINT>
__device__ inline __attribute__((always_inline)) uint32_t lw_ent_range(LwEntReader &r, const LW_K LwEntRun *runs, const LW_K uint32_t *lut,
		const LW_K float *vq, uint32_t pass, uint32_t &n, uint32_t cls_at, uint32_t el, uint32_t psize, uint32_t half, uint32_t acc_base,
		uint32_t dump, uint32_t &todo)
{
	uint32_t st, t0, t1, e, ph = 0, neg = 0;
	const uint32_t lane = LW_ENT_LANE();
	const uint32_t lutlo = (uint32_t)(uintptr_t)lut, luthi = (uint32_t)((uintptr_t)lut >> 32);
	const uint32_t vqlo = (uint32_t)(uintptr_t)vq, vqhi = (uint32_t)((uintptr_t)vq >> 32);
	const uint32_t pass32 = pass * 32u, passmask = 0x100u << pass, dper = DEINT == 2 ? 2u : 1u, half4 = half * 4u;
	n = LW_ENT_SCALAR(n); // (wave-uniform like everything scalar here; the compiler keeps the loop-carried ones in vector registers)
	el = LW_ENT_SCALAR(el);
	cls_at = LW_ENT_SCALAR(cls_at);
	if (DEINT == 0)
		asm volatile(LW_ENT_RANGE_ENTER LW_ENT_RANGE_VISIT LW_ENT_RANGE_PLACE0 LW_ENT_RANGE_TAIL LW_ENT_RANGE_OPERANDS);
	else
		asm volatile(LW_ENT_RANGE_ENTER LW_ENT_RANGE_VISIT LW_ENT_RANGE_PLACE2 LW_ENT_RANGE_TAIL LW_ENT_RANGE_OPERANDS);
	todo = 0u - neg;
	return st;
}

// The same for PASS 0 of a single vector, class words included: every group of cpc partitions starts with its class word
// (an ordinary classbook whose digits are in the image), so the whole pass is one statement -- in C++ the class words cost a
// statement per group, each with its drain.  gc: partitions of the current group still covered by the class word read last
// (0 at a group's start).  Further return code 4: the class word at the head of the window needs the tree (nothing of it
// consumed).
template <int DEINT>
__device__ inline __attribute__((always_inline)) uint32_t lw_ent_range_cw(LwEntReader &r, const LW_K LwEntRun *runs, const LW_K uint32_t *lut,
		const LW_K float *vq, uint32_t &n, uint32_t cls_at, uint32_t el, uint32_t psize, uint32_t half, uint32_t acc_base, uint32_t dump,
		uint32_t &todo, uint32_t gc, const LW_K uint32_t *cblut, uint32_t cbmask, uint32_t cbbits, const LW_K uint16_t *digits, uint32_t cpc)
{
	uint32_t st, t0, t1, e, ph = 0, neg = 0;
	const uint32_t lane = LW_ENT_LANE();
	const uint32_t lutlo = (uint32_t)(uintptr_t)lut, luthi = (uint32_t)((uintptr_t)lut >> 32);
	const uint32_t vqlo = (uint32_t)(uintptr_t)vq, vqhi = (uint32_t)((uintptr_t)vq >> 32);
	const uint32_t pass32 = 0u, passmask = 0x100u, dper = DEINT == 2 ? 2u : 1u, half4 = half * 4u;
	const uint32_t q2 = 2u * (lane < cpc ? lane : cpc - 1u); // (the lanes beyond the last digit copy the last digit again)
	n = LW_ENT_SCALAR(n);
	el = LW_ENT_SCALAR(el);
	cls_at = LW_ENT_SCALAR(cls_at);
	gc = LW_ENT_SCALAR(gc);
	cbmask = LW_ENT_SCALAR(cbmask);
	cbbits = LW_ENT_SCALAR(cbbits);
	cpc = LW_ENT_SCALAR(cpc);
#define LW_ENT_RANGE_CW_OPERANDS                                                                                      \
	: [win] "+s"(r.win), [have] "+s"(r.have), [left] "+s"(r.left), [nxt] "+s"(r.nxt), [wo] "+s"(r.wo), [neg] "+s"(neg),  \
	  [n] "+s"(n), [el] "+s"(el), [ph] "+s"(ph), [gc] "+s"(gc), [e] "=&s"(e), [st] "=&s"(st), [t0] "=&s"(t0), [t1] "=&s"(t1) \
	: [w] "s"(r.w), [runs] "s"(runs), [lutlo] "s"(lutlo), [luthi] "s"(luthi), [vqlo] "s"(vqlo), [vqhi] "s"(vqhi),        \
	  [pass32] "s"(pass32), [passmask] "s"(passmask), [dper] "s"(dper), [psize] "s"(psize), [cls] "s"(cls_at),           \
	  [accbase] "s"(acc_base), [half4] "s"(half4), [lane] "v"(lane), [dump] "v"(dump),                                    \
	  [cblut] "s"(LW_ENT_SCALAR64((uintptr_t)cblut)), [cbmask] "s"(cbmask), [cbbits] "s"(cbbits),                          \
	  [digits] "s"(LW_ENT_SCALAR64((uintptr_t)digits)), [cpc] "s"(cpc), [q2] "v"(q2)                                       \
	: LW_ENT_ASM_CLOBBERS
	if (DEINT == 0)
		asm volatile(LW_ENT_RANGE_ENTER LW_ENT_RANGE_CLASSWORD LW_ENT_RANGE_VISIT LW_ENT_RANGE_PLACE0 LW_ENT_RANGE_TAIL LW_ENT_RANGE_CW_OPERANDS);
	else
		asm volatile(LW_ENT_RANGE_ENTER LW_ENT_RANGE_CLASSWORD LW_ENT_RANGE_VISIT LW_ENT_RANGE_PLACE2 LW_ENT_RANGE_TAIL LW_ENT_RANGE_CW_OPERANDS);
	todo = 0u - neg;
	return st;
}
#endif

// audio.rs:620-717 for `nch` vectors of `actual` elements.  Element e of vector j is out[map(j, e)]: for residue type 2 the
// ONE interleaved vector of ch * n/2 elements is written straight to its channel-major place (audio.rs:748-754: element i
// belongs to channel i % ch, bin i / ch) -- every element receives the same additions in the same order as in the
// reference's interleaved buffer.  `out` holds zeros on entry.  `cls` = scratch for nch * (parts + cpc) digits.
//
// The additions run BEHIND the decoder: a codeword's vector row is requested as soon as its entry number is known, and
// added after the next codeword's table look-up has been started (LwEntPend, one behind: the C++ form of the loop, which the
// host runs for every codeword and the device for the rare ones; lw_ent_run, the device's steady state, stays two behind)
// -- the look-up chain (window -> table -> length -> window) is the only thing a packet cannot overlap, everything else
// hides under it.  Every addition reads the accumulator, also the first one to an element (0.0f + e, as the reference's +=
// on its zeroed vector).  On the device all 64 lanes take part in every addition without masking: a lane beyond the
// codeword's dimension adds to its own dump slot behind the accumulators.
struct LwEntPend {
	LW_LV(uint32_t, at); // this lane's accumulator (LW_ENT_AT)
	LW_LV(float, v);
	LW_LV(uint32_t, dump); // device: this lane's dump slot
	uint32_t n;          // host: elements pending
	LW_HD void clear(LwEntAcc out, uint32_t dump_el)
	{
		n = 0;
		LW_ENT_LANES(d, 0u) {
			LW_L(dump) = LW_ENT_AT(out, dump_el + d);
			LW_L(at) = LW_L(dump);
			LW_L(v) = 0.0f;
		}
	}
	// (nothing is added twice: a flushed vector's place is taken by the dump slot)
	LW_HD void flush(LwEntAcc out)
	{
		LW_ENT_LANES(d, n) {
			LW_ENT_ACC(out, LW_L(at)) = LW_ENT_ACC(out, LW_L(at)) + LW_L(v);
#if defined(__HIP_DEVICE_COMPILE__)
			LW_L(at) = LW_L(dump);
#endif
		}
		n = 0;
	}
};

// accumulator index of element `el` of the (interleaved, DEINT != 0) vector
template <int DEINT>
LW_HD uint32_t lw_ent_place(uint32_t el, uint32_t deint, uint32_t half, uint64_t cmap, bool ident)
{
	if (DEINT == 0)
		return el;
	const uint32_t v = DEINT == 2 ? el & 1u : el % deint, q = DEINT == 2 ? el >> 1 : el / deint;
	return (ident ? v : LW_ENT_CMAP(cmap, v)) * half + q;
}

// What the partitions of one residue vector share
struct LwEntVec {
	const LW_K LwEntRun *runs; // [class][pass] of this residue
	LwEntAcc out;
	LwEntDigits cls;           // this vector's digits: [partition]
	uint32_t base;             // first element of partition 0 (types 0/1: incl. the channel's base; type 2: interleaved index)
	uint32_t psize, half, deint_ch;
	uint32_t room0;            // elements from the start of partition 0 to the end of the vector (types 1/2: audio.rs:604-608)
	uint64_t cmap;
	bool ident;
};

// One partition of one vector in one pass, from its codeword i0 on (the host's whole path; on the device what lw_ent_range
// hands back: the partitions with a rare book shape, the rest of a partition behind a code that needs the tree).
// false: the packet ends here (nothing left pending).
// DEINT: 0 = the vectors are channels (residue types 0 and 1); 2 = type 2 over two channels; -1 = type 2 over `deint_ch`
// channels.
template <int DEINT>
LW_HD bool lw_ent_partition(const LwEntTables &T, const LwEntVec &V, LwEntReader &r, LwEntPend &pend, uint32_t pass, uint32_t pc, uint32_t i0)
{
	LwEntAcc out = V.out;
	const uint32_t cv = LW_ENT_SCALAR(V.cls[pc]); // class | passes it uses << 8
	if (!((cv >> (8u + pass)) & 1u))
		return true;
	const LW_K LwEntRun &rd = V.runs[(cv & 0xffu) * 8u + pass];
	const uint32_t shape = rd.shape, count = rd.count, step = rd.step, adv = rd.adv;
	const uint32_t dims = (shape >> 8) & 0xffu, lut_bits = shape & 0xffu;
	const int32_t single = (int16_t)(shape >> 16);
	const LW_K uint32_t *lut = (const LW_K uint32_t *)((const LW_K char *)T.lut + rd.lut_off);
	const LW_K float *vq = (const LW_K float *)((const LW_K char *)T.vq + rd.vq_off);
	const LW_K int32_t *nodes = rd.nodes_off != 0xFFFFFFFFu ? T.nodes + rd.nodes_off : nullptr;
	// first element of the partition in its vector; audio.rs:587-618 (the whole partition lies inside the vector, dims
	// divides the partition size): type 0: element d of codeword i at i + d * step; types 1/2: at i * dims + d
	const uint32_t el0 = V.base + pc * V.psize + i0 * adv;
	const uint32_t dper = DEINT == 2 ? 2u : DEINT ? V.deint_ch : 1u;
	if (i0 >= count)
		return true;
	if ((rd.fast >> dper) & 1u) {
		// the usual run: a lane's element moves by a constant from one codeword to the next (an interleaved vector: its
		// channel stays, its bin moves by dims / channels)
		const uint32_t lut_mask = (1u << lut_bits) - 1u;
		LW_LV(uint32_t, at);
		LW_LV(uint32_t, inc);
		LW_LV(uint32_t, row);
#if defined(__HIP_DEVICE_COMPILE__)
		// (the row's address is vector work -- one 24-bit multiply-add per lane on the table entry as it is, length bits and
		// all: the scalar unit, which paces this kernel, is left alone)
		const uint32_t vdims4 = dims * 4u;
#define LW_ENT_ROW(idx) (*(const LW_K float *)((const LW_K char *)vq + lw_ent_mad24((idx), vdims4, row)))
#else
#define LW_ENT_ROW(idx) vq[((idx) & 0xffffffu) * dims + row[d]]
#endif
		LW_ENT_LANES(d, dims) {
			const bool on = d < dims;
			LW_L(at) = LW_ENT_AT(out, on ? lw_ent_place<DEINT>(el0 + d * step, V.deint_ch, V.half, V.cmap, V.ident) : T.res_floats + d);
			LW_L(inc) = on ? (adv / dper) * (LW_ENT_AT(out, 1u) - LW_ENT_AT(out, 0u)) : 0u;
			LW_L(row) = on ? d * LW_ENT_ROW_UNIT : 0u;
		}
		uint32_t todo = count - i0;
		while (todo) {
			uint32_t e;
#if defined(__HIP_DEVICE_COMPILE__)
			// the steady state by hand (lw_ent_run, which starts and ends with nothing pending): it comes back for what is
			// rare -- a code beyond the two table levels, the end of the packet
			pend.flush(out);
			const uint32_t before = todo;
			const uint32_t st = lw_ent_run(r, lut, vq, lut_mask, lut_bits, todo, e, at, inc, row, vdims4, pend.dump);
			at += inc * (before - todo);
			if (st == 0u)
				break;
			if (st == 2u) {
				r.left = 0;
				return false;
			}
#else
			e = r.probe(lut, lut_mask); // this codeword's table look-up is under way ...
			pend.flush(out);            // ... while the previous codeword's vector is added
#endif
			uint32_t idx;
			if (!r.finish(lut, lut_bits, nodes, e, idx))
				return false;
			LW_ENT_LANES(d, dims) {
				pend.LW_L(at) = LW_L(at);
				pend.LW_L(v) = LW_ENT_ROW(idx);
				LW_L(at) += LW_L(inc);
			}
			pend.n = dims;
			todo--;
		}
		return true;
	}
	// single-entry / empty books, interleaved vectors whose channel count does not divide the dimension, books whose
	// dimension does not divide the partition size: their last codeword of a partition reaches into the next partition,
	// and one that would reach past the end of the vector is read and dropped, which ends the partition (audio.rs:600-612)
	const uint32_t room = V.room0 - pc * V.psize;
	const LwEntBookRegs cb = {lut, vq, nodes, (1u << lut_bits) - 1u, lut_bits, dims, single};
	uint32_t el = el0;
	for (uint32_t i = i0; i < count; i++, el += adv) {
		uint32_t idx;
		if (!r.code(cb, idx)) {
			pend.flush(out);
			return false;
		}
		pend.flush(out);
		if (adv != 1u && i * dims + dims > room)
			break;
		LW_ENT_LANES(d, dims) {
			const bool on = d < dims;
			pend.LW_L(at) = LW_ENT_AT(out, on ? lw_ent_place<DEINT>(el + d * step, V.deint_ch, V.half, V.cmap, V.ident) : T.res_floats + d);
			pend.LW_L(v) = vq[idx * dims + (on ? d : 0u)];
		}
		pend.n = dims;
	}
	return true;
}

// The class words of pass 0 when lw_ent_partitions reads them itself (device: an ordinary classbook whose digits are in the image)
struct LwEntCw {
	LwEntBookRegs book;
	const LW_K uint16_t *digits; // [entry][cpc]: class | passes of the class << 8
	uint32_t cpc;
};

// Partitions [p0, p1) of one vector in one pass.  false: the packet ends here.
// own_cw / cw (device, pass 0 of a single vector, p0 at a group's start): every group of cpc partitions starts with its class word,
// read here as well -- the whole pass is then one asm statement (lw_ent_range_cw).
template <int DEINT>
LW_HD bool lw_ent_partitions(const LwEntTables &T, const LwEntVec &V, LwEntReader &r, LwEntPend &pend, uint32_t pass, uint32_t p0, uint32_t p1,
		const bool own_cw, const LwEntCw cw) // (by value: a struct reached through a pointer stays in scratch memory, and what is
                                            //  loaded from there is divergent to the compiler)
{
	uint32_t n = p1 - p0;
	uint32_t covered = 0; // (cw) partitions from here on whose class word has been read by the C++ path below
	(void)covered;
	while (n) {
		uint32_t i0 = 0;
#if defined(__HIP_DEVICE_COMPILE__)
		if (DEINT == 0 || (DEINT == 2 && V.ident)) {
			// the device's steady state (the statements start and end with nothing pending)
			pend.flush(V.out);
			uint32_t todo, st;
			const uint32_t pc = p1 - n, cls_at = (uint32_t)(uintptr_t)(V.cls + pc), el = V.base + pc * V.psize;
			const uint32_t acc = (uint32_t)(uintptr_t)V.out;
			if (own_cw) {
				// gc: 0 at a group's start; inside a group (after a return below) the rest of the group
				const uint32_t gc = covered ? covered : (cw.cpc - (pc - p0) % cw.cpc) % cw.cpc;
				st = lw_ent_range_cw<(DEINT == 0 ? 0 : 2)>(r, V.runs, T.lut, T.vq, n, cls_at, el, V.psize, V.half, acc, pend.dump, todo, gc,
						cw.book.lut, cw.book.lut_mask, cw.book.lut_bits, cw.digits, cw.cpc);
				covered = 0;
			} else {
				st = lw_ent_range<(DEINT == 0 ? 0 : 2)>(r, V.runs, T.lut, T.vq, pass, n, cls_at, el, V.psize, V.half, acc, pend.dump, todo);
			}
			if (st == 0u)
				break;
			if (st == 2u) {
				r.left = 0;
				return false;
			}
			if (st == 4u) { // the class word through the tree (the range stopped at a group's start)
				uint32_t t;
				if (!r.code(cw.book, t))
					return false;
				const uint32_t at = p1 - n, cpc = cw.cpc;
				const LW_K uint16_t *dg = cw.digits + t * cpc;
				LW_ENT_LANES(q, cpc) {
					const uint32_t qq = q < cpc ? q : cpc - 1u;
					V.cls[at + qq] = dg[qq];
				}
				covered = cpc < n ? cpc : n;
				continue;
			}
			if (st == 1u) {
				const uint32_t cv = LW_ENT_SCALAR(V.cls[p1 - n]);
				i0 = V.runs[(cv & 0xffu) * 8u + pass].count - todo;
			}
			if (own_cw) { // (the partition below belongs to the current group: what is left of the group behind it stays covered)
				const uint32_t in_group = (p1 - n - p0) % cw.cpc;
				covered = cw.cpc - 1u - in_group < n - 1u ? cw.cpc - 1u - in_group : n - 1u;
			}
		}
#endif
		if (!lw_ent_partition<DEINT>(T, V, r, pend, pass, p1 - n, i0))
			return false;
		n--;
	}
	return true;
}

template <int DEINT>
LW_HD void lw_ent_residue(const LwEntTables &T, const LW_K LwEntResidue &rs, LwEntReader &r, uint32_t nch, uint32_t actual,
		uint32_t dnd, LwEntAcc out, uint32_t half, uint32_t deint_ch, LwEntDigits cls, uint64_t cmap, const bool general)
{
	const uint32_t begin = rs.begin < actual ? rs.begin : actual, end = rs.end < actual ? rs.end : actual;
	const uint32_t cpc = rs.cpc, psize = rs.psize;
	const uint32_t n_to_read = end - begin;
	const uint32_t parts = n_to_read / psize;
	if (n_to_read == 0)
		return;
	const uint32_t stride = parts + cpc;
	const uint32_t ncls = rs.classifications, digits_off = rs.digits_off, used_any = rs.used_any;
	const LwEntBookRegs classbook = lw_ent_book(T, rs.classbook);
	LwEntPend pend;
	pend.clear(out, T.res_floats);
	LwEntVec V;
	V.runs = T.runs + rs.runs_off;
	V.out = out;
	V.psize = psize;
	V.room0 = actual - begin;
	V.half = half;
	V.deint_ch = deint_ch;
	V.cmap = cmap;
	V.ident = true;
	if (general) // (a compile-time constant at both call sites: the one-submap kernel carries no channel map at all)
		for (uint32_t v = 0, nv = DEINT ? deint_ch : nch; v < nv; v++)
			V.ident &= LW_ENT_CMAP(cmap, v) == v;
	// (device) a single vector with an ordinary classbook whose digits are in the image: pass 0 reads its class words inside the
	// asm statement of its partitions (lw_ent_partitions, cw)
	LwEntCw cw = {classbook, T.digits + (digits_off != 0xFFFFFFFFu ? digits_off : 0u), cpc};
	bool own_cw = false;
#if defined(__HIP_DEVICE_COMPILE__)
	own_cw = nch == 1u && !(dnd & 1u) && classbook.single == -1 && digits_off != 0xFFFFFFFFu && (DEINT == 0 || DEINT == 2);
	if (DEINT == 2 && general) // (never instantiated: the interleaved vectors of several submaps go through DEINT = -1)
		own_cw = false;
#endif
	// (pass 0 always runs: it reads the class words even when no class of this residue has a book in any pass, audio.rs:664-676;
	// the bit reader is shared with the submaps behind this one)
	for (uint32_t pass = 0; pass < 8 && (pass == 0 || (used_any >> pass) != 0); pass++) {
		uint32_t pc = 0;
		while (pc < parts) {
			if (pass == 0 && !own_cw) {
				for (uint32_t j = 0; j < nch; j++) {
					if ((dnd >> j) & 1u)
						continue;
					uint32_t t;
					if (!r.code(classbook, t)) {
						pend.flush(out);
						return; // end of packet is normal (audio.rs:655-660)
					}
					LwEntDigits c = cls + j * stride + pc;
					if (digits_off != 0xFFFFFFFFu) {
						const LW_K uint16_t *dg = T.digits + digits_off + t * cpc;
						LW_ENT_LANES(q, cpc) { // (device: no masking -- the lanes beyond the last digit copy the last digit again)
							const uint32_t qq = q < cpc ? q : cpc - 1u;
							c[qq] = dg[qq];
						}
					} else {
						for (uint32_t q = cpc; q-- > 0;) {
							const uint32_t cl = t % ncls;
							c[q] = (uint16_t)(cl | ((uint32_t)rs.vals_used[cl] << 8));
							t /= ncls;
						}
					}
				}
			}
			// the partitions this group's class words cover; behind pass 0 nothing separates the groups, so a single
			// vector's partitions are one range to the end
			uint32_t pc_end = pc + cpc < parts ? pc + cpc : parts;
			if (nch == 1u) {
				if (pass != 0u || own_cw)
					pc_end = parts;
				if (!(dnd & 1u)) {
					V.cls = cls;
					V.base = (DEINT ? 0u : (V.ident ? 0u : LW_ENT_CMAP(cmap, 0u)) * half) + begin;
					if (!lw_ent_partitions<DEINT>(T, V, r, pend, pass, pc, pc_end, pass == 0u && own_cw, cw))
						return;
				}
				pc = pc_end;
				continue;
			}
			for (; pc < pc_end; pc++)
				for (uint32_t j = 0; j < nch; j++) {
					if ((dnd >> j) & 1u)
						continue;
					V.cls = cls + j * stride;
					V.base = (DEINT ? 0u : (V.ident ? j : LW_ENT_CMAP(cmap, j)) * half) + begin;
					if (!lw_ent_partitions<DEINT>(T, V, r, pend, pass, pc, pc + 1u, false, cw))
						return;
				}
		}
	}
	pend.flush(out);
}

// Floors and residues of one packet (what lw::entropy_decode does after the prologue).  floor_out [ch][fstride],
// res_out [ch][n/2] zero on entry (device: + LW_ENT_DUMP_FLOATS behind T.res_floats), ws = T.ws_bytes of scratch (4-byte aligned).
LW_HD void lw_ent_decode_packet(const LwEntTables &T, const LW_K uint32_t *words, uint32_t len_bytes, uint32_t start_bit,
		uint32_t mode, uint32_t n, uint16_t *floor_out, LwEntAcc res_out, LwEntPosts y, LwEntDigits cls, const bool general)
{
	LwEntReader r;
	r.init(words, len_bytes, start_bit);
	const LW_K LwEntMode &m = T.modes[mode];
	const uint32_t ch = T.ch, half = n >> 1;
	uint32_t no_residue = 0; // bit c: channel c has no residue
	// floor_decode, audio.rs:557-585
	for (uint32_t c = 0; c < ch; c++) {
		const LW_K LwEntFloor &fl = T.floors[LW_ENT_SCALAR(m.floor_of_ch[c])];
		uint16_t *rec = floor_out + c * T.fstride;
		if (!lw_ent_floor_decode(T, fl, r, y)) {
			rec[0] = LW_FLOOR_UNUSED;
			no_residue |= 1u << c;
		} else {
			lw_ent_floor_record(fl, y, rec);
		}
	}
	// audio.rs:948-955
	for (uint32_t i = 0; i < m.n_coupling; i++) {
		const uint32_t pair = (1u << m.mag[i]) | (1u << m.ang[i]);
		if ((no_residue & pair) != pair)
			no_residue &= ~pair;
	}
	const uint32_t all = (1u << ch) - 1u;
	if (!general) { // one submap holding every channel (T.general == 0): vector j = channel j
		const LW_K LwEntResidue &rs = T.residues[LW_ENT_SCALAR(m.submap_residue[0])];
		if (rs.type != 2) {
			lw_ent_residue<0>(T, rs, r, ch, half, no_residue, res_out, half, 0u, cls, 0, false);
			return;
		}
		// audio.rs:722-760: type 2 = one interleaved vector of ch * n/2 elements, decoded unless EVERY channel is marked
		if ((no_residue & all) == all)
			return;
		if (ch == 2)
			lw_ent_residue<2>(T, rs, r, 1u, ch * half, 0u, res_out, half, ch, cls, 0, false);
		else
			lw_ent_residue<-1>(T, rs, r, 1u, ch * half, 0u, res_out, half, ch, cls, 0, false);
		return;
	}
	// audio.rs:957-986: submap by submap, the vectors of a submap = its channels in channel order
	for (uint32_t sm = 0; sm < m.n_submaps; sm++) {
		uint32_t dnd = 0;
		uint64_t cmap = 0;
		uint32_t sub_ch = 0;
		bool any = false;
		for (uint32_t c = 0; c < ch; c++)
			if (m.mux[c] == sm) {
				dnd |= ((no_residue >> c) & 1u) << sub_ch;
				any |= !((no_residue >> c) & 1u);
				cmap |= (uint64_t)c << (4u * sub_ch);
				sub_ch++;
			}
		if (sub_ch == 0)
			continue;
		const LW_K LwEntResidue &rs = T.residues[LW_ENT_SCALAR(m.submap_residue[sm])];
		if (rs.type != 2)
			lw_ent_residue<0>(T, rs, r, sub_ch, half, dnd, res_out, half, 0u, cls, cmap, true);
		else if (any) // audio.rs:722-760: type 2 = one interleaved vector of sub_ch * n/2 elements, decoded unless EVERY channel is marked
			lw_ent_residue<-1>(T, rs, r, 1u, sub_ch * half, 0u, res_out, half, sub_ch, cls, cmap, true);
	}
}
