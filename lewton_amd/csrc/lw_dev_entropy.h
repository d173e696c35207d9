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
// dimension divides the partition size, at most 8 channels.
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
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) float *LwEntAcc;
typedef __attribute__((address_space(3))) uint32_t *LwEntPosts;
typedef __attribute__((address_space(3))) uint8_t *LwEntDigits;
#define LW_ENT_EACH(d, cnt)                                                                                          \
	for (uint32_t d = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), once_ = 1; once_ && d < (cnt); once_ = 0)
#else
typedef float *LwEntAcc;
typedef uint32_t *LwEntPosts;
typedef uint8_t *LwEntDigits;
#define LW_ENT_EACH(d, cnt) for (uint32_t d = 0; d < (cnt); d++)
#endif

#define LW_ENT_MAX_CH 8
#define LW_ENT_MAX_CLASSES 64
#define LW_ENT_MAX_COUPLING 16
#define LW_ENT_LINK 0x80000000u

struct alignas(16) LwEntBook { // 16 bytes, read with one load
	uint32_t lut_off;  // first-level table (2^lut_bits entries) in the image's u32 pool; sub-tables follow at offsets relative to it
	uint32_t vq_off;   // entries * dims floats in the image's f32 pool
	uint8_t lut_bits, dims;
	int16_t single;    // >= 0: single-entry book, any one bit decodes this entry (huffman_tree.rs:202-217); -2: empty book
	uint32_t nodes_off; // binary tree (2 ints per node) in the image's i32 pool for codes beyond the two table levels; ~0u: none
};

struct LwEntFloor {
	uint8_t multiplier, range_bits, n_part, F;
	uint32_t range;
	uint8_t partition_class[32];
	uint8_t class_dim[16], class_sub[16], class_master[16];
	int16_t sub_books[16][8];
	uint8_t lo_idx[LW_MAX_POSTS], hi_idx[LW_MAX_POSTS], sorted_idx[LW_MAX_POSTS];
	uint8_t pad0;
	uint32_t dx[LW_MAX_POSTS];
	uint32_t pad1;
	uint64_t adx_magic[LW_MAX_POSTS];
};

struct LwEntResidue {
	uint8_t type, classifications, classbook, cpc;
	uint32_t begin, end, psize;
	uint32_t digits_off; // u8 [classbook entries][cpc] in the image's byte pool, or 0xFFFFFFFF: digits by division
	uint8_t vals_used[LW_ENT_MAX_CLASSES];
	uint8_t val_i[LW_ENT_MAX_CLASSES][8];
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
	const LwEntBook *books;
	const LwEntFloor *floors;
	const LwEntResidue *residues;
	const LwEntMode *modes;
	const uint32_t *lut;
	const float *vq;
	const uint8_t *bytes;
	const int32_t *nodes;
	uint32_t ch, fstride;
	uint32_t ws_bytes;  // per-packet scratch: posts (4 * LW_MAX_POSTS rounded up) + classification digits
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

#define LW_ENT_POSTS_BYTES ((4u * LW_MAX_POSTS + 15u) & ~15u)

// A codebook as the decode loops hold it: everything in registers, fetched with ONE load of the 16-byte table entry.  (Read
// field by field through a pointer, every codeword paid 3-4 dependent table accesses before its own look-up: the compiler
// may not keep byte-typed fields in registers across the residue stores.)
struct LwEntBookRegs {
	const uint32_t *lut;
	const float *vq;
	const int32_t *nodes; // tree for the (rare) codes longer than the two table levels, or null
	uint32_t lut_mask, lut_bits, dims;
	int32_t single;
};

LW_HD LwEntBookRegs lw_ent_book(const LwEntTables &T, uint32_t bi)
{
	const LwEntBook b = T.books[bi];
	LwEntBookRegs r;
	r.lut = T.lut + b.lut_off;
	r.vq = T.vq + b.vq_off;
	r.lut_bits = b.lut_bits;
	r.lut_mask = (1u << b.lut_bits) - 1u;
	r.dims = b.dims;
	r.single = b.single;
	r.nodes = b.nodes_off != 0xFFFFFFFFu ? T.nodes + b.nodes_off : nullptr;
	return r;
}

// LSb-first reader with the bit window in registers: `win` holds the next `have` bits (>= 32 after peek()), `nxt` the word
// after them, requested one refill ahead -- the packet bytes are read sequentially whatever the code lengths, so their
// loads are never on a codeword's dependency chain.  The pool keeps >= 3 zero words behind every packet.
struct LwEntReader {
	const uint32_t *w;
	uint32_t nbits, pos;
	uint64_t win;
	uint32_t have, wi, nxt;

	LW_HD void init(const uint32_t *words, uint32_t len_bytes, uint32_t start_bit)
	{
		w = words;
		nbits = len_bytes * 8u;
		pos = start_bit;
		const uint32_t i = pos >> 5, s = pos & 31u;
		win = (uint64_t)(w[i] >> s);
		have = 32u - s;
		nxt = w[i + 1];
		wi = i + 2;
	}
	// the next 32 bits, zero past the end of the packet
	LW_HD uint32_t peek()
	{
		if (have < 32u) {
			win |= (uint64_t)nxt << have;
			have += 32u;
			nxt = w[wi];
			wi++;
		}
		return (uint32_t)win;
	}
	LW_HD void skip(uint32_t n) // n <= 32, after peek()
	{
		win >>= n;
		have -= n;
		pos += n;
	}
	// bitpacking.rs:291-297: a fixed-width read that does not fit fails without consuming anything; n <= 32
	LW_HD bool read(uint32_t n, uint32_t &v)
	{
		if (n == 0) {
			v = 0;
			return true;
		}
		if (pos + n > nbits)
			return false;
		const uint32_t x = peek();
		v = n >= 32 ? x : (x & ((1u << n) - 1u));
		skip(n);
		return true;
	}
	// huffman_tree.rs:362-381 through the two table levels: a code that runs past the end consumes the rest and fails
	// (after that every read fails on its bounds check; the window is not looked at again)
	// the same in two halves, so that a caller can put other work between the table load and its first use: probe() starts
	// the first-level look-up of an ordinary book (any value for the special ones), finish() does everything else
	LW_HD uint32_t probe(const LwEntBookRegs &b)
	{
		if (b.single != -1)
			return 0;
		return b.lut[peek() & b.lut_mask];
	}
	LW_HD bool finish(const LwEntBookRegs &b, uint32_t e, uint32_t &sym)
	{
		if (b.single != -1)
			return code(b, sym);
		if (e & LW_ENT_LINK)
			e = b.lut[(e & 0xffffffu) + (((uint32_t)win >> b.lut_bits) & ((1u << ((e >> 24) & 0x7fu)) - 1u))];
		const uint32_t len = e >> 24;
		if (len == 0)
			return walk(b, sym);
		if (len > nbits - pos) {
			pos = nbits;
			return false;
		}
		skip(len);
		sym = e & 0xffffffu;
		return true;
	}
	LW_HD bool code(const LwEntBookRegs &b, uint32_t &sym)
	{
		if (b.single >= 0) {
			if (pos + 1 > nbits)
				return false;
			(void)peek();
			skip(1);
			sym = (uint32_t)b.single;
			return true;
		}
		if (b.single == -2) { // empty book: the reference panics; like the host stage, the packet ends here
			pos = nbits;
			return false;
		}
		const uint32_t x = peek();
		uint32_t e = b.lut[x & b.lut_mask];
		if (e & LW_ENT_LINK)
			e = b.lut[(e & 0xffffffu) + ((x >> b.lut_bits) & ((1u << ((e >> 24) & 0x7fu)) - 1u))];
		const uint32_t len = e >> 24;
		if (len == 0)
			return walk(b, sym);
		if (len > nbits - pos) {
			pos = nbits;
			return false;
		}
		skip(len);
		sym = e & 0xffffffu;
		return true;
	}
	// a code beyond the table levels (19+ bits: one in 2^18 codewords of a real encoder's books): bit by bit through the tree,
	// straight from the packet words
	LW_HD bool walk(const LwEntBookRegs &b, uint32_t &sym)
	{
		if (!b.nodes) {
			pos = nbits;
			return false;
		}
		int32_t node = 0;
		uint32_t p = pos;
		for (;;) {
			if (p >= nbits) {
				pos = nbits;
				return false;
			}
			const uint32_t bit = (w[p >> 5] >> (p & 31u)) & 1u;
			p++;
			const int32_t c = b.nodes[2 * node + (int32_t)bit];
			if (c == (int32_t)0x80000000) {
				pos = nbits;
				return false;
			}
			if (c < 0) {
				sym = (uint32_t)~c;
				uint32_t left = p - pos;
				while (left) { // (more than 32 bits are possible)
					const uint32_t s = left < 32u ? left : 32u;
					(void)peek();
					skip(s);
					left -= s;
				}
				return true;
			}
			node = c;
		}
	}
};

// audio.rs:215-251; y = the packet's scratch.  false = unused floor (FloorSpecialCase::Unused)
LW_HD bool lw_ent_floor_decode(const LwEntTables &T, const LwEntFloor &fl, LwEntReader &r, LwEntPosts y)
{
	uint32_t nonzero;
	if (!r.read(1, nonzero) || !nonzero)
		return false;
	uint32_t k = 0, v;
	if (!r.read(fl.range_bits, v))
		return false;
	y[k++] = v;
	if (!r.read(fl.range_bits, v))
		return false;
	y[k++] = v;
	for (uint32_t p = 0; p < fl.n_part; p++) {
		const uint32_t c = fl.partition_class[p];
		const uint32_t cdim = fl.class_dim[c], cbits = fl.class_sub[c];
		const uint32_t csub = (1u << cbits) - 1u;
		uint32_t cval = 0;
		if (cbits && !r.code(lw_ent_book(T, fl.class_master[c]), cval))
			return false;
		for (uint32_t d = 0; d < cdim; d++) {
			const int book = fl.sub_books[c][cval & csub];
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
LW_HD uint32_t lw_ent_render_point(uint32_t y0, uint32_t y1, uint32_t dx, uint64_t adx_magic)
{
	const int32_t dy = (int32_t)(y1 - y0);
	const uint32_t ady = dy < 0 ? 0u - (uint32_t)dy : (uint32_t)dy;
	const uint32_t num = ady * dx;
	// (num * magic) >> 64 in 64-bit pieces: adx >= 2 (a post lies strictly between its neighbours), so magic <= 2^63 + 1
	const uint64_t hi = (uint64_t)num * (adx_magic >> 32), lo = (uint64_t)num * (adx_magic & 0xffffffffu);
	const uint32_t off = (uint32_t)((hi + (lo >> 32)) >> 32);
	return dy < 0 ? y0 - off : y0 + off;
}

// audio.rs:391-435 -> device record: per post in ascending-x order, (final_y * multiplier) | active flag.  y is updated in
// place (a post's neighbours precede it in header order).
LW_HD void lw_ent_floor_record(const LwEntFloor &fl, LwEntPosts y, uint16_t *rec)
{
	const uint32_t F = fl.F, range = fl.range;
	uint32_t act0 = 3u, act1 = 0u, act2 = 0u; // step2 flags of posts 0-31, 32-63, 64
	for (uint32_t i = 2; i < F; i++) {
		const uint32_t lo = fl.lo_idx[i], hi = fl.hi_idx[i];
		const int32_t predicted = (int32_t)lw_ent_render_point(y[lo], y[hi], fl.dx[i], fl.adx_magic[i]);
		const int32_t val = (int32_t)y[i];
		const int32_t highroom = (int32_t)(range - (uint32_t)predicted);
		const int32_t lowroom = predicted;
		const int32_t room = (int32_t)((uint32_t)(highroom < lowroom ? highroom : lowroom) * 2u);
		if (val > 0) {
			const uint32_t idx[3] = {lo, hi, i};
			for (int q = 0; q < 3; q++) {
				const uint32_t j = idx[q];
				if (j < 32)
					act0 |= 1u << j;
				else if (j < 64)
					act1 |= 1u << (j - 32);
				else
					act2 |= 1u;
			}
			uint32_t fy;
			if (val >= room) {
				fy = highroom > lowroom ? (uint32_t)predicted + (uint32_t)val - (uint32_t)lowroom
				                        : (uint32_t)predicted - (uint32_t)val + (uint32_t)highroom - 1u;
			} else {
				const int32_t t = (val % 2 == 1) ? (int32_t)(0u - (uint32_t)val - 1u) : val;
				fy = (uint32_t)predicted + (uint32_t)(t >> 1);
			}
			y[i] = fy;
		} else {
			y[i] = (uint32_t)predicted;
		}
	}
	for (uint32_t sidx = 0; sidx < F; sidx++) {
		const uint32_t i = fl.sorted_idx[sidx];
		const uint32_t fy = y[i] < range - 1u ? y[i] : range - 1u; // :431-433
		const uint32_t on = i < 32 ? (act0 >> i) & 1u : i < 64 ? (act1 >> (i - 32)) & 1u : act2 & 1u;
		rec[sidx] = (uint16_t)(((fy * fl.multiplier) & 0xffu) | (on ? LW_POST_ACTIVE : 0u));
	}
}

// audio.rs:620-717 for `nch` vectors of `actual` elements.  Element e of vector j is out[map(j, e)]: for residue type 2 the
// ONE interleaved vector of ch * n/2 elements is written straight to its channel-major place (audio.rs:748-754: element i
// belongs to channel i % ch, bin i / ch) -- every element receives the same additions in the same order as in the
// reference's interleaved buffer.  `out` holds zeros on entry.  `cls` = scratch for nch * (parts + cpc) digits.
//
// The additions run ONE CODEWORD BEHIND the decoder: a codeword's vector row is requested as soon as its entry number is
// known, and added (LwEntPending::flush) after the NEXT codeword's table look-up has been started -- the look-up chain
// (window -> table -> length -> window) is the only thing a packet cannot overlap, everything else hides under it.
// The first pass that touches a partition finds zeros there (a pass adds to an element at most once): it stores
// 0.0f + e without reading the accumulator.
struct LwEntPending {
	const float *row; // the codeword's VQ row, or null: nothing pending
	uint32_t dims, base, step, deint, half;
	uint64_t cmap;    // channel of vector j in byte j (a submap's vectors are a subset of the packet's channels)
	bool ident;       // vector j is channel j (the usual one-submap case)
	bool first;
#if defined(__HIP_DEVICE_COMPILE__)
	float rowv; // lane d holds row[d]: requested when the codeword was decoded
#endif
	LW_HD void stash(const float *r, uint32_t n_dims, uint32_t base_el, uint32_t step_, uint32_t deint_, uint32_t half_, bool first_)
	{
		row = r;
		dims = n_dims;
		base = base_el;
		step = step_;
		deint = deint_;
		half = half_;
		first = first_;
#if defined(__HIP_DEVICE_COMPILE__)
		LW_ENT_EACH(d, n_dims)
			rowv = r[d];
#endif
	}
	// element d of the row goes to: type 0 base + d * step; types 1/2 base + d, de-interleaved for type 2
	LW_HD void flush(LwEntAcc out)
	{
		if (!row)
			return;
		LW_ENT_EACH(d, dims) {
#if defined(__HIP_DEVICE_COMPILE__)
			const float e = rowv;
#else
			const float e = row[d];
#endif
			uint32_t at = base + d * step;
			if (deint) { // type 2: element `at` of the interleaved vector belongs to the submap's vector at % deint, bin at / deint
				const uint32_t v = deint == 2 ? at & 1u : at % deint, q = deint == 2 ? at >> 1 : at / deint;
				at = (ident ? v : (uint32_t)((cmap >> (8u * v)) & 0xffu)) * half + q;
			}
			out[at] = (first ? 0.0f : out[at]) + e;
		}
		row = nullptr;
	}
};

LW_HD void lw_ent_residue(const LwEntTables &T, const LwEntResidue &rs, LwEntReader &r, uint32_t nch, uint32_t actual,
		const bool *dnd, LwEntAcc out, uint32_t half, uint32_t deint_ch, LwEntDigits cls, uint64_t cmap, const bool general)
{
	const uint32_t begin = rs.begin < actual ? rs.begin : actual, end = rs.end < actual ? rs.end : actual;
	const uint32_t cpc = rs.cpc, psize = rs.psize;
	const uint32_t n_to_read = end - begin;
	const uint32_t parts = n_to_read / psize;
	if (n_to_read == 0)
		return;
	const uint32_t stride = parts + cpc;
	const uint32_t ncls = rs.classifications, rtype = rs.type, digits_off = rs.digits_off;
	uint32_t used_any = 0; // passes some class of this residue uses at all
	for (uint32_t c = 0; c < ncls; c++)
		used_any |= rs.vals_used[c];
	const LwEntBookRegs classbook = lw_ent_book(T, rs.classbook);
	LwEntPending pend;
	pend.row = nullptr;
	pend.cmap = cmap;
	pend.ident = true;
	if (general) // (a compile-time constant at both call sites: the one-submap kernel carries no channel map at all)
		for (uint32_t v = 0, nv = deint_ch ? deint_ch : nch; v < nv; v++)
			pend.ident &= ((cmap >> (8u * v)) & 0xffu) == v;
	for (uint32_t pass = 0; pass < 8 && (used_any >> pass) != 0; pass++) {
		uint32_t pc = 0;
		while (pc < parts) {
			if (pass == 0) {
				for (uint32_t j = 0; j < nch; j++) {
					if (dnd[j])
						continue;
					uint32_t t;
					if (!r.code(classbook, t)) {
						pend.flush(out);
						return; // end of packet is normal (audio.rs:655-660)
					}
					LwEntDigits c = cls + j * stride + pc;
					if (digits_off != 0xFFFFFFFFu) {
						const uint8_t *dg = T.bytes + digits_off + t * cpc;
						LW_ENT_EACH(q, cpc)
							c[q] = dg[q];
					} else {
						for (uint32_t q = cpc; q-- > 0;) {
							c[q] = (uint8_t)(t % ncls);
							t /= ncls;
						}
					}
				}
			}
			for (uint32_t k = 0; k < cpc && pc < parts; k++, pc++) {
				for (uint32_t j = 0; j < nch; j++) {
					if (dnd[j])
						continue;
					const uint32_t cl = cls[j * stride + pc];
					const uint32_t vu = rs.vals_used[cl];
					if (!(vu & (1u << pass)))
						continue;
					const LwEntBookRegs cb = lw_ent_book(T, rs.val_i[cl][pass]);
					const uint32_t dims = cb.dims;
					const bool first = (vu & ((1u << pass) - 1u)) == 0;
					// audio.rs:587-618 (the whole partition lies inside the vector; dims divides the partition size):
					// type 0: psize / dims codewords, element d of codeword i at i + d * step; types 1/2: at i * dims + d
					const uint32_t step = rtype == 0 ? psize / dims : 1u;
					const uint32_t count = psize / dims, adv = rtype == 0 ? 1u : dims;
					uint32_t at = (deint_ch ? 0u : (pend.ident ? j : (uint32_t)((cmap >> (8u * j)) & 0xffu)) * half) + begin + pc * psize;
					for (uint32_t i = 0; i < count; i++, at += adv) {
						const uint32_t e = r.probe(cb); // this codeword's table look-up is under way ...
						pend.flush(out);                 // ... while the previous codeword's vector is added
						uint32_t idx;
						if (!r.finish(cb, e, idx))
							return;
						pend.stash(cb.vq + idx * dims, dims, at, step, deint_ch, half, first);
					}
				}
			}
		}
	}
	pend.flush(out);
}

// Floors and residues of one packet (what lw::entropy_decode does after the prologue).  floor_out [ch][fstride],
// res_out [ch][n/2] zero on entry, ws = T.ws_bytes of scratch (4-byte aligned).
LW_HD void lw_ent_decode_packet(const LwEntTables &T, const uint32_t *words, uint32_t len_bytes, uint32_t start_bit,
		uint32_t mode, uint32_t n, uint16_t *floor_out, LwEntAcc res_out, LwEntPosts y, LwEntDigits cls, const bool general)
{
	LwEntReader r;
	r.init(words, len_bytes, start_bit);
	const LwEntMode &m = T.modes[mode];
	const uint32_t ch = T.ch, half = n >> 1;
	bool no_residue[LW_ENT_MAX_CH];
	// floor_decode, audio.rs:557-585
	for (uint32_t c = 0; c < ch; c++) {
		const LwEntFloor &fl = T.floors[m.floor_of_ch[c]];
		uint16_t *rec = floor_out + c * T.fstride;
		if (!lw_ent_floor_decode(T, fl, r, y)) {
			rec[0] = LW_FLOOR_UNUSED;
			no_residue[c] = true;
		} else {
			lw_ent_floor_record(fl, y, rec);
			no_residue[c] = false;
		}
	}
	// audio.rs:948-955
	for (uint32_t i = 0; i < m.n_coupling; i++) {
		const uint32_t mg = m.mag[i], an = m.ang[i];
		if (!(no_residue[mg] && no_residue[an]))
			no_residue[mg] = no_residue[an] = false;
	}
	if (!general) { // one submap holding every channel (T.general == 0): vector j = channel j
		const LwEntResidue &rs = T.residues[m.submap_residue[0]];
		if (rs.type != 2) {
			lw_ent_residue(T, rs, r, ch, half, no_residue, res_out, half, 0u, cls, 0, false);
			return;
		}
		// audio.rs:722-760: type 2 = one interleaved vector of ch * n/2 elements, decoded unless EVERY channel is marked
		bool any = false;
		for (uint32_t c = 0; c < ch; c++)
			any |= !no_residue[c];
		if (!any)
			return;
		const bool one_dnd[1] = {false};
		lw_ent_residue(T, rs, r, 1u, ch * half, one_dnd, res_out, half, ch, cls, 0, false);
		return;
	}
	// audio.rs:957-986: submap by submap, the vectors of a submap = its channels in channel order
	for (uint32_t sm = 0; sm < m.n_submaps; sm++) {
		bool dnd[LW_ENT_MAX_CH];
		uint64_t cmap = 0;
		uint32_t sub_ch = 0;
		bool any = false;
		for (uint32_t c = 0; c < ch; c++)
			if (m.mux[c] == sm) {
				dnd[sub_ch] = no_residue[c];
				any |= !no_residue[c];
				cmap |= (uint64_t)c << (8u * sub_ch);
				sub_ch++;
			}
		if (sub_ch == 0)
			continue;
		const LwEntResidue &rs = T.residues[m.submap_residue[sm]];
		if (rs.type != 2) {
			lw_ent_residue(T, rs, r, sub_ch, half, dnd, res_out, half, 0u, cls, cmap, true);
		} else if (any) {
			// audio.rs:722-760: type 2 = one interleaved vector of sub_ch * n/2 elements, decoded unless EVERY channel is marked
			const bool one_dnd[1] = {false};
			lw_ent_residue(T, rs, r, 1u, sub_ch * half, one_dnd, res_out, half, sub_ch, cls, cmap, true);
		}
	}
}
