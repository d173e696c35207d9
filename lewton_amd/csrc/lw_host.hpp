// Host-side model of a Vorbis stream for the MI355X decode path: bit reader, Huffman decoder,
// header structures.  Product code (C++17); independent of oracle/.
//
// Reference behaviour restated here (paths relative to RustAudio/lewton 0.10.2):
//   bit reader      src/bitpacking.rs:93-161, :285-300
//   Huffman         src/huffman_tree.rs:183-221 (validation), :362-381 (walk); spec 3.2.1
//   header model    src/header.rs:188-211, :363-481
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace lw {

// ---- status codes shared with include/lewton_amd.h -------------------------------------------
enum : int {
	OK = 0,
	AUDIO_END_OF_PACKET = 1,
	AUDIO_BAD_FORMAT = 2,
	AUDIO_IS_HEADER = 3,
	AUDIO_BUFFER_NOT_ADDRESSABLE = 4,
	HDR_END_OF_PACKET = 16,
	HDR_NOT_VORBIS = 17,
	HDR_UNSUPPORTED_VERSION = 18,
	HDR_BAD_FORMAT = 19,
	HDR_BAD_TYPE = 20,
	HDR_IS_AUDIO = 21,
	HDR_UTF8 = 22,
	HDR_BUFFER_NOT_ADDRESSABLE = 23,
	ERR_NULL_ARG = 32,
	ERR_DEVICE = 33,
	ERR_CAPACITY = 34,
	ERR_STATE_MISMATCH = 35
};

inline unsigned ilog(uint64_t v)
{
	return v ? 64u - (unsigned)__builtin_clzll(v) : 0u;
}

// LSb-first bit reader over a packet.  A fixed-width read that does not fit fails without
// consuming anything; a zero-width read returns 0 (bitpacking.rs:291-297).
struct BitReader {
	const uint8_t *d = nullptr;
	uint64_t nbits = 0, pos = 0;

	BitReader(const uint8_t *data, size_t len) : d(data), nbits((uint64_t)len * 8), pos(0) {}

	uint64_t remaining() const { return nbits - pos; }

	// up to 57 bits starting at pos, zero padded past the end
	uint64_t window() const
	{
		const uint64_t byte = pos >> 3;
		const uint64_t total = nbits >> 3; // nbits is a whole number of bytes
		uint64_t w = 0;
		if (byte + 8 <= total) { // everywhere but in the last 7 bytes of the packet: one unaligned load (little-endian host)
			std::memcpy(&w, d + byte, 8);
			return w >> (pos & 7);
		}
		const unsigned navail = (unsigned)(total > byte ? total - byte : 0);
		for (unsigned i = 0; i < navail; i++)
			w |= (uint64_t)d[byte + i] << (8 * i);
		return w >> (pos & 7);
	}

	bool read(unsigned n, uint32_t &v)
	{
		if (n == 0) {
			v = 0;
			return true;
		}
		if (pos + n > nbits)
			return false;
		v = (uint32_t)(window() & (n >= 32 ? 0xffffffffull : ((1ull << n) - 1)));
		pos += n;
		return true;
	}

	bool read64(unsigned n, uint64_t &v)
	{
		if (n <= 32) {
			uint32_t t;
			if (!read(n, t))
				return false;
			v = t;
			return true;
		}
		if (pos + n > nbits)
			return false;
		uint32_t lo = 0, hi = 0;
		read(32, lo);
		read(n - 32, hi);
		v = ((uint64_t)hi << 32) | lo;
		return true;
	}

	bool flag(bool &b)
	{
		uint32_t v;
		if (!read(1, v))
			return false;
		b = v != 0;
		return true;
	}
};

// Huffman decoder for one codebook.  Codewords follow the Vorbis assignment rule (each entry, in
// entry order, takes the lowest free codeword of its length); decoding uses a LUT on the next
// lut_bits bits, a second-level table on up to SUB_BITS more for prefixes shared by longer codes, and a
// binary-tree walk for what is longer still.  Results are identical to a bit-by-bit walk, including at
// the end of a packet (a code that runs past the end consumes the rest and fails).
struct Huffman {
	static constexpr unsigned SUB_BITS = 8;
	static constexpr uint32_t LINK = 0x80000000u;
	// width of the first table level: the longest code, capped at 10 bits (12 for books with >= 2048 used entries, 11 from
	// 512): small books stay small in L1, large ones resolve most codes without the second, dependent load
	unsigned lut_bits = 10;
	// first 2^lut_bits entries: (len << 24) | symbol, or LINK | (sub_bits << 24) | offset of the prefix's sub-table in
	// this vector (entries of the same form, len = whole code length); len == 0 -> walk the tree
	std::vector<uint32_t> lut;
	std::vector<int32_t> nodes; // 2 ints per node: child for bit 0 / bit 1; >= 0 node index, < 0 = ~symbol, INT32_MIN = none
	int32_t single = -1;        // single-entry book: any one bit decodes this entry (huffman_tree.rs:202-217)
	uint32_t used = 0;
	bool has_lut = false;       // lut is populated and the book has more than one entry (CodeReader's fast path)

	enum BuildResult { VALID = 0, OVERSPECIFIED = 1, UNDERPOPULATED = 2, INVALID_SINGLE = 3 };
	BuildResult build(const uint8_t *lengths, size_t n);

	// table entry for the code at the start of window `w` (>= lut_bits + SUB_BITS bits, zero padded); len 0 = not in the tables
	inline uint32_t lookup(uint64_t w) const
	{
		uint32_t e = lut[w & ((1u << lut_bits) - 1)];
		if (e & LINK)
			e = lut[(e & 0xffffffu) + ((w >> lut_bits) & ((1u << ((e >> 24) & 0x7fu)) - 1))];
		return e;
	}

	inline bool decode(BitReader &r, uint32_t &sym) const
	{
		if (single >= 0) {
			uint32_t b;
			if (!r.read(1, b))
				return false;
			sym = (uint32_t)single;
			return true;
		}
		const uint64_t w = r.window();
		const uint64_t rem = r.remaining();
		if (!lut.empty()) {
			const uint32_t e = lookup(w);
			const unsigned len = e >> 24;
			if (len) {
				if (len > rem) {
					r.pos = r.nbits;
					return false;
				}
				r.pos += len;
				sym = e & 0xffffffu;
				return true;
			}
		}
		// long code (or no LUT): walk
		if (nodes.empty()) {
			r.pos = r.nbits; // empty book: the reference panics; treat as end of packet
			return false;
		}
		int32_t node = 0;
		uint64_t p = r.pos;
		for (;;) {
			if (p >= r.nbits) {
				r.pos = r.nbits;
				return false;
			}
			const unsigned bit = (r.d[p >> 3] >> (p & 7)) & 1u;
			p++;
			const int32_t c = nodes[2 * node + bit];
			if (c == INT32_MIN) {
				r.pos = r.nbits;
				return false;
			}
			if (c < 0) {
				sym = (uint32_t)~c;
				r.pos = p;
				return true;
			}
			node = c;
		}
	}
};

struct Codebook {
	uint16_t dims = 0;
	uint32_t entries = 0;
	bool has_vq = false;
	std::vector<float> vq; // entries * dims (header.rs:495-531)
	Huffman huff;
};

struct Floor1 {
	uint8_t multiplier = 1;
	std::vector<uint8_t> partition_class;
	uint8_t class_dim[16] = {0}, class_sub[16] = {0}, class_master[16] = {0};
	int16_t sub_books[16][8];
	std::vector<uint32_t> x_list;
	// derived, header-only (audio.rs:253-292 evaluated once instead of per packet)
	std::vector<uint16_t> lo_idx, hi_idx; // per post (header order), valid for i >= 2
	std::vector<uint32_t> dx;             // x_list[i] - x_list[lo_idx[i]]
	std::vector<uint64_t> adx_magic;      // 2^64 / (x_list[hi] - x_list[lo]) + 1: n / adx == (n * magic) >> 64 for every u32 n
	std::vector<uint16_t> sorted_idx;     // floor1_x_list_sorted[i].0 (header.rs:887-889)
	std::vector<uint32_t> sorted_x;       // floor1_x_list_sorted[i].1
	uint32_t range() const
	{
		static const uint32_t r[4] = {256, 128, 86, 64};
		return r[multiplier - 1];
	}
};

struct Floor0 {
	uint8_t order = 0, amp_bits = 0, amp_offset = 0, n_books = 0;
	uint8_t book_list[16] = {0};
	std::vector<float> bark_cos_omega[2];
};

struct Floor {
	int type = 1;
	Floor0 f0;
	Floor1 f1;
};

// Codeword reader of the residue loops: the bit window lives in registers across codewords and is topped up without
// a branch (w |= next 8 bytes << avail; the bytes already covered are OR-ed again with themselves).  The fast path runs
// while the 8 bytes at `byte` are inside the packet, where a table hit (<= 12 + SUB_BITS bits) cannot run into
// the end; everything else (longer codes, the tail of the packet, single-entry books) goes through Huffman::decode,
// whose results this reader reproduces exactly.
struct CodeReader {
	BitReader &r;
	const uint64_t total; // bytes in the packet
	uint64_t w = 0, byte = 0; // window (bit 0 = next unread bit) and the offset of the first byte not yet counted in avail
	unsigned avail = 0;       // bits of w accounted for: the read position is 8 * byte - avail
	bool live = false;        // w / byte / avail are valid (else r.pos is the read position)
	explicit CodeReader(BitReader &r_) : r(r_), total(r_.nbits >> 3) {}
	inline bool next(const Huffman &h, uint32_t &sym)
	{
		if (h.has_lut) {
			if (!live) {
				const uint64_t b = r.pos >> 3;
				if (b + 16 <= total) {
					const unsigned drop = (unsigned)(r.pos & 7);
					uint64_t x;
					std::memcpy(&x, r.d + b, 8);
					w = x >> drop;
					avail = 56 - drop;
					byte = b + 7;
					live = true;
				}
			}
			if (live) {
				if (byte + 8 <= total) {
					uint64_t x;
					std::memcpy(&x, r.d + byte, 8);
					w |= x << avail;
					byte += (63 - avail) >> 3;
					avail |= 56;
					const uint32_t e = h.lookup(w);
					const unsigned len = e >> 24;
					if (len) {
						w >>= len;
						avail -= len;
						sym = e & 0xffffffu;
						return true;
					}
				}
			}
		}
		sync();
		return h.decode(r, sym);
	}
	void sync()
	{
		if (live)
			r.pos = byte * 8 - avail;
		live = false;
	}
};

struct ResidueBook {
	uint8_t vals_used = 0;
	uint8_t val_i[8] = {0};
};

struct Residue {
	uint8_t type = 0;
	uint32_t begin = 0, end = 0, partition_size = 1;
	uint8_t classifications = 1, classbook = 0;
	std::vector<ResidueBook> books;
	// derived: the classification digits of every classbook entry ([entry][classbook.dims], audio.rs:662-668 evaluated at
	// setup time); empty when the table would be large, then the digits are computed per codeword
	std::vector<uint8_t> class_digits;
};

struct Mapping {
	std::vector<uint8_t> mag, ang;
	std::vector<uint8_t> mux;
	std::vector<uint8_t> submap_floor, submap_residue;
};

struct Mode {
	bool blockflag = false;
	uint8_t mapping = 0;
};

// CachedBlocksizeDerived, header_cached.rs:19-110
struct BlocksizeTables {
	std::vector<float> A, B, C, window;
	std::vector<uint32_t> bitrev;
	void init(uint8_t bs);
};

struct Ident {
	uint8_t channels = 0;
	uint32_t sample_rate = 0;
	int32_t br_max = 0, br_nom = 0, br_min = 0;
	uint8_t bs0 = 0, bs1 = 0;
	BlocksizeTables tab[2];
};

struct Setup {
	std::vector<Codebook> codebooks;
	std::vector<Floor> floors;
	std::vector<Residue> residues;
	std::vector<Mapping> mappings;
	std::vector<Mode> modes;
};

struct Comment {
	std::string vendor;
	std::vector<std::pair<std::string, std::string>> list;
};

// header.rs:221, :309, :1082.  Return nullptr and set err on failure.
std::unique_ptr<Ident> read_header_ident(const uint8_t *pkt, size_t len, int &err);
std::unique_ptr<Comment> read_header_comment(const uint8_t *pkt, size_t len, int &err);
std::unique_ptr<Setup> read_header_setup(const uint8_t *pkt, size_t len, uint8_t channels, uint8_t bs0, uint8_t bs1,
		int &err);

uint32_t lookup1_values(uint32_t entries, uint16_t dims); // header.rs:616
float float32_unpack(uint32_t v);                         // bitpacking.rs:304

// Window geometry of one packet, audio.rs:1056-1073 (= :889-906)
struct WindowInfo {
	uint32_t n, left_start, right_start, right_end;
	bool left_use_bs1;
};
WindowInfo window_info(const Ident &id, bool blockflag, bool prev_flag, bool next_flag);

} // namespace lw
