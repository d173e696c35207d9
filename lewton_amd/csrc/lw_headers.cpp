// Header parsing and derived host tables (product code).
//
// Restates, for the host side of the MI355X decode path:
//   src/header.rs:124-150 (capture pattern), :221-259 (ident), :309-355 (comment), :495-531 (VQ unpack),
//   :562-648 (lookup1_values), :673-768 (codebook), :771-918 (floor), :922-981 (residue),
//   :985-1057 (mapping), :1060-1076 (mode), :1082-1154 (setup);
//   src/header_cached.rs:34-110 (twiddles/window/bitrev), :129-158 (bark map);
//   src/huffman_tree.rs:183-221 (tree validity).
#include "lw_host.hpp"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>

namespace lw {

// ---------------------------------------------------------------------------------------------
// Huffman
// ---------------------------------------------------------------------------------------------
// Codeword assignment by free-node bookkeeping: avail[d] holds the (left-aligned) path of the one
// free node at depth d that lies to the right of everything assigned so far.  Taking the deepest free
// node of depth <= len and descending along zeros yields the lowest free codeword of that length,
// which is what inserting "at the leftmost free leaf" (huffman_tree.rs:66-123) produces.
Huffman::BuildResult Huffman::build(const uint8_t *lengths, size_t n)
{
	lut.clear();
	nodes.clear();
	single = -1;
	used = 0;
	has_lut = false;
	uint32_t avail[33];
	std::memset(avail, 0, sizeof(avail));
	std::vector<uint32_t> code(n, 0); // left-aligned path bits
	size_t last = 0;
	bool first = true;
	for (size_t i = 0; i < n; i++) {
		const unsigned len = lengths[i];
		if (len == 0)
			continue;
		if (len > 32)
			return OVERSPECIFIED;
		used++;
		last = i;
		if (first) {
			first = false;
			code[i] = 0;
			for (unsigned d = 1; d <= len; d++)
				avail[d] = 1u << (32 - d);
			continue;
		}
		unsigned z = len;
		while (z > 0 && !avail[z])
			z--;
		if (z == 0)
			return OVERSPECIFIED;
		const uint32_t c = avail[z];
		avail[z] = 0;
		code[i] = c;
		for (unsigned y = len; y > z; y--)
			avail[y] = c + (1u << (32 - y));
	}
	if (used == 1) {
		if (lengths[last] != 1)
			return INVALID_SINGLE;
		single = (int32_t)last;
		return VALID;
	}
	if (used > 1) {
		for (unsigned d = 1; d <= 32; d++)
			if (avail[d])
				return UNDERPOPULATED;
	}
	if (used == 0)
		return VALID; // decodes nothing; reading from it is a stream error

	// decode structures
	{
		unsigned maxlen = 0;
		for (size_t i = 0; i < n; i++)
			maxlen = std::max<unsigned>(maxlen, lengths[i]);
		lut_bits = std::min(maxlen, used >= 2048 ? 12u : used >= 512 ? 11u : 10u);
	}
	const unsigned LUT_BITS = lut_bits;
	lut.assign((size_t)1 << LUT_BITS, 0);
	nodes.assign(2, INT32_MIN);
	for (size_t i = 0; i < n; i++) {
		const unsigned len = lengths[i];
		if (!len)
			continue;
		// bits in stream order: b0 = top bit of code
		uint32_t lsb_first = 0;
		for (unsigned k = 0; k < len && k < LUT_BITS; k++)
			lsb_first |= ((code[i] >> (31 - k)) & 1u) << k;
		if (len <= LUT_BITS && i < (1u << 24)) {
			for (uint32_t hi = 0; hi < (1u << (LUT_BITS - len)); hi++)
				lut[lsb_first | (hi << len)] = ((uint32_t)len << 24) | (uint32_t)i;
		}
		int32_t node = 0;
		for (unsigned k = 0; k < len; k++) {
			const unsigned bit = (code[i] >> (31 - k)) & 1u;
			if (k + 1 == len) {
				nodes[2 * node + bit] = ~(int32_t)i;
			} else {
				int32_t c = nodes[2 * node + bit];
				if (c == INT32_MIN) {
					c = (int32_t)(nodes.size() / 2);
					nodes.push_back(INT32_MIN);
					nodes.push_back(INT32_MIN);
					nodes[2 * node + bit] = c;
				}
				node = c;
			}
		}
	}
	// second level: one sub-table per LUT_BITS-bit prefix that only long codes share, indexed by the following
	// min(longest code - LUT_BITS, SUB_BITS) bits; codes longer than that keep a zero entry (tree walk)
	std::vector<uint8_t> sub_bits((size_t)1 << LUT_BITS, 0);
	auto stream_bits = [&](size_t i) { // bit k of the result = k-th bit of entry i's code in stream order
		uint32_t v = 0;
		for (unsigned k = 0; k < lengths[i]; k++)
			v |= ((code[i] >> (31 - k)) & 1u) << k;
		return v;
	};
	for (size_t i = 0; i < n; i++)
		if (lengths[i] > LUT_BITS) {
			uint8_t &sb = sub_bits[stream_bits(i) & ((1u << LUT_BITS) - 1)];
			sb = std::max<uint8_t>(sb, (uint8_t)std::min<unsigned>(lengths[i] - LUT_BITS, SUB_BITS));
		}
	// table memory stays proportional to the book (a setup header must not buy megabytes of tables per codebook with a few
	// kilobytes of code lengths); prefixes that no longer fit are left to the tree walk
	const size_t cap = std::min<size_t>((size_t)1 << 24, ((size_t)1 << LUT_BITS) + 8 * (size_t)used);
	for (size_t p = 0; p < ((size_t)1 << LUT_BITS); p++)
		if (sub_bits[p] && lut.size() + ((size_t)1 << sub_bits[p]) <= cap) {
			lut[p] = LINK | ((uint32_t)sub_bits[p] << 24) | (uint32_t)lut.size();
			lut.resize(lut.size() + ((size_t)1 << sub_bits[p]), 0);
		}
	for (size_t i = 0; i < n && i < (1u << 24); i++) {
		const unsigned len = lengths[i];
		if (len <= LUT_BITS)
			continue;
		const uint32_t bits = stream_bits(i);
		const uint32_t link = lut[bits & ((1u << LUT_BITS) - 1)];
		const unsigned sb = (link >> 24) & 0x7fu;
		if (!(link & LINK) || len - LUT_BITS > sb)
			continue;
		const unsigned own = len - LUT_BITS;
		for (uint32_t hi = 0; hi < (1u << (sb - own)); hi++)
			lut[(link & 0xffffffu) + ((bits >> LUT_BITS) | (hi << own))] = ((uint32_t)len << 24) | (uint32_t)i;
	}
	has_lut = true;
	return VALID;
}

// ---------------------------------------------------------------------------------------------
// Tables, header_cached.rs:34-110.  f32 arithmetic in the reference's order; libm sinf/cosf like
// Rust's f32::sin/cos on linux-gnu.  This TU is compiled with -ffp-contract=off.
// ---------------------------------------------------------------------------------------------
static const float PI_F = 3.14159265358979323846264338327950288f;

static uint32_t reverse_bits32(uint32_t x)
{
	x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
	x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
	x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
	x = ((x >> 8) & 0x00ff00ffu) | ((x & 0x00ff00ffu) << 8);
	return (x >> 16) | (x << 16);
}

void BlocksizeTables::init(uint8_t bs)
{
	const uint32_t n = 1u << bs, n2 = n >> 1, n4 = n >> 2, n8 = n >> 3;
	A.resize(n2);
	B.resize(n2);
	C.resize(n4);
	window.resize(n2);
	bitrev.resize(n8);
	const float pi_4_n = 4.0f * PI_F / (float)n;
	const float pi_05_n = 0.5f * PI_F / (float)n;
	const float pi_2_n = 2.0f * PI_F / (float)n;
	for (uint32_t k = 0; k < n4; k++) {
		A[2 * k] = cosf((float)k * pi_4_n);
		A[2 * k + 1] = -sinf((float)k * pi_4_n);
		B[2 * k] = cosf((float)(2 * k + 1) * pi_05_n) * 0.5f;
		B[2 * k + 1] = sinf((float)(2 * k + 1) * pi_05_n) * 0.5f;
	}
	for (uint32_t k = 0; k < n8; k++) {
		C[2 * k] = cosf((float)(2 * k + 1) * pi_2_n);
		C[2 * k + 1] = -sinf((float)(2 * k + 1) * pi_2_n);
	}
	for (uint32_t i = 0; i < n2; i++) {
		const float v = sinf(0.5f * PI_F * ((float)i + 0.5f) / (float)n2);
		window[i] = sinf(0.5f * PI_F * v * v);
	}
	for (uint32_t i = 0; i < n8; i++)
		bitrev[i] = (reverse_bits32(i) >> (32 - bs + 3)) << 2;
}

WindowInfo window_info(const Ident &id, bool blockflag, bool prev_flag, bool next_flag)
{
	WindowInfo w;
	const uint32_t n = 1u << (blockflag ? id.bs1 : id.bs0);
	const uint32_t n0 = 1u << id.bs0;
	w.n = n;
	if (!blockflag || prev_flag) {
		w.left_start = 0;
		w.left_use_bs1 = blockflag;
	} else {
		w.left_start = (n - n0) >> 2;
		w.left_use_bs1 = false;
	}
	if (!blockflag || next_flag) {
		w.right_start = n >> 1;
		w.right_end = n;
	} else {
		w.right_start = (n * 3 - n0) >> 2;
		w.right_end = (n * 3 + n0) >> 2;
	}
	return w;
}

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
float float32_unpack(uint32_t val)
{
	const uint32_t sgn = val & 0x80000000u;
	const uint32_t exp = (val & 0x7fe00000u) >> 21;
	const double mant = (double)(val & 0x1fffffu);
	const double sm = sgn ? -mant : mant;
	return (float)sm * exp2f((float)exp - 788.0f);
}

// header.rs:616-648 restated as "largest r with r^dims <= entries"
static bool pow_leq(uint32_t base, uint16_t dims, uint32_t limit)
{
	uint64_t acc = 1;
	for (uint16_t d = 0; d < dims; d++) {
		acc *= base;
		if (acc > limit)
			return false;
	}
	return true;
}

uint32_t lookup1_values(uint32_t entries, uint16_t dims)
{
	if (dims >= 32)
		return entries == 0 ? 0u : 1u;
	if (dims == 0)
		return entries == 0 ? 0u : 0xffffffffu; // header.rs:664-668 (x^0 = 1 <= entries for every x)
	if (dims == 1)
		return entries;
	// binary search on the base; base <= 65535 for dims >= 2
	uint32_t lo = 0, hi = 65536;
	while (hi - lo > 1) {
		const uint32_t mid = lo + (hi - lo) / 2;
		if (pow_leq(mid, dims, entries))
			lo = mid;
		else
			hi = mid;
	}
	return lo;
}

namespace {

struct Fail {
	int code;
};

struct Rd {
	BitReader r;
	Rd(const uint8_t *p, size_t n) : r(p, n) {}
	uint32_t u(unsigned n)
	{
		uint32_t v;
		if (!r.read(n, v))
			throw Fail{HDR_END_OF_PACKET};
		return v;
	}
	bool flag() { return u(1) != 0; }
};

// header.rs:124-150
uint8_t header_begin(Rd &rd)
{
	const uint32_t t = rd.u(8);
	if ((t & 1) == 0)
		throw Fail{HDR_IS_AUDIO};
	static const uint8_t pat[6] = {'v', 'o', 'r', 'b', 'i', 's'};
	for (int i = 0; i < 6; i++)
		if (rd.u(8) != pat[i])
			throw Fail{HDR_NOT_VORBIS}; // `&&` chain: later bytes are not read
	return (uint8_t)t;
}

void bad()
{
	throw Fail{HDR_BAD_FORMAT};
}

// header.rs:673-768
void read_codebook(Rd &rd, Codebook &cb)
{
	if (rd.u(24) != 0x564342)
		bad();
	cb.dims = (uint16_t)rd.u(16);
	cb.entries = rd.u(24);
	const bool ordered = rd.flag();
	std::vector<uint8_t> lengths;
	lengths.reserve(cb.entries);
	if (!ordered) {
		const bool sparse = rd.flag();
		for (uint32_t i = 0; i < cb.entries; i++) {
			if (sparse && !rd.flag())
				lengths.push_back(0);
			else
				lengths.push_back((uint8_t)(rd.u(5) + 1));
		}
	} else {
		uint32_t cur = 0;
		uint32_t len = rd.u(5) + 1;
		while (cur < cb.entries) {
			const uint32_t number = rd.u(ilog(cb.entries - cur));
			const uint64_t end = (uint64_t)cur + number;
			if (end > cb.entries)
				bad(); // the reference pushes then checks (:717-724); the outcome is the same error
			lengths.insert(lengths.end(), number, (uint8_t)std::min<uint32_t>(len, 255));
			cur += number;
			len++;
		}
	}
	const uint32_t lookup_type = rd.u(4);
	if (lookup_type > 2)
		bad();
	cb.has_vq = lookup_type != 0;
	if (lookup_type != 0) {
		const float vmin = float32_unpack(rd.u(32));
		const float vdelta = float32_unpack(rd.u(32));
		const unsigned vbits = rd.u(4) + 1;
		const bool seq_p = rd.flag();
		const uint64_t lookup_values =
			lookup_type == 1 ? (uint64_t)lookup1_values(cb.entries, cb.dims) : (uint64_t)cb.entries * cb.dims;
		if (lookup_values * vbits > rd.r.remaining())
			throw Fail{HDR_END_OF_PACKET}; // the reads below would run out; avoids a giant allocation
		if ((uint64_t)cb.entries * cb.dims > (1ull << 27))
			throw Fail{HDR_BUFFER_NOT_ADDRESSABLE};
		std::vector<uint32_t> mult((size_t)lookup_values);
		for (auto &m : mult)
			m = rd.u(vbits);
		// lookup_vec_val_decode, header.rs:495-531
		cb.vq.resize((size_t)cb.entries * cb.dims);
		for (uint32_t e = 0; e < cb.entries; e++) {
			float last = 0.0f;
			uint64_t div = 1;
			size_t mo2 = (size_t)e * cb.dims;
			for (uint16_t d = 0; d < cb.dims; d++) {
				uint32_t m;
				if (lookup_type == 1) {
					m = mult[(size_t)((e / (uint32_t)div) % lookup_values)];
					div *= lookup_values;
				} else {
					m = mult[mo2++];
				}
				const float elem = (float)m * vdelta + vmin + last;
				if (seq_p)
					last = elem;
				cb.vq[(size_t)e * cb.dims + d] = elem;
			}
		}
	}
	if (cb.huff.build(lengths.data(), lengths.size()) != Huffman::VALID)
		bad(); // From<HuffmanError>, header.rs:75-79
}

float bark(float x)
{
	return 13.1f * atanf(0.00074f * x) + 2.24f * atanf(0.0000000185f * x * x) + 0.0001f * x;
}

// cos(omega) of every spectrum bin of a floor-0 curve (Vorbis I 6.2.3: map[i] = min(bark_map_size - 1, floor(bark(rate i / 2n)
// bark_map_size / bark(rate / 2))), omega = pi map[i] / bark_map_size), without the spec's sentinel element n.  The f32
// operation order is the reference's (header_cached.rs:142-158): Nyquist first, one Nyquist / n step per bin, one
// bark_map_size / bark(Nyquist) scale -- any other association rounds differently and moves bins across map entries.
std::vector<float> bark_map_cos_omega(uint32_t n, uint16_t rate, uint16_t bark_map_size)
{
	const float nyquist = (float)rate / 2.0f;
	const float bin_hz = nyquist / (float)n;                    // frequency step of one bin
	const float entries_per_bark = (float)bark_map_size / bark(nyquist);
	const float last_entry = (float)bark_map_size - 1.0f;
	const float omega_step = PI_F / (float)bark_map_size;
	std::vector<float> cos_omega(n);
	for (uint32_t bin = 0; bin < n; bin++) {
		const float entry = fminf(floorf(bark((float)bin * bin_hz) * entries_per_bark), last_entry);
		cos_omega[bin] = cosf(entry * omega_step);
	}
	return cos_omega;
}

// header.rs:771-918
void read_floor(Rd &rd, Floor &fl, uint16_t codebook_cnt, uint8_t bs0, uint8_t bs1)
{
	const uint32_t type = rd.u(16);
	if (type == 0) {
		Floor0 &f = fl.f0;
		fl.type = 0;
		f.order = (uint8_t)rd.u(8);
		const uint16_t rate = (uint16_t)rd.u(16);
		const uint16_t bms = (uint16_t)rd.u(16);
		f.amp_bits = (uint8_t)rd.u(6);
		if (f.amp_bits > 64)
			bad();
		f.amp_offset = (uint8_t)rd.u(8);
		f.n_books = (uint8_t)(rd.u(4) + 1);
		for (unsigned i = 0; i < f.n_books; i++) {
			const uint32_t v = rd.u(8);
			if (v > codebook_cnt)
				bad();
			f.book_list[i] = (uint8_t)v;
		}
		f.bark_cos_omega[0] = bark_map_cos_omega(1u << (bs0 - 1), rate, bms);
		f.bark_cos_omega[1] = bark_map_cos_omega(1u << (bs1 - 1), rate, bms);
		return;
	}
	if (type != 1)
		bad();
	Floor1 &f = fl.f1;
	fl.type = 1;
	std::memset(f.sub_books, 0xff, sizeof(f.sub_books));
	const uint32_t parts = rd.u(5);
	int max_class = -1;
	for (uint32_t i = 0; i < parts; i++) {
		const uint8_t c = (uint8_t)rd.u(4);
		max_class = std::max(max_class, (int)c);
		f.partition_class.push_back(c);
	}
	for (int c = 0; c <= max_class; c++) {
		f.class_dim[c] = (uint8_t)(rd.u(3) + 1);
		f.class_sub[c] = (uint8_t)rd.u(2);
		if (f.class_sub[c]) {
			const uint32_t mb = rd.u(8);
			if (mb >= codebook_cnt)
				bad();
			f.class_master[c] = (uint8_t)mb;
		}
		for (unsigned j = 0; j < (1u << f.class_sub[c]); j++) {
			const int book = (int)rd.u(8) - 1;
			if (book >= (int)codebook_cnt)
				bad();
			f.sub_books[c][j] = (int16_t)book;
		}
	}
	f.multiplier = (uint8_t)(rd.u(2) + 1);
	const uint32_t rangebits = rd.u(4);
	uint32_t values = 2;
	for (uint8_t c : f.partition_class)
		values += f.class_dim[c];
	if (values > 65)
		bad();
	f.x_list = {0u, 1u << rangebits};
	for (uint8_t c : f.partition_class)
		for (unsigned j = 0; j < f.class_dim[c]; j++)
			f.x_list.push_back(rd.u(rangebits));
	const size_t F = f.x_list.size();
	std::vector<uint16_t> order(F);
	for (size_t i = 0; i < F; i++)
		order[i] = (uint16_t)i;
	std::stable_sort(order.begin(), order.end(), [&](uint16_t a, uint16_t b) { return f.x_list[a] < f.x_list[b]; });
	f.sorted_idx = order;
	f.sorted_x.resize(F);
	for (size_t i = 0; i < F; i++)
		f.sorted_x[i] = f.x_list[order[i]];
	{
		uint32_t last = 1; // header.rs:892-900 (yes: it starts from 1)
		for (size_t i = 0; i < F; i++) {
			if (f.sorted_x[i] == last)
				bad();
			last = f.sorted_x[i];
		}
	}
	// neighbours of every post among the earlier posts (audio.rs:253-292): header-only, so done once here
	f.lo_idx.assign(F, 0);
	f.hi_idx.assign(F, 0);
	f.dx.assign(F, 0);
	f.adx_magic.assign(F, 0);
	for (size_t i = 2; i < F; i++) {
		int lo = -1, hi = -1;
		for (size_t j = 0; j < i; j++) {
			if (f.x_list[j] < f.x_list[i] && (lo < 0 || f.x_list[j] > f.x_list[lo]))
				lo = (int)j;
			if (f.x_list[j] > f.x_list[i] && (hi < 0 || f.x_list[j] < f.x_list[hi]))
				hi = (int)j;
		}
		if (lo < 0 || hi < 0)
			bad(); // unreachable after the duplicate check (posts 0 and 1 bracket every x < 2^rangebits)
		f.lo_idx[i] = (uint16_t)lo;
		f.hi_idx[i] = (uint16_t)hi;
		f.dx[i] = f.x_list[i] - f.x_list[lo];
		f.adx_magic[i] = UINT64_MAX / (uint64_t)(f.x_list[hi] - f.x_list[lo]) + 1; // adx >= 2: lo < i < hi are distinct integers
	}
}

// header.rs:922-981, ResidueBook::read_book :446-469
void read_residue(Rd &rd, Residue &rs, const std::vector<Codebook> &cbs)
{
	const uint32_t type = rd.u(16);
	if (type > 2)
		bad();
	rs.type = (uint8_t)type;
	rs.begin = rd.u(24);
	rs.end = rd.u(24);
	if (rs.begin > rs.end)
		bad();
	rs.partition_size = rd.u(24) + 1;
	rs.classifications = (uint8_t)(rd.u(6) + 1);
	rs.classbook = (uint8_t)rd.u(8);
	std::vector<uint8_t> cascade;
	for (unsigned i = 0; i < rs.classifications; i++) {
		const uint32_t low = rd.u(3);
		uint32_t high = 0;
		if (rd.flag())
			high = rd.u(5);
		cascade.push_back((uint8_t)((high << 3) | low));
	}
	rs.books.resize(rs.classifications);
	for (unsigned i = 0; i < rs.classifications; i++) {
		rs.books[i].vals_used = cascade[i];
		for (unsigned k = 0; k < 7; k++) {
			if (!(cascade[i] & (1u << k)))
				continue;
			const uint32_t v = rd.u(8);
			if (v >= cbs.size() || !cbs[v].has_vq)
				bad();
			rs.books[i].val_i[k] = (uint8_t)v;
		}
	}
	if (rs.classbook >= cbs.size())
		bad();
	const Codebook &cb = cbs[rs.classbook];
	if (cb.dims && (uint64_t)cb.entries * cb.dims <= (1u << 16)) {
		rs.class_digits.resize((size_t)cb.entries * cb.dims);
		for (uint32_t e = 0; e < cb.entries; e++) {
			uint32_t t = e;
			for (size_t i = cb.dims; i-- > 0;) {
				rs.class_digits[(size_t)e * cb.dims + i] = (uint8_t)(t % rs.classifications);
				t /= rs.classifications;
			}
		}
	}
}

// header.rs:985-1057
void read_mapping(Rd &rd, Mapping &m, uint8_t chan_ilog, uint8_t channels, size_t n_floors, size_t n_residues)
{
	if (rd.u(16) > 0)
		bad();
	const uint32_t submaps = rd.flag() ? rd.u(4) + 1 : 1;
	const uint32_t steps = rd.flag() ? rd.u(8) + 1 : 0;
	for (uint32_t i = 0; i < steps; i++) {
		const uint32_t mg = rd.u(chan_ilog);
		const uint32_t an = rd.u(chan_ilog);
		if (an == mg || mg >= channels || an >= channels)
			bad();
		m.mag.push_back((uint8_t)mg);
		m.ang.push_back((uint8_t)an);
	}
	if (rd.u(2) != 0)
		bad();
	m.mux.assign(channels, 0);
	if (submaps > 1) {
		for (unsigned c = 0; c < channels; c++) {
			const uint32_t v = rd.u(4);
			if (v >= submaps)
				bad();
			m.mux[c] = (uint8_t)v;
		}
	}
	for (uint32_t i = 0; i < submaps; i++) {
		rd.u(8);
		const uint32_t fl = rd.u(8);
		const uint32_t rs = rd.u(8);
		if (fl >= n_floors || rs >= n_residues)
			bad();
		m.submap_floor.push_back((uint8_t)fl);
		m.submap_residue.push_back((uint8_t)rs);
	}
}

} // namespace

std::unique_ptr<Ident> read_header_ident(const uint8_t *pkt, size_t len, int &err)
{
	err = OK;
	try {
		Rd rd(pkt, len);
		const uint8_t t = header_begin(rd);
		if (t != 1)
			throw Fail{HDR_BAD_TYPE};
		if (rd.u(32) != 0)
			throw Fail{HDR_UNSUPPORTED_VERSION};
		auto id = std::make_unique<Ident>();
		id->channels = (uint8_t)rd.u(8);
		id->sample_rate = rd.u(32);
		id->br_max = (int32_t)rd.u(32);
		id->br_nom = (int32_t)rd.u(32);
		id->br_min = (int32_t)rd.u(32);
		id->bs0 = (uint8_t)rd.u(4);
		id->bs1 = (uint8_t)rd.u(4);
		const uint32_t framing = rd.u(8);
		if (id->bs0 < 6 || id->bs0 > 13 || id->bs1 < 6 || id->bs1 > 13 || framing != 1 || id->bs0 > id->bs1 ||
				id->channels == 0 || id->sample_rate == 0)
			bad();
		id->tab[0].init(id->bs0);
		id->tab[1].init(id->bs1);
		return id;
	} catch (const Fail &f) {
		err = f.code;
		return nullptr;
	}
}

// header.rs:309-355 (byte oriented; lenient about non-UTF-8 and '='-less comments like the reference)
static bool valid_utf8(const uint8_t *s, size_t n)
{
	size_t i = 0;
	while (i < n) {
		const uint8_t c = s[i];
		size_t k;
		uint32_t cp;
		if (c < 0x80) {
			i++;
			continue;
		} else if ((c & 0xe0) == 0xc0) {
			k = 1;
			cp = c & 0x1f;
		} else if ((c & 0xf0) == 0xe0) {
			k = 2;
			cp = c & 0x0f;
		} else if ((c & 0xf8) == 0xf0) {
			k = 3;
			cp = c & 0x07;
		} else {
			return false;
		}
		if (i + k >= n)
			return false;
		for (size_t j = 1; j <= k; j++) {
			if ((s[i + j] & 0xc0) != 0x80)
				return false;
			cp = (cp << 6) | (s[i + j] & 0x3f);
		}
		if ((k == 1 && cp < 0x80) || (k == 2 && cp < 0x800) || (k == 3 && (cp < 0x10000 || cp > 0x10ffff)) ||
				(cp >= 0xd800 && cp <= 0xdfff))
			return false;
		i += k + 1;
	}
	return true;
}

std::unique_ptr<Comment> read_header_comment(const uint8_t *pkt, size_t len, int &err)
{
	err = OK;
	size_t pos = 0;
	auto need = [&](size_t n) {
		if (pos + n > len || pos + n < pos)
			throw Fail{HDR_END_OF_PACKET};
	};
	auto u8 = [&]() {
		need(1);
		return pkt[pos++];
	};
	auto u32 = [&]() {
		need(4);
		uint32_t v = (uint32_t)pkt[pos] | ((uint32_t)pkt[pos + 1] << 8) | ((uint32_t)pkt[pos + 2] << 16) |
			((uint32_t)pkt[pos + 3] << 24);
		pos += 4;
		return v;
	};
	try {
		const uint8_t t = u8();
		if ((t & 1) == 0)
			throw Fail{HDR_IS_AUDIO};
		static const uint8_t pat[6] = {'v', 'o', 'r', 'b', 'i', 's'};
		for (int i = 0; i < 6; i++)
			if (u8() != pat[i])
				throw Fail{HDR_NOT_VORBIS};
		if (t != 3)
			throw Fail{HDR_BAD_TYPE};
		auto c = std::make_unique<Comment>();
		const uint32_t vl = u32();
		need(vl);
		if (!valid_utf8(pkt + pos, vl))
			throw Fail{HDR_UTF8};
		c->vendor.assign((const char *)pkt + pos, vl);
		pos += vl;
		const uint32_t cnt = u32();
		for (uint32_t i = 0; i < cnt; i++) {
			const uint32_t cl = u32();
			need(cl);
			const uint8_t *s = pkt + pos;
			pos += cl;
			if (!valid_utf8(s, cl))
				continue;
			const void *eq = std::memchr(s, '=', cl);
			if (!eq)
				continue;
			const size_t k = (const uint8_t *)eq - s;
			c->list.emplace_back(std::string((const char *)s, k), std::string((const char *)s + k + 1, cl - k - 1));
		}
		if (u8() != 1)
			bad();
		return c;
	} catch (const Fail &f) {
		err = f.code;
		return nullptr;
	}
}

std::unique_ptr<Setup> read_header_setup(const uint8_t *pkt, size_t len, uint8_t channels, uint8_t bs0, uint8_t bs1,
		int &err)
{
	err = OK;
	try {
		Rd rd(pkt, len);
		const uint8_t t = header_begin(rd);
		if (t != 5)
			throw Fail{HDR_BAD_TYPE};
		auto s = std::make_unique<Setup>();
		const uint8_t chan_ilog = (uint8_t)ilog((uint64_t)(uint8_t)(channels - 1));
		const uint32_t n_cb = rd.u(8) + 1;
		s->codebooks.resize(n_cb);
		for (auto &cb : s->codebooks)
			read_codebook(rd, cb);
		const uint32_t n_time = rd.u(6) + 1;
		for (uint32_t i = 0; i < n_time; i++)
			if (rd.u(16) != 0)
				bad();
		const uint32_t n_fl = rd.u(6) + 1;
		s->floors.resize(n_fl);
		for (auto &fl : s->floors)
			read_floor(rd, fl, (uint16_t)n_cb, bs0, bs1);
		const uint32_t n_rs = rd.u(6) + 1;
		s->residues.resize(n_rs);
		for (auto &rs : s->residues)
			read_residue(rd, rs, s->codebooks);
		const uint32_t n_mp = rd.u(6) + 1;
		s->mappings.resize(n_mp);
		for (auto &m : s->mappings)
			read_mapping(rd, m, chan_ilog, channels, n_fl, n_rs);
		const uint32_t n_md = rd.u(6) + 1;
		s->modes.resize(n_md);
		for (auto &md : s->modes) {
			md.blockflag = rd.flag();
			const uint32_t wt = rd.u(16), tt = rd.u(16), mp = rd.u(8);
			if (wt != 0 || tt != 0 || mp >= n_mp)
				bad();
			md.mapping = (uint8_t)mp;
		}
		if (!rd.flag())
			bad();
		return s;
	} catch (const Fail &f) {
		err = f.code;
		return nullptr;
	} catch (const std::bad_alloc &) {
		err = HDR_BUFFER_NOT_ADDRESSABLE;
		return nullptr;
	}
}

} // namespace lw
