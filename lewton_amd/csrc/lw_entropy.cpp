// Host entropy stage (product code).  See lw_entropy.hpp.
#include "lw_entropy.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace lw {

int read_prologue(const Ident &id, const Setup &s, BitReader &r, Prologue &p)
{
	bool hdr;
	if (!r.flag(hdr))
		return AUDIO_END_OF_PACKET;
	if (hdr)
		return AUDIO_IS_HEADER; // audio.rs:922-924
	uint32_t mode;
	if (!r.read(ilog((uint64_t)s.modes.size() - 1), mode))
		return AUDIO_END_OF_PACKET;
	if (mode >= s.modes.size())
		return AUDIO_BAD_FORMAT; // audio.rs:926-930
	p.mode = (uint8_t)mode;
	p.blockflag = s.modes[mode].blockflag;
	p.bs = p.blockflag ? id.bs1 : id.bs0;
	p.n = 1u << p.bs;
	p.prev_flag = p.next_flag = false;
	if (p.blockflag) {
		if (!r.flag(p.prev_flag) || !r.flag(p.next_flag))
			return AUDIO_END_OF_PACKET;
	}
	return OK;
}

int decoded_sample_count(const Ident &id, const Setup &s, const uint8_t *pkt, size_t len, size_t &count)
{
	BitReader r(pkt, len);
	Prologue p;
	const int rc = read_prologue(id, s, r, p);
	if (rc)
		return rc;
	const WindowInfo w = window_info(id, p.blockflag, p.prev_flag, p.next_flag);
	count = w.right_start - w.left_start;
	return OK;
}

namespace {

// FL_REF_PANICS: a state in which the reference indexes out of bounds / divides by zero (a Rust panic, i.e. no defined
// result); the packet is reported as AUDIO_BAD_FORMAT instead of touching memory the setup does not describe
enum FloorResult { FL_OK, FL_UNUSED, FL_UNDECODABLE, FL_REF_PANICS };

// audio.rs:215-251
FloorResult floor_one_decode(BitReader &r, const Setup &s, const Floor1 &fl, uint32_t *y)
{
	bool nonzero;
	if (!r.flag(nonzero) || !nonzero)
		return FL_UNUSED;
	const unsigned b = ilog(fl.range() - 1);
	size_t k = 0;
	if (!r.read(b, y[k]))
		return FL_UNUSED;
	k++;
	if (!r.read(b, y[k]))
		return FL_UNUSED;
	k++;
	CodeReader cr(r); // only codewords from here on; a failed read leaves r at the end of the packet like BitReader would
	for (uint8_t c : fl.partition_class) {
		const unsigned cdim = fl.class_dim[c], cbits = fl.class_sub[c];
		const uint32_t csub = (1u << cbits) - 1;
		uint32_t cval = 0;
		if (cbits && !cr.next(s.codebooks[fl.class_master[c]].huff, cval))
			return FL_UNUSED;
		for (unsigned d = 0; d < cdim; d++) {
			const int book = fl.sub_books[c][cval & csub];
			cval >>= cbits;
			if (book >= 0) {
				if (!cr.next(s.codebooks[book].huff, y[k]))
					return FL_UNUSED;
			} else {
				y[k] = 0;
			}
			k++;
		}
	}
	cr.sync();
	return FL_OK;
}

// floor_zero_decode, audio.rs:109-158: cosines of the LSP coefficients + amplitude.  Any bit-reader underrun is
// "unused" (From<()> for FloorSpecialCase), a book number outside the list or a codebook without a VQ lookup is
// "undecodable".
FloorResult floor_zero_decode(BitReader &r, const Setup &s, const Floor0 &fl, float *coeff, uint64_t &amplitude)
{
	if (!r.read64(fl.amp_bits, amplitude) || amplitude == 0)
		return FL_UNUSED;
	uint32_t booknumber;
	if (!r.read(ilog(fl.n_books), booknumber))
		return FL_UNUSED;
	if (booknumber >= fl.n_books)
		return FL_UNDECODABLE;
	// header.rs:793 accepts a book number equal to the codebook count (`>` instead of `>=`); the reference then panics on
	// `codebooks[idx]` here (:127)
	if (fl.book_list[booknumber] >= s.codebooks.size())
		return FL_REF_PANICS;
	const Codebook &cb = s.codebooks[fl.book_list[booknumber]];
	size_t n = 0;
	float last = 0.0f;
	for (;;) {
		float last_new = last;
		uint32_t idx;
		if (!cb.huff.decode(r, idx))
			return FL_UNUSED;
		if (!cb.has_vq)
			return FL_UNDECODABLE;
		// floor0_order 0 or 1: the reference collects the first vector (order 0: all of it, whatever its length) and
		// returns Ok; floor_zero_compute_curve then evaluates `(order - 2) / 2` resp. `(order - 3) / 2` in usize, which
		// wraps, and panics indexing the coefficients (:176-191).  Nothing is written to coeff[] for such a floor.
		if (fl.order < 2 && cb.dims != 0)
			return FL_REF_PANICS;
		for (size_t d = 0; d < cb.dims; d++) {
			const float e = cb.vq[(size_t)idx * cb.dims + d];
			coeff[n++] = cosf(last + e);
			last_new = e;
			if (n == fl.order)
				return FL_OK;
		}
		last += last_new;
		if (n >= fl.order)
			return fl.order < 2 ? FL_REF_PANICS : FL_OK; // (order 0 with a zero-dimensional book: Ok, then the curve panics)
	}
}

// floor_zero_compute_curve, audio.rs:160-212: one value per run of equal bark-map entries
void floor_zero_curve(const float *cosc, uint64_t amplitude, const Floor0 &fl, bool blockflag, size_t n, float *out)
{
	const float *bark_cos = fl.bark_cos_omega[blockflag ? 1 : 0].data();
	// `((1 << bits) - 1) as f32` (:167) is i32 arithmetic: in a release build the shift count is taken mod 32 and the
	// subtraction wraps (bits = 31 gives i32::MAX, bits = 32 gives 0 and an infinite / NaN curve)
	const int32_t denom = (int32_t)((1u << (fl.amp_bits & 31u)) - 1u);
	const float common = (float)amplitude * (float)fl.amp_offset / (float)denom;
	size_t i = 0;
	while (i < n) {
		const float cos_omega = bark_cos[i];
		size_t p_ub, q_ub;
		float p, q;
		if (fl.order & 1) {
			p_ub = ((size_t)fl.order - 3) / 2;
			q_ub = ((size_t)fl.order - 1) / 2;
			p = 1.0f - cos_omega * cos_omega;
			q = 0.25f;
		} else {
			p_ub = q_ub = ((size_t)fl.order - 2) / 2;
			p = (1.0f - cos_omega) / 2.0f;
			q = (1.0f + cos_omega) / 2.0f;
		}
		for (size_t j = 0; j < p_ub + 1; j++) {
			const float pm = cosc[2 * j + 1] - cos_omega;
			p *= 4.0f * pm * pm;
		}
		for (size_t j = 0; j < q_ub + 1; j++) {
			const float qm = cosc[2 * j] - cos_omega;
			q *= 4.0f * qm * qm;
		}
		const float lfv = expf(0.11512925f * (common / sqrtf(p + q) - (float)fl.amp_offset));
		do {
			out[i++] = lfv;
		} while (i < n && bark_cos[i] == cos_omega);
	}
}

// audio.rs:354-367 with wrapping u32 arithmetic (release-mode Rust).  x - x0 and the divisor adx = x1 - x0 are
// header constants of the post: the division is a multiplication by the post's precomputed 2^64 / adx + 1, exact for
// every 32-bit dividend (the divisor is below 2^16).
inline uint32_t render_point(uint32_t y0, uint32_t y1, uint32_t dx, uint64_t adx_magic)
{
	const int32_t dy = (int32_t)(y1 - y0);
	const uint32_t ady = dy < 0 ? 0u - (uint32_t)dy : (uint32_t)dy;
	const uint32_t off = (uint32_t)(((unsigned __int128)(uint32_t)(ady * dx) * adx_magic) >> 64);
	return dy < 0 ? y0 - off : y0 + off;
}

// audio.rs:391-435 -> device record: per post in ascending-x order, (final_y * multiplier) | active flag
void floor_one_record(const uint32_t *y, const Floor1 &fl, uint16_t *rec)
{
	const size_t F = fl.x_list.size();
	const uint32_t range = fl.range();
	uint32_t final_y[LW_MAX_POSTS];
	bool step2[LW_MAX_POSTS];
	final_y[0] = y[0];
	final_y[1] = y[1];
	step2[0] = step2[1] = true;
	for (size_t i = 2; i < F; i++) {
		const unsigned lo = fl.lo_idx[i], hi = fl.hi_idx[i];
		const int32_t predicted = (int32_t)render_point(final_y[lo], final_y[hi], fl.dx[i], fl.adx_magic[i]);
		const int32_t val = (int32_t)y[i];
		const int32_t highroom = (int32_t)(range - (uint32_t)predicted);
		const int32_t lowroom = predicted;
		const int32_t room = (int32_t)((uint32_t)std::min(highroom, lowroom) * 2u);
		if (val > 0) {
			step2[lo] = step2[hi] = step2[i] = true;
			uint32_t fy;
			if (val >= room) {
				fy = highroom > lowroom ? (uint32_t)predicted + (uint32_t)val - (uint32_t)lowroom
				                        : (uint32_t)predicted - (uint32_t)val + (uint32_t)highroom - 1u;
			} else {
				const int32_t t = (val % 2 == 1) ? (int32_t)(0u - (uint32_t)val - 1u) : val;
				fy = (uint32_t)predicted + (uint32_t)(t >> 1);
			}
			final_y[i] = fy;
		} else {
			final_y[i] = (uint32_t)predicted;
			step2[i] = false;
		}
	}
	for (size_t sidx = 0; sidx < F; sidx++) {
		const unsigned i = fl.sorted_idx[sidx];
		const uint32_t fy = std::min(range - 1, final_y[i]); // :431-433
		rec[sidx] = (uint16_t)((fy * fl.multiplier) & 0xffu) | (step2[i] ? LW_POST_ACTIVE : 0u);
	}
}

// v[0..DIMS) += e[0..DIMS): the same f32 additions in the same order for every DIMS (lanes are independent)
template <unsigned DIMS> inline void add_entry(float *v, const float *e)
{
	for (unsigned j = 0; j < DIMS; j++)
		v[j] += e[j];
}

// audio.rs:587-618.
// 1 = partition done, 0 = end of packet, -1 = the reference panics (residue type 0 divides by a zero book dimension)
inline int read_partition(CodeReader &cr, const Codebook &cb, unsigned rtype, unsigned psize, float *v, size_t vec_len)
{
	const unsigned dims = cb.dims;
	const Huffman &h = cb.huff;
	uint32_t idx;
	if (dims == 0) {
		// a lookup-type-2 book may have zero dimensions (header.rs:548-560 accepts it).  Type 0: `partition_size / 0`
		// panics (:592).  Types 1/2: every vector is empty, `i` never advances (:600-612) and codewords are consumed
		// until the packet ends.
		if (rtype == 0)
			return -1;
		while (cr.next(h, idx)) {
		}
		return 0;
	}
	if (rtype == 0) {
		const unsigned step = psize / dims;
		for (unsigned i = 0; i < step; i++) {
			if (!cr.next(h, idx))
				return 0;
			const float *e = &cb.vq[(size_t)idx * dims];
			for (unsigned j = 0; j < dims; j++)
				v[i + j * step] += e[j];
		}
		return 1;
	}
	const float *vq = cb.vq.data();
	if ((size_t)psize <= vec_len && psize % dims == 0) { // the whole partition is inside the vector
		switch (dims) {
#define LW_PART_LOOP(D)                               \
	case D:                                           \
		for (unsigned i = 0; i < psize; i += D) {     \
			if (!cr.next(h, idx))                     \
				return 0;                             \
			add_entry<D>(v + i, vq + (size_t)idx * D); \
		}                                             \
		return 1;
			LW_PART_LOOP(1)
			LW_PART_LOOP(2)
			LW_PART_LOOP(4)
			LW_PART_LOOP(8)
#undef LW_PART_LOOP
		default:
			break;
		}
	}
	unsigned i = 0;
	while (i < psize) {
		if (!cr.next(h, idx))
			return 0;
		if ((size_t)i + dims > vec_len)
			break;
		const float *e = vq + (size_t)idx * dims;
		for (unsigned j = 0; j < dims; j++)
			v[i + j] += e[j];
		i += dims;
	}
	return 1;
}

// audio.rs:620-717; `vectors` = ch * (cur_blocksize/2) zeros.  false = Err(()) (packet undecodable)
bool residue_inner(BitReader &r, const Setup &s, const Residue &rs, size_t cur_blocksize, const bool *dnd, size_t ch,
		float *vectors, EntropyScratch &scr)
{
	const size_t actual = cur_blocksize / 2;
	const size_t begin = std::min<size_t>(rs.begin, actual), end = std::min<size_t>(rs.end, actual);
	const Codebook &classbook = s.codebooks[rs.classbook];
	const size_t cpc = classbook.dims;
	const size_t n_to_read = end - begin;
	const size_t parts = n_to_read / rs.partition_size;
	if (n_to_read == 0)
		return true;
	if (cpc == 0)
		return false;
	const size_t stride = parts + cpc;
	scr.cls.assign(ch * stride, 0);
	uint32_t *cls = scr.cls.data();
	const uint32_t ncls = rs.classifications;
	const uint8_t *digits = rs.class_digits.empty() ? nullptr : rs.class_digits.data();
	CodeReader cr(r);
	// every exit leaves r.pos where the reference's reader stands: a failed read went through Huffman::decode (which moved
	// r.pos to the end of the packet), a completed loop syncs below
	for (unsigned pass = 0; pass < 8; pass++) {
		size_t pc = 0;
		while (pc < parts) {
			if (pass == 0) {
				for (size_t j = 0; j < ch; j++) {
					if (dnd[j])
						continue;
					uint32_t t;
					if (!cr.next(classbook.huff, t))
						return true; // end of packet is normal (audio.rs:655-660)
					uint32_t *c = cls + j * stride + pc;
					if (digits) {
						const uint8_t *dg = digits + (size_t)t * cpc;
						for (size_t i = 0; i < cpc; i++)
							c[i] = dg[i];
					} else {
						for (size_t i = cpc; i-- > 0;) {
							c[i] = t % ncls;
							t /= ncls;
						}
					}
				}
			}
			for (size_t k = 0; k < cpc && pc < parts; k++, pc++) {
				for (size_t j = 0; j < ch; j++) {
					if (dnd[j])
						continue;
					const ResidueBook &rb = rs.books[cls[j * stride + pc]];
					if (!(rb.vals_used & (1u << pass)))
						continue;
					const size_t offs = begin + pc * rs.partition_size;
					const int pr = read_partition(cr, s.codebooks[rb.val_i[pass]], rs.type, rs.partition_size,
							vectors + j * actual + offs, actual - offs);
					if (pr < 0)
						return false;
					if (pr == 0)
						return true;
				}
			}
		}
	}
	cr.sync();
	return true;
}

// audio.rs:722-760
bool residue_decode(BitReader &r, const Setup &s, const Residue &rs, size_t n, const bool *dnd, size_t ch, float *out,
		EntropyScratch &scr)
{
	const size_t half = n / 2;
	bool any = false;
	for (size_t j = 0; j < ch; j++)
		any |= !dnd[j];
	if (rs.type != 2 || !any) // (the type-2 path below writes every element of `out`)
		std::memset(out, 0, sizeof(float) * ch * half);
	if (rs.type != 2)
		return residue_inner(r, s, rs, n, dnd, ch, out, scr);
	if (!any)
		return true;
	const size_t bs2 = (size_t)(uint16_t)((uint16_t)n * (uint16_t)ch); // `cur_blocksize * ch as u16` wraps (:745)
	if (bs2 / 2 < ch * half)
		return false; // wrapped: the reference panics slicing the short vector; report the packet as bad
	const bool one_dnd[1] = {false};
	scr.interleaved.assign(ch * half, 0.0f);
	if (!residue_inner(r, s, rs, bs2, one_dnd, 1, scr.interleaved.data(), scr))
		return false;
	const float *v = scr.interleaved.data();
	if (ch == 2) {
		float *o0 = out, *o1 = out + half;
		for (size_t k = 0; k < half; k++) {
			o0[k] = v[2 * k];
			o1[k] = v[2 * k + 1];
		}
	} else {
		for (size_t j = 0; j < ch; j++)
			for (size_t k = 0; k < half; k++)
				out[j * half + k] = v[k * ch + j];
	}
	return true;
}

} // namespace

int entropy_decode(const Ident &id, const Setup &s, const uint8_t *pkt, size_t len, Prologue &p, uint16_t *floor_out,
		unsigned fstride, float *residue_out, EntropyScratch &scr, uint64_t *bits_consumed, float *fcurve_out)
{
	BitReader r(pkt, len);
	int rc = read_prologue(id, s, r, p);
	if (rc)
		return rc;
	const Mapping &map = s.mappings[s.modes[p.mode].mapping];
	const size_t ch = id.channels;
	const size_t half = p.n / 2;
	bool no_residue[256];
	// floor_decode, audio.rs:557-585
	for (size_t c = 0; c < ch; c++) {
		const Floor &fl = s.floors[map.submap_floor[map.mux[c]]];
		uint16_t *rec = floor_out + c * fstride;
		if (fl.type == 0) {
			if (!fcurve_out)
				return AUDIO_BAD_FORMAT; // the caller did not provide room for explicit floor curves
			float coeff[256 + 8];
			uint64_t amplitude = 0;
			const FloorResult fr = floor_zero_decode(r, s, fl.f0, coeff, amplitude);
			if (fr == FL_REF_PANICS)
				return AUDIO_BAD_FORMAT;
			if (fr == FL_UNDECODABLE)
				return AUDIO_END_OF_PACKET;
			if (fr == FL_UNUSED) {
				rec[0] = LW_FLOOR_UNUSED;
				no_residue[c] = true;
			} else {
				floor_zero_curve(coeff, amplitude, fl.f0, p.blockflag, half, fcurve_out + c * half);
				rec[0] = LW_FLOOR_EXPLICIT;
				no_residue[c] = false;
			}
			continue;
		}
		uint32_t y[LW_MAX_POSTS];
		const FloorResult fr = floor_one_decode(r, s, fl.f1, y);
		if (fr == FL_UNDECODABLE)
			return AUDIO_END_OF_PACKET;
		if (fr == FL_UNUSED) {
			rec[0] = LW_FLOOR_UNUSED;
			no_residue[c] = true;
		} else {
			floor_one_record(y, fl.f1, rec);
			no_residue[c] = false;
		}
	}
	// audio.rs:948-955
	for (size_t i = 0; i < map.mag.size(); i++) {
		const unsigned m = map.mag[i], a = map.ang[i];
		if (!(no_residue[m] && no_residue[a]))
			no_residue[m] = no_residue[a] = false;
	}
	// audio.rs:957-986
	const size_t n_submaps = map.submap_floor.size();
	for (size_t sm = 0; sm < n_submaps; sm++) {
		bool dnd[256];
		size_t chans[256];
		size_t sub_ch = 0;
		for (size_t c = 0; c < ch; c++) {
			if (map.mux[c] == sm) {
				dnd[sub_ch] = no_residue[c];
				chans[sub_ch++] = c;
			}
		}
		const Residue &rs = s.residues[map.submap_residue[sm]];
		// channels of a submap are usually contiguous and in order: decode straight into the output block
		bool contiguous = sub_ch > 0;
		for (size_t k = 1; k < sub_ch; k++)
			contiguous &= chans[k] == chans[0] + k;
		if (contiguous) {
			if (!residue_decode(r, s, rs, p.n, dnd, sub_ch, residue_out + chans[0] * half, scr))
				return AUDIO_BAD_FORMAT;
		} else {
			scr.sub.assign(sub_ch * half + 1, 0.0f);
			if (!residue_decode(r, s, rs, p.n, dnd, sub_ch, scr.sub.data(), scr))
				return AUDIO_BAD_FORMAT;
			for (size_t k = 0; k < sub_ch; k++)
				std::memcpy(residue_out + chans[k] * half, scr.sub.data() + k * half, sizeof(float) * half);
		}
	}
	if (bits_consumed)
		*bits_consumed = r.pos;
	return OK;
}

} // namespace lw
