// Flattened setup header for the device entropy stage: see lw_dev_entropy.hpp / lw_dev_entropy.h.  Product code.
#include "lw_dev_entropy.hpp"

#include <algorithm>
#include <cstring>

namespace lw {

namespace {

template <class T> size_t put(std::vector<uint8_t> &blob, const T *p, size_t n)
{
	const size_t at = (blob.size() + 15) & ~(size_t)15;
	blob.resize(at + n * sizeof(T));
	if (n)
		std::memcpy(blob.data() + at, p, n * sizeof(T));
	return at;
}

} // namespace

bool dev_entropy_build(const Ident &id, const Setup &s, unsigned fstride, DevEntropyImage &img, const char **why)
{
	const char *dummy;
	if (!why)
		why = &dummy;
	*why = "";
	const size_t ch = id.channels;
	const size_t n1 = (size_t)1 << id.bs1;
	img.general = 0;
	if (ch == 0 || ch > LW_ENT_MAX_CH) {
		*why = "more than 16 channels";
		return false;
	}
	if (n1 * ch >= 65536) {
		*why = "blocksize_1 * channels does not fit the reference's u16 product (audio.rs:745)";
		return false;
	}
	if (s.codebooks.size() > 256 || s.floors.size() > 64 || s.residues.size() > 64 || s.modes.size() > 64) {
		*why = "setup too large";
		return false;
	}
	std::vector<bool> book_used(s.codebooks.size(), false), floor_used(s.floors.size(), false), res_used(s.residues.size(), false);
	std::vector<LwEntMode> modes(s.modes.size());
	for (size_t mi = 0; mi < s.modes.size(); mi++) {
		const Mapping &map = s.mappings[s.modes[mi].mapping];
		LwEntMode &m = modes[mi];
		std::memset(&m, 0, sizeof(m));
		m.blockflag = s.modes[mi].blockflag ? 1 : 0;
		const size_t nsm = map.submap_floor.size();
		if (nsm == 0 || nsm > 16 || map.submap_residue.size() != nsm || map.mux.size() < ch) {
			*why = "mapping shape";
			return false;
		}
		if (map.mag.size() > LW_ENT_MAX_COUPLING) {
			*why = "more than 16 coupling steps";
			return false;
		}
		m.n_coupling = (uint8_t)map.mag.size();
		for (size_t i = 0; i < map.mag.size(); i++) {
			m.mag[i] = map.mag[i];
			m.ang[i] = map.ang[i];
		}
		m.n_submaps = (uint8_t)nsm;
		if (nsm != 1)
			img.general = 1;
		for (size_t c = 0; c < ch; c++) {
			if (map.mux[c] >= nsm) {
				*why = "mapping mux";
				return false;
			}
			m.mux[c] = map.mux[c];
			m.floor_of_ch[c] = map.submap_floor[map.mux[c]];
			floor_used[map.submap_floor[map.mux[c]]] = true;
		}
		for (size_t sm = 0; sm < nsm; sm++) {
			m.submap_residue[sm] = map.submap_residue[sm];
			bool has = false;
			for (size_t c = 0; c < ch; c++)
				has |= map.mux[c] == sm;
			if (has) // (a submap without channels decodes nothing, audio.rs:957-986)
				res_used[map.submap_residue[sm]] = true;
		}
	}
	std::vector<LwEntFloor> floors(s.floors.size());
	for (size_t fi = 0; fi < s.floors.size(); fi++) {
		LwEntFloor &f = floors[fi];
		std::memset(&f, 0, sizeof(f));
		if (!floor_used[fi])
			continue;
		if (s.floors[fi].type != 1) {
			*why = "floor type 0";
			return false;
		}
		const Floor1 &fl = s.floors[fi].f1;
		const size_t F = fl.x_list.size();
		if (fl.partition_class.size() > 32 || F > LW_MAX_POSTS || F < 2) {
			*why = "floor 1 shape";
			return false;
		}
		f.multiplier = fl.multiplier;
		f.range = fl.range();
		f.range_bits = ilog(fl.range() - 1);
		f.n_part = (uint32_t)fl.partition_class.size();
		f.F = (uint32_t)F;
		size_t posts = 2;
		for (size_t p = 0; p < fl.partition_class.size(); p++) {
			const unsigned c = fl.partition_class[p];
			if (c >= 16) {
				*why = "floor 1 class number";
				return false;
			}
			posts += fl.class_dim[c];
			if (fl.class_sub[c]) {
				if (fl.class_master[c] >= s.codebooks.size()) {
					*why = "floor 1 master book out of range";
					return false;
				}
				book_used[fl.class_master[c]] = true;
			}
			if (fl.class_sub[c] > 3) {
				*why = "floor 1 subclass bits";
				return false;
			}
			f.part[p] = (uint32_t)fl.class_dim[c] | (uint32_t)fl.class_sub[c] << 8 | (uint32_t)fl.class_master[c] << 16 | (uint32_t)c << 24;
			for (unsigned k = 0; k < (1u << fl.class_sub[c]); k++) {
				const int b = fl.sub_books[c][k];
				if (b >= 0) {
					if ((size_t)b >= s.codebooks.size()) {
						*why = "floor 1 sub book out of range";
						return false;
					}
					book_used[b] = true;
				}
			}
		}
		if (posts != F) {
			*why = "floor 1 post count";
			return false;
		}
		for (int c = 0; c < 16; c++)
			for (int k = 0; k < 8; k++)
				f.sub_books[c * 8 + k] = fl.sub_books[c][k];
		std::vector<unsigned> level(F, 0);
		for (size_t i = 0; i < F; i++) {
			unsigned lo = 0, hi = 0;
			if (i >= 2) {
				lo = fl.lo_idx[i];
				hi = fl.hi_idx[i];
				level[i] = 1 + std::max(level[lo], level[hi]);
				f.n_levels = std::max<uint32_t>(f.n_levels, level[i]);
				f.dx[i] = fl.dx[i];
				f.magic_lo[i] = (uint32_t)(fl.adx_magic[i] & 0xffffffffu);
				f.magic_hi[i] = (uint32_t)(fl.adx_magic[i] >> 32);
			}
			f.post[i] = lo | hi << 8 | level[i] << 16 | (uint32_t)fl.sorted_idx[i] << 24;
		}
	}
	std::vector<LwEntResidue> residues(s.residues.size());
	std::vector<uint16_t> digits;
	std::vector<LwEntRun> runs;
	std::vector<uint32_t> run_book, run_psize; // the book (0xFFFFFFFF: none) and the partition size of every run record
	size_t cls_bytes = 0;
	for (size_t ri = 0; ri < s.residues.size(); ri++) {
		LwEntResidue &r = residues[ri];
		std::memset(&r, 0, sizeof(r));
		r.digits_off = 0xFFFFFFFFu;
		if (!res_used[ri])
			continue;
		const Residue &rs = s.residues[ri];
		if (rs.type > 2 || rs.classifications == 0 || rs.classifications > LW_ENT_MAX_CLASSES || rs.books.size() < rs.classifications ||
				rs.partition_size == 0 || rs.classbook >= s.codebooks.size()) {
			*why = "residue shape";
			return false;
		}
		const Codebook &cbk = s.codebooks[rs.classbook];
		if (cbk.dims == 0 || cbk.dims > 64) {
			*why = "classbook dimensions";
			return false;
		}
		book_used[rs.classbook] = true;
		r.type = rs.type;
		r.classifications = rs.classifications;
		r.classbook = rs.classbook;
		r.cpc = (uint8_t)cbk.dims;
		r.begin = rs.begin;
		r.end = rs.end;
		r.psize = rs.partition_size;
		r.runs_off = (uint32_t)runs.size();
		runs.resize(runs.size() + (size_t)rs.classifications * 8);
		run_book.resize(runs.size(), 0xFFFFFFFFu);
		run_psize.resize(runs.size(), rs.partition_size);
		for (unsigned c = 0; c < rs.classifications; c++) {
			r.vals_used[c] = rs.books[c].vals_used;
			r.used_any |= rs.books[c].vals_used;
			for (unsigned p = 0; p < 8; p++) {
				LwEntRun &run = runs[r.runs_off + c * 8 + p];
				std::memset(&run, 0, sizeof(run));
				run.shape = LW_ENT_SHAPE(0, 0, -2);
				run.nodes_off = 0xFFFFFFFFu;
				if (!(rs.books[c].vals_used & (1u << p)))
					continue;
				const unsigned bi = rs.books[c].val_i[p];
				if (bi >= s.codebooks.size()) {
					*why = "residue book out of range";
					return false;
				}
				const Codebook &cb = s.codebooks[bi];
				if (cb.dims == 0 || cb.dims > 64 || !cb.has_vq || cb.vq.size() < (size_t)cb.entries * cb.dims) {
					*why = "a residue book without a vector lookup";
					return false;
				}
				book_used[bi] = true;
				run.count = rs.type == 0 ? rs.partition_size / cb.dims : (rs.partition_size + cb.dims - 1) / cb.dims;
				run.step = rs.type == 0 ? rs.partition_size / cb.dims : 1u;
				run.adv = rs.type == 0 ? 1u : cb.dims;
				run_book[r.runs_off + c * 8 + p] = bi; // (the book's table entries are filled in below)
			}
		}
		if (!rs.class_digits.empty()) {
			r.digits_off = (uint32_t)digits.size();
			for (uint8_t cl : rs.class_digits)
				digits.push_back((uint16_t)(cl | ((unsigned)rs.books[cl].vals_used << 8)));
		}
		const size_t nch = rs.type == 2 ? 1 : ch, actual = rs.type == 2 ? ch * (n1 / 2) : n1 / 2;
		cls_bytes = std::max(cls_bytes, 2 * nch * (actual / rs.partition_size + cbk.dims));
	}
	std::vector<LwEntBook> books(s.codebooks.size());
	std::vector<uint32_t> lut;
	std::vector<float> vq;
	std::vector<int32_t> nodes;
	for (size_t bi = 0; bi < s.codebooks.size(); bi++) {
		LwEntBook &b = books[bi];
		std::memset(&b, 0, sizeof(b));
		int single = -1;
		unsigned lut_bits = 0;
		b.shape = LW_ENT_SHAPE(0, 0, -1);
		if (!book_used[bi])
			continue;
		const Codebook &cb = s.codebooks[bi];
		b.nodes_off = 0xFFFFFFFFu;
		if (cb.huff.single < 0 && !cb.huff.has_lut) {
			single = -2; // empty book (the host stage ends the packet at its first codeword)
		} else if (cb.huff.single >= 0) {
			if (cb.huff.single > 32767) {
				*why = "single-entry book with a large entry number";
				return false;
			}
			single = cb.huff.single;
		} else {
			b.lut_off = (uint32_t)lut.size();
			lut_bits = cb.huff.lut_bits;
			for (uint32_t e : cb.huff.lut) // (length 0 = "not in the tables": its own marker here, see LW_ENT_WALK)
				lut.push_back(!(e & Huffman::LINK) && (e >> 24) == 0 ? LW_ENT_WALK : e);
			bool walks = false; // some code is longer than the two table levels: ship the tree as well
			for (uint32_t e : cb.huff.lut)
				walks |= !(e & Huffman::LINK) && (e >> 24) == 0;
			if (walks && !cb.huff.nodes.empty()) {
				b.nodes_off = (uint32_t)nodes.size();
				nodes.insert(nodes.end(), cb.huff.nodes.begin(), cb.huff.nodes.end());
			}
		}
		b.shape = LW_ENT_SHAPE(lut_bits, std::min<unsigned>(cb.dims, 255), single);
		if (cb.has_vq && !cb.vq.empty()) {
			while (vq.size() % 8)
				vq.push_back(0.0f);
			b.vq_off = (uint32_t)vq.size();
			vq.insert(vq.end(), cb.vq.begin(), cb.vq.end());
		}
	}
	for (size_t k = 0; k < runs.size(); k++) {
		if (run_book[k] == 0xFFFFFFFFu)
			continue; // a (class, pass) without a book
		LwEntRun &run = runs[k];
		const LwEntBook &b = books[run_book[k]];
		run.lut_off = b.lut_off * 4;
		run.vq_off = b.vq_off * 4;
		run.nodes_off = b.nodes_off;
		run.shape = b.shape;
		const unsigned dims = (b.shape >> 8) & 0xffu;
		const bool whole = run.adv == 1 || run.count * dims == run_psize[k]; // (type 0, or the dimension divides the partition)
		if ((int16_t)(b.shape >> 16) == -1 && run.count >= 1 && dims && whole)
			for (unsigned d = 1; d <= 8; d++)
				if (dims % d == 0)
					run.fast |= 1u << d;
	}
	lut.push_back(0);
	vq.push_back(0.0f);
	digits.push_back(0);
	runs.emplace_back();
	nodes.push_back(0);
	img.blob.clear();
	img.off_books = put(img.blob, books.data(), books.size());
	img.off_floors = put(img.blob, floors.data(), floors.size());
	img.off_residues = put(img.blob, residues.data(), residues.size());
	img.off_modes = put(img.blob, modes.data(), modes.size());
	img.off_lut = put(img.blob, lut.data(), lut.size());
	img.off_vq = put(img.blob, vq.data(), vq.size());
	img.off_digits = put(img.blob, digits.data(), digits.size());
	img.off_runs = put(img.blob, runs.data(), runs.size());
	img.off_nodes = put(img.blob, nodes.data(), nodes.size());
	img.ch = (uint32_t)ch;
	img.fstride = fstride;
	img.ws_bytes = (uint32_t)((LW_ENT_POSTS_BYTES + cls_bytes + 15) & ~(size_t)15);
	img.res_floats = (uint32_t)(ch * (n1 / 2));
	// one wave = one packet with its accumulators, dump slots and work space in LDS (lw_launch_entropy): what does not fit a
	// workgroup's 160 KB (many channels of 8192-point blocks, or residues cut into thousands of tiny partitions, whose class digits
	// are the work space) keeps the host stage -- found by the random-setup campaign: the launch failed with "invalid argument"
	if (((size_t)img.res_floats + LW_ENT_DUMP_FLOATS) * 4 + img.ws_bytes > LW_ENT_MAX_LDS) {
		*why = "accumulators and class digits of a packet exceed a workgroup's LDS";
		return false;
	}
	return true;
}

LwEntTables dev_entropy_view(const DevEntropyImage &img, const uint8_t *base)
{
	LwEntTables T;
	T.books = (const LW_K LwEntBook *)(base + img.off_books);
	T.floors = (const LW_K LwEntFloor *)(base + img.off_floors);
	T.residues = (const LW_K LwEntResidue *)(base + img.off_residues);
	T.modes = (const LW_K LwEntMode *)(base + img.off_modes);
	T.lut = (const LW_K uint32_t *)(base + img.off_lut);
	T.vq = (const LW_K float *)(base + img.off_vq);
	T.digits = (const LW_K uint16_t *)(base + img.off_digits);
	T.runs = (const LW_K LwEntRun *)(base + img.off_runs);
	T.nodes = (const LW_K int32_t *)(base + img.off_nodes);
	T.ch = img.ch;
	T.fstride = img.fstride;
	T.ws_bytes = img.ws_bytes;
	T.res_floats = img.res_floats;
	T.general = img.general;
	return T;
}

} // namespace lw
