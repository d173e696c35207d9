// Host side of the specialised long-block kernel: see lw_fast.hpp.  Product code.
#include "lw_fast.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>

namespace lw {

namespace {
struct Writer {
	std::vector<uint8_t> &buf;
	uint32_t reserve(size_t bytes)
	{
		const size_t off = (buf.size() + 15) & ~(size_t)15;
		buf.resize(off + bytes, 0);
		return (uint32_t)off;
	}
	float *f(uint32_t off) { return reinterpret_cast<float *>(buf.data() + off); }
	uint16_t *h(uint32_t off) { return reinterpret_cast<uint16_t *>(buf.data() + off); }
};
} // namespace

// The unit list of the modes with `blockflag` (shared by every specialised kernel).  Native shape: every such mode uses the same
// coupling list of disjoint channel pairs and the same floor per channel, at most LW_FAST_MAX_FLOORS distinct floor-1
// configurations of at most max_posts posts -- the units then are the coupling steps plus the remaining channels two by two.
// Every other shape gets the canonicalising pre-pass (LwPrepPlan, lw_fast.hpp): uncoupled units, the channels whose floor the
// kernels cannot stage on the unit floor.  Returns nullptr, or why the stream shape is not covered at all.
struct UnitPlan {
	std::vector<uint32_t> pre;   // (allow_pre) per unit: coupling steps evaluated inside the wave (LwFastPlan::pre), or empty
	std::vector<LwFastUnit> units;
	std::vector<int> floor_slot; // per floor of the setup: staged slot or -1
	uint32_t n_staged = 0;       // staged slots incl. the unit floor's
	uint8_t staged_F[LW_FAST_MAX_FLOORS] = {0};
	uint8_t mode_mask[32] = {0};
	LwPrepPlan prep;
};

// blocksize_0 = blocksize_1 and a mode with the block flag: the two flags then name the same block shape (one size, full slopes on
// both sides whatever the window flags say, audio.rs:1056-1073), and every mode of the stream is served as ONE class, the long one
bool lw_unified_classes(const Ident &id, const Setup &s)
{
	if (id.bs0 != id.bs1)
		return false;
	for (const Mode &m : s.modes)
		if (m.blockflag)
			return true;
	return false;
}

// ---- coupling steps inside the waves (LwFastPlan::pre): the values the channels have after the steps BEHIND the disjoint prefix
// (applied first, last step of the list first), as expression strings over the raw vectors: R<c>, M(e,e), A(e,e)
static void pre_exprs(const Mapping &mp, size_t ch, size_t prefix, std::vector<std::string> &expr)
{
	expr.resize(ch);
	for (size_t c = 0; c < ch; c++)
		expr[c] = "R" + std::to_string(c);
	for (size_t k = mp.mag.size(); k-- > prefix;) {
		const size_t m = mp.mag[k], a = mp.ang[k];
		const std::string nm = "M(" + expr[m] + "," + expr[a] + ")", na = "A(" + expr[m] + "," + expr[a] + ")";
		expr[m] = nm;
		expr[a] = na;
	}
}

// the program of one unit: registers r0 = raw channel a, r1 = raw channel b, t0 / t1 = up to two more raw channels; up to
// LW_PRE_MAX_OPS steps (m_reg, a_reg) in place until r0 (and r1) hold the wanted expressions.  False: not expressible that way.
static bool pre_program(const std::vector<std::string> &expr, int ch_a, int ch_b, uint32_t &prog)
{
	const std::string &want_a = expr[(size_t)ch_a];
	const std::string want_b = ch_b >= 0 ? expr[(size_t)ch_b] : std::string();
	// the raw channels the two expressions mention, besides the unit's own
	std::vector<int> extra;
	for (const std::string *e : {&want_a, &want_b})
		for (size_t i = 0; i < e->size(); i++)
			if ((*e)[i] == 'R') {
				const int c = std::atoi(e->c_str() + i + 1);
				if (c != ch_a && c != ch_b && std::find(extra.begin(), extra.end(), c) == extra.end())
					extra.push_back(c);
			}
	if (extra.size() > 2)
		return false;
	struct State {
		std::string r[4];
		uint8_t ops[LW_PRE_MAX_OPS];
		uint32_t n;
	};
	State s0;
	s0.r[0] = "R" + std::to_string(ch_a);
	s0.r[1] = ch_b >= 0 ? "R" + std::to_string(ch_b) : std::string();
	s0.r[2] = extra.size() > 0 ? "R" + std::to_string(extra[0]) : std::string();
	s0.r[3] = extra.size() > 1 ? "R" + std::to_string(extra[1]) : std::string();
	s0.n = 0;
	auto done = [&](const State &st) { return st.r[0] == want_a && (ch_b < 0 || st.r[1] == want_b); };
	std::vector<State> level{s0};
	for (uint32_t depth = 0; depth <= LW_PRE_MAX_OPS; depth++) {
		for (const State &st : level)
			if (done(st)) {
				prog = (extra.size() > 0 ? (uint32_t)extra[0] : LW_PRE_NO_CH) | ((extra.size() > 1 ? (uint32_t)extra[1] : LW_PRE_NO_CH) << 8) | (st.n << 28);
				for (uint32_t i = 0; i < st.n; i++)
					prog |= (uint32_t)st.ops[i] << (16 + 4 * i);
				return true;
			}
		if (depth == LW_PRE_MAX_OPS)
			break;
		std::vector<State> next;
		for (const State &st : level)
			for (int m = 0; m < 4; m++)
				for (int a = 0; a < 4; a++) {
					if (m == a || st.r[m].empty() || st.r[a].empty() || st.r[m].size() + st.r[a].size() > 200)
						continue;
					State nx = st;
					nx.r[m] = "M(" + st.r[m] + "," + st.r[a] + ")";
					nx.r[a] = "A(" + st.r[m] + "," + st.r[a] + ")";
					nx.ops[nx.n++] = (uint8_t)((m << 2) | a);
					next.push_back(std::move(nx));
				}
		level.swap(next);
	}
	return false;
}

static const char *plan_units(const Ident &id, const Setup &s, bool blockflag, size_t max_posts, UnitPlan &up, bool allow_pre = false)
{
	const size_t ch = id.channels, nmodes = s.modes.size();
	const bool unified = lw_unified_classes(id, s);
	std::vector<size_t> modes;
	for (size_t m = 0; m < nmodes && m < 256; m++)
		if ((bool)s.modes[m].blockflag == blockflag || (unified && blockflag)) {
			modes.push_back(m);
			up.mode_mask[m >> 3] |= (uint8_t)(1u << (m & 7));
		}
	if (modes.empty())
		return blockflag ? "no long mode" : "no short mode";
	if ((ch + 1) / 2 > LW_FAST_WAVES)
		return "more units than waves in a workgroup";
	const Mapping &ref = s.mappings[s.modes[modes[0]].mapping];
	auto floor_of = [&](size_t m, size_t c) -> int {
		const Mapping &mp = s.mappings[s.modes[m].mapping];
		return (int)mp.submap_floor[mp.mux[c]];
	};
	// coupling: native when every covered mode has the reference list and its steps are disjoint pairs
	const char *why = nullptr;
	std::vector<int> partner(ch, -1), role(ch, 0);
	for (size_t k = 0; k < ref.mag.size() && !why; k++) {
		const int m = ref.mag[k], a = ref.ang[k];
		if (partner[m] != -1 || partner[a] != -1)
			why = "a channel takes part in more than one coupling step";
		partner[m] = a;
		partner[a] = m;
		role[m] = 1;
		role[a] = 2;
	}
	bool coupling_same = true;
	for (size_t m : modes) {
		const Mapping &mp = s.mappings[s.modes[m].mapping];
		if (mp.mag != ref.mag || mp.ang != ref.ang) {
			coupling_same = false;
			if (!why)
				why = blockflag ? "long modes with different coupling lists" : "short modes with different coupling lists";
		}
	}
	// floors: a channel is native when all covered modes give it the same floor-1 configuration of at most max_posts posts
	std::vector<int> native(ch, -1);
	std::vector<size_t> use(s.floors.size(), 0);
	bool all_native = true;
	for (size_t c = 0; c < ch; c++) {
		const int fl = floor_of(modes[0], c);
		bool same = true;
		for (size_t m : modes)
			same = same && floor_of(m, c) == fl;
		if (!same) {
			if (!why)
				why = blockflag ? "long modes with different floors per channel" : "short modes with different floors per channel";
		} else if (s.floors[fl].type != 1) {
			if (!why)
				why = "floor type 0";
		} else if (s.floors[fl].f1.sorted_x.size() > max_posts) {
			if (!why)
				why = "a floor with more posts than the kernel's lanes take";
		} else if (!s.floors[fl].f1.sorted_x.empty() && s.floors[fl].f1.sorted_x.back() > LW_FLOOR_EXACT_ADX) {
			// (rangebits 13 .. 15, header.rs:871-873: posts up to x = 32768 whatever the block size.  A line between two active
			// posts may then span more bins than the kernels' two-FMA form of render_line is exact for -- proven for adx <= 4096,
			// tests/test_fast_model.py, test_big_model.py; round 6's random-setup campaign met a line of 32 638 bins that came out one
			// step off in a few bins.  k_prep evaluates such floors: its form falls back to the integer division beyond 4096.)
			if (!why)
				why = "a floor with posts beyond x = 4096";
		} else {
			native[c] = fl;
			use[fl]++;
			continue;
		}
		all_native = false;
	}
	std::vector<int> staged; // floors by use, most used first
	for (size_t f = 0; f < s.floors.size(); f++)
		if (use[f])
			staged.push_back((int)f);
	std::stable_sort(staged.begin(), staged.end(), [&](int a, int b) { return use[a] > use[b]; });
	const bool need_unit = !all_native || staged.size() > LW_FAST_MAX_FLOORS;
	if (staged.size() > LW_FAST_MAX_FLOORS && !why)
		why = "more distinct floor configurations than the kernel stages";
	const size_t keep = need_unit ? LW_FAST_MAX_FLOORS - 1 : LW_FAST_MAX_FLOORS;
	if (staged.size() > keep)
		staged.resize(keep);
	up.floor_slot.assign(s.floors.size(), -1);
	for (int f : staged) {
		up.floor_slot[f] = (int)up.n_staged;
		up.staged_F[up.n_staged++] = (uint8_t)s.floors[f].f1.sorted_x.size();
	}
	if (need_unit) {
		up.prep.unit_slot = (int)up.n_staged;
		up.staged_F[up.n_staged++] = 2;
	}
	std::vector<int> slot(ch, -1);
	for (size_t c = 0; c < ch; c++)
		slot[c] = native[c] >= 0 && up.floor_slot[native[c]] >= 0 ? up.floor_slot[native[c]] : up.prep.unit_slot;
	// A coupling list that is not disjoint pairs, everything else native (libvorbis' 5.1): the steps behind the disjoint prefix
	// inside the waves, if every unit's channels can be had with a short program (LwFastPlan::pre) -- no pre-pass then
	bool pre_on = false;
	std::vector<std::string> expr;
	if (allow_pre && why && !need_unit && coupling_same && !std::strcmp(why, "a channel takes part in more than one coupling step")) {
		size_t prefix = 0;
		std::vector<bool> seen(ch, false);
		while (prefix < ref.mag.size() && !seen[ref.mag[prefix]] && !seen[ref.ang[prefix]]) {
			seen[ref.mag[prefix]] = seen[ref.ang[prefix]] = true;
			prefix++;
		}
		pre_exprs(ref, ch, prefix, expr);
		std::fill(partner.begin(), partner.end(), -1);
		std::fill(role.begin(), role.end(), 0);
		for (size_t k = 0; k < prefix; k++) {
			partner[ref.mag[k]] = ref.ang[k];
			partner[ref.ang[k]] = ref.mag[k];
			role[ref.mag[k]] = 1;
			role[ref.ang[k]] = 2;
		}
		pre_on = true; // (withdrawn below if a unit's program does not exist)
	}
	up.prep.on = why != nullptr && !pre_on;
	up.prep.why = why ? why : "";
	if (up.prep.on) {
		up.prep.action.assign(nmodes * ch, LW_PREP_NONE);
		for (size_t m : modes)
			for (size_t c = 0; c < ch; c++) {
				const bool pm = slot[c] == up.prep.unit_slot && up.prep.unit_slot >= 0;
				up.prep.action[m * ch + c] = pm ? LW_PREP_PREMUL : LW_PREP_COPY;
				up.prep.premul = up.prep.premul || pm;
			}
		std::fill(partner.begin(), partner.end(), -1); // every coupling step is k_prep's
	}
	// Units: a coupling step's two channels share a wave (the inverse coupling needs both); the channels that are in no step
	// are paired up two by two as well, without coupling -- a wave works through two channels as one software pipeline, a
	// single channel leaves half of it empty (5.1: two coupled pairs + one uncoupled pair = 3 waves per packet instead of 4).
	std::vector<bool> done(ch, false);
	int pending_single = -1; // an uncoupled channel waiting for a partner
	for (size_t c = 0; c <= ch; c++) {
		LwFastUnit u{};
		if (c == ch) { // the odd one out stays a single-channel unit
			if (pending_single < 0)
				break;
			u.ch_a = (int8_t)pending_single;
			u.ch_b = -1;
			u.coupled = 0;
			pending_single = -1;
		} else if (done[c]) {
			continue;
		} else if (partner[c] >= 0) {
			const int m = role[c] == 1 ? (int)c : partner[c], a = role[c] == 1 ? partner[c] : (int)c;
			u.ch_a = (int8_t)m;
			u.ch_b = (int8_t)a;
			u.coupled = 1;
			done[m] = done[a] = true;
		} else if (pending_single < 0) {
			pending_single = (int)c;
			done[c] = true;
			continue;
		} else {
			u.ch_a = (int8_t)pending_single;
			u.ch_b = (int8_t)c;
			u.coupled = 0;
			pending_single = -1;
			done[c] = true;
		}
		const int sa = slot[(size_t)u.ch_a], sb = u.ch_b >= 0 ? slot[(size_t)u.ch_b] : 0;
		u.floor_a = (uint8_t)sa;
		u.floor_b = (uint8_t)sb;
		u.F_a = up.staged_F[sa];
		u.F_b = up.staged_F[sb];
		up.units.push_back(u);
	}
	if (up.units.size() > LW_FAST_WAVES)
		return "more units than waves in a workgroup";
	if (pre_on) {
		for (const LwFastUnit &u : up.units) {
			uint32_t prog = 0;
			if (!pre_program(expr, u.ch_a, u.ch_b, prog)) {
				// not with two more channels and three steps: the pre-pass after all (plan again without the attempt)
				up = UnitPlan();
				return plan_units(id, s, blockflag, max_posts, up, false);
			}
			up.pre.push_back(prog);
		}
		up.prep.why = why; // (census: what the waves do themselves)
	}
	return nullptr;
}

// xsf / sid16 of the unit floor (LwPrepPlan): posts at x = 0 and x = n / 2, every bin in interval 0
static void fill_unit_floor(float *xsf64, uint16_t *sid, size_t n_sid, uint32_t n2)
{
	for (size_t i = 0; i < 64; i++)
		xsf64[i] = i == 0 ? 0.0f : i == 1 ? (float)n2 : std::numeric_limits<float>::infinity();
	for (size_t i = 0; i < n_sid; i++)
		sid[i] = 0;
}

void build_fast_plan(const Ident &id, const Setup &s, LwFastPlan &plan)
{
	plan = LwFastPlan();
	if (id.bs1 != LW_FAST_BS) {
		plan.why_not = "blocksize_1 is not 11";
		return;
	}
	const uint32_t n = 1u << id.bs1, n2 = n / 2, n8 = n / 8;
	UnitPlan up;
	if (const char *why = plan_units(id, s, true, 64, up, true)) {
		plan.why_not = why;
		return;
	}
	std::memcpy(plan.long_mode_mask, up.mode_mask, sizeof(plan.long_mode_mask));
	plan.units = up.units;
	plan.prep = up.prep;
	plan.pre = up.pre;
	// the same units with every channel pair split over two waves
	for (const LwFastUnit &u : up.units) {
		if (u.ch_b < 0) {
			plan.units_split.push_back(u);
			continue;
		}
		LwFastUnit a = u, b = u;
		a.coupled = u.coupled ? LW_UNIT_SPLIT_MAG : 0; // ch_a stays; ch_b only as the partner of the coupling step
		if (!u.coupled)
			a.ch_b = -1;
		b.ch_a = u.ch_b;
		b.ch_b = u.coupled ? u.ch_a : (int8_t)-1;
		b.coupled = u.coupled ? LW_UNIT_SPLIT_ANG : 0;
		b.floor_a = u.floor_b;
		b.F_a = u.F_b;
		b.floor_b = u.floor_a;
		b.F_b = u.F_a;
		plan.units_split.push_back(a);
		plan.units_split.push_back(b);
	}
	if (plan.units_split.size() > LW_FAST_WAVES || !plan.pre.empty()) // (PRE: whole pairs only)
		plan.units_split.clear();
	plan.n_staged_floors = up.n_staged;
	std::memcpy(plan.staged_floor_F, up.staged_F, sizeof(plan.staged_floor_F));
	const std::vector<int> &floor_slot = up.floor_slot;

	// ---- LDS image
	const BlocksizeTables &t = id.tab[1];
	const float *A = t.A.data(), *B = t.B.data(), *C = t.C.data(), *W = t.window.data();
	Writer w{plan.image};
	LwFastImage &o = plan.off;
	o.apair = w.reserve(n2 * 4);
	std::memcpy(w.f(o.apair), A, n2 * 4);
	o.tw_s2 = w.reserve(4 * 64 * 8);
	for (uint32_t x = 0; x < 4; x++)
		for (uint32_t l = 0; l < 64; l++) {
			const uint32_t p = 64 * x + l, a = n2 - 4 - 4 * p;
			w.f(o.tw_s2)[2 * (64 * x + l)] = A[a];
			w.f(o.tw_s2)[2 * (64 * x + l) + 1] = A[a + 1];
		}
	o.tw_l0 = w.reserve(2 * 64 * 8);
	for (uint32_t b = 0; b < 2; b++)
		for (uint32_t l = 0; l < 64; l++) {
			const uint32_t r = 127 - 64 * b - l;
			w.f(o.tw_l0)[2 * (64 * b + l)] = A[8 * r];
			w.f(o.tw_l0)[2 * (64 * b + l) + 1] = A[8 * r + 1];
		}
	o.tw_l1 = w.reserve(64 * 8);
	for (uint32_t l = 0; l < 64; l++) {
		const uint32_t r = 63 - l;
		w.f(o.tw_l1)[2 * l] = A[16 * r];
		w.f(o.tw_l1)[2 * l + 1] = A[16 * r + 1];
	}
	o.tw_l2 = w.reserve(4 * 8 * 8);
	for (uint32_t yy = 0; yy < 4; yy++)
		for (uint32_t lo3 = 0; lo3 < 8; lo3++) {
			const uint32_t r = 31 - (8 * yy + lo3);
			w.f(o.tw_l2)[2 * (8 * yy + lo3)] = A[32 * r];
			w.f(o.tw_l2)[2 * (8 * yy + lo3) + 1] = A[32 * r + 1];
		}
	o.tw_l3 = w.reserve(2 * 8 * 8);
	for (uint32_t b = 0; b < 2; b++)
		for (uint32_t lo3 = 0; lo3 < 8; lo3++) {
			const uint32_t r = 15 - (8 * b + lo3);
			w.f(o.tw_l3)[2 * (8 * b + lo3)] = A[64 * r];
			w.f(o.tw_l3)[2 * (8 * b + lo3) + 1] = A[64 * r + 1];
		}
	o.tw_l4 = w.reserve(8 * 8);
	for (uint32_t lo3 = 0; lo3 < 8; lo3++) {
		const uint32_t r = 7 - lo3;
		w.f(o.tw_l4)[2 * lo3] = A[128 * r];
		w.f(o.tw_l4)[2 * lo3 + 1] = A[128 * r + 1];
	}
	o.a2 = w.reserve(16);
	w.f(o.a2)[0] = A[n8];
	o.c4 = w.reserve(2 * 64 * 16);
	o.b_lo = w.reserve(2 * 64 * 16);
	o.b_hi = w.reserve(2 * 64 * 16);
	o.win = w.reserve(2 * 64 * 32);
	for (uint32_t c = 0; c < 2; c++)
		for (uint32_t l = 0; l < 64; l++) {
			const uint32_t mp = 2 * l + c, e = 64 * c + l;
			for (uint32_t j = 0; j < 4; j++) {
				w.f(o.c4)[4 * e + j] = C[4 * mp + j];
				w.f(o.b_lo)[4 * e + j] = B[4 * mp + j];
				w.f(o.b_hi)[4 * e + j] = B[4 * (255 - mp) + j];
			}
			const uint32_t q[4] = {511 - 2 * mp, 510 - 2 * mp, 1 + 2 * mp, 2 * mp};
			for (uint32_t k = 0; k < 4; k++) {
				w.f(o.win)[8 * e + 2 * k] = W[q[k]];
				w.f(o.win)[8 * e + 2 * k + 1] = W[n2 - 1 - q[k]];
			}
		}
	o.inv_db = w.reserve(256 * 4); // filled by the runtime (it owns the spec table)
	o.xsf = w.reserve(LW_FAST_MAX_FLOORS * 64 * 4);
	o.sid16 = w.reserve(LW_FAST_MAX_FLOORS * 4 * 64 * 4 * 2);
	for (size_t fl = 0; fl < s.floors.size(); fl++) {
		const int slot = floor_slot[fl];
		if (slot < 0)
			continue;
		const Floor1 &f1 = s.floors[fl].f1;
		const size_t F = f1.sorted_x.size();
		for (size_t i = 0; i < 64; i++)
			w.f(o.xsf)[64 * slot + i] = i < F ? (float)f1.sorted_x[i] : std::numeric_limits<float>::infinity();
		for (uint32_t x = 0; x < 4; x++)
			for (uint32_t l = 0; l < 64; l++)
				for (uint32_t j = 0; j < 4; j++) {
					const uint32_t k = 4 * (64 * x + l) + j;
					size_t sidx = 0; // largest s with xs[s] <= k (xs[0] = 0)
					while (sidx + 1 < F && f1.sorted_x[sidx + 1] <= k)
						sidx++;
					w.h(o.sid16)[((slot * 4 + x) * 64 + l) * 4 + j] = (uint16_t)(16 * sidx);
				}
	}
	if (up.prep.unit_slot >= 0)
		fill_unit_floor(w.f(o.xsf) + 64 * up.prep.unit_slot, w.h(o.sid16) + (size_t)up.prep.unit_slot * 4 * 64 * 4, 4 * 64 * 4, n2);
	const size_t quantum = 1024;
	o.total = (uint32_t)((plan.image.size() + quantum - 1) / quantum * quantum);
	plan.image.resize(o.total, 0);
	if (o.apair != LWI_APAIR || o.tw_s2 != LWI_TW_S2 || o.tw_l0 != LWI_TW_L0 || o.tw_l1 != LWI_TW_L1 || o.tw_l2 != LWI_TW_L2 ||
			o.tw_l3 != LWI_TW_L3 || o.tw_l4 != LWI_TW_L4 || o.a2 != LWI_A2 || o.c4 != LWI_C4 || o.b_lo != LWI_B_LO ||
			o.b_hi != LWI_B_HI || o.win != LWI_WIN || o.inv_db != LWI_INV_DB || o.xsf != LWI_XSF || o.sid16 != LWI_SID16 ||
			o.total != LWI_TOTAL) {
		plan.why_not = "LDS image layout differs from the kernel's compile-time layout (LWI_*)";
		return;
	}
	plan.eligible = true;
}

namespace {
// the LDS image of k_short<L> (index conventions of lw_fast.hpp: l = lane inside the block, 0 .. L-1)
template <int L>
void fill_blk_image(const BlocksizeTables &t, const Setup &s, const std::vector<int> &floor_slot, int unit_slot, std::vector<uint8_t> &image)
{
	typedef LwBlkLayout<L> Y;
	const uint32_t P = 8u * L, n = 4u * P, n2 = n / 2, n8 = n / 8;
	const float *A = t.A.data(), *B = t.B.data(), *C = t.C.data(), *W = t.window.data();
	image.assign(Y::TOTAL, 0);
	auto f = [&](uint32_t off) { return reinterpret_cast<float *>(image.data() + off); };
	auto h = [&](uint32_t off) { return reinterpret_cast<uint16_t *>(image.data() + off); };
	std::memcpy(f(Y::APAIR), A, n2 * 4);
	for (uint32_t x = 0; x < 4; x++)
		for (uint32_t l = 0; l < L; l++) {
			const uint32_t p = L * x + l, a = n2 - 4 - 4 * p; // imdct.rs:385-430
			f(Y::TW_S2)[2 * p] = A[a];
			f(Y::TW_S2)[2 * p + 1] = A[a + 1];
		}
	for (uint32_t b = 0; b < 2; b++)
		for (uint32_t l = 0; l < L; l++) {
			const uint32_t r = 2 * L - 1 - L * b - l; // imdct.rs:445-446: A[8 r]
			f(Y::TW_L0)[2 * (L * b + l)] = A[8 * r];
			f(Y::TW_L0)[2 * (L * b + l) + 1] = A[8 * r + 1];
		}
	for (uint32_t l = 0; l < L; l++) {
		const uint32_t r = L - 1 - l; // imdct.rs:449-452: A[16 r]
		f(Y::TW_L1)[2 * l] = A[16 * r];
		f(Y::TW_L1)[2 * l + 1] = A[16 * r + 1];
	}
	// stages l >= 2 (imdct.rs:454-477): twiddle A[(8 << l) r], r = the complement of the pair bits below the butterfly's bit
	if (L == 32) {
		for (uint32_t b = 0; b < 2; b++)
			for (uint32_t lo3 = 0; lo3 < 8; lo3++) {
				const uint32_t r = 15 - (8 * b + lo3);
				f(Y::TW_C2)[2 * (8 * b + lo3)] = A[32 * r];
				f(Y::TW_C2)[2 * (8 * b + lo3) + 1] = A[32 * r + 1];
			}
		for (uint32_t lo3 = 0; lo3 < 8; lo3++) {
			const uint32_t r = 7 - lo3;
			f(Y::TW_C3)[2 * lo3] = A[64 * r];
			f(Y::TW_C3)[2 * lo3 + 1] = A[64 * r + 1];
		}
	} else if (L == 16) {
		for (uint32_t lo3 = 0; lo3 < 8; lo3++) {
			const uint32_t r = 7 - lo3;
			f(Y::TW_C2)[2 * lo3] = A[32 * r];
			f(Y::TW_C2)[2 * lo3 + 1] = A[32 * r + 1];
		}
	}
	f(Y::A2)[0] = A[n8];
	for (uint32_t c = 0; c < 2; c++)
		for (uint32_t l = 0; l < L; l++) {
			const uint32_t mp = 2 * l + c, e = L * c + l;
			for (uint32_t j = 0; j < 4; j++) {
				f(Y::C4)[4 * e + j] = C[4 * mp + j];
				f(Y::B_LO)[4 * e + j] = B[4 * mp + j];
				f(Y::B_HI)[4 * e + j] = B[4 * (P / 2 - 1 - mp) + j];
			}
			const uint32_t q[4] = {P - 1 - 2 * mp, P - 2 - 2 * mp, 1 + 2 * mp, 2 * mp};
			for (uint32_t k = 0; k < 4; k++) {
				f(Y::WIN)[8 * e + 2 * k] = W[q[k]];
				f(Y::WIN)[8 * e + 2 * k + 1] = W[n2 - 1 - q[k]];
			}
		}
	// (INV_DB is filled by the runtime: it owns the spec table)
	for (size_t fl = 0; fl < s.floors.size(); fl++) {
		const int slot = floor_slot[fl];
		if (slot < 0)
			continue;
		const Floor1 &f1 = s.floors[fl].f1;
		const size_t F = f1.sorted_x.size();
		for (size_t i = 0; i < 64; i++)
			f(Y::XSF)[64 * slot + i] = i < F ? (float)f1.sorted_x[i] : std::numeric_limits<float>::infinity();
		for (uint32_t x = 0; x < 4; x++)
			for (uint32_t l = 0; l < L; l++)
				for (uint32_t j = 0; j < 4; j++) {
					const uint32_t k = 4 * (L * x + l) + j;
					size_t sidx = 0; // largest s with xs[s] <= k (xs[0] = 0)
					while (sidx + 1 < F && f1.sorted_x[sidx + 1] <= k)
						sidx++;
					h(Y::SID16)[((slot * 4 + x) * L + l) * 4 + j] = (uint16_t)(16 * sidx);
				}
	}
	if (unit_slot >= 0)
		fill_unit_floor(f(Y::XSF) + 64 * unit_slot, h(Y::SID16) + (size_t)unit_slot * 4 * L * 4, 4 * L * 4, n2);
}
} // namespace

namespace {
// one channel per wave: every channel pair of `units` as two LW_UNIT_SPLIT_* halves (see build_fast_plan)
void split_units(const std::vector<LwFastUnit> &units, std::vector<LwFastUnit> &out)
{
	out.clear();
	for (const LwFastUnit &u : units) {
		if (u.ch_b < 0) {
			out.push_back(u);
			continue;
		}
		LwFastUnit a = u, b = u;
		a.coupled = u.coupled ? LW_UNIT_SPLIT_MAG : 0; // ch_a stays; ch_b only as the partner of the coupling step
		if (!u.coupled)
			a.ch_b = -1;
		b.ch_a = u.ch_b;
		b.ch_b = u.coupled ? u.ch_a : (int8_t)-1;
		b.coupled = u.coupled ? LW_UNIT_SPLIT_ANG : 0;
		b.floor_a = u.floor_b;
		b.F_a = u.F_b;
		b.floor_b = u.floor_a;
		b.F_b = u.F_a;
		out.push_back(a);
		out.push_back(b);
	}
}

// the LDS image and the HBM interval table of k_long12 (LwL12Layout)
void fill_l12_image(const BlocksizeTables &t, const Setup &s, const std::vector<int> &floor_slot, int unit_slot, std::vector<uint8_t> &image,
		std::vector<uint8_t> &sid)
{
	typedef LwL12Layout Y;
	const uint32_t n = 4096, n2 = n / 2, n8 = n / 8;
	const float *A = t.A.data(), *B = t.B.data(), *C = t.C.data(), *W = t.window.data();
	image.assign(Y::TOTAL, 0);
	sid.assign(Y::SID_BYTES, 0);
	auto f = [&](uint32_t off) { return reinterpret_cast<float *>(image.data() + off); };
	auto put = [&](uint32_t off, uint32_t idx, uint32_t a) { // float2 number idx of a table <- (A[a], A[a + 1])
		f(off)[2 * idx] = A[a];
		f(off)[2 * idx + 1] = A[a + 1];
	};
	for (uint32_t x = 0; x < 8; x++)
		for (uint32_t l = 0; l < 64; l++)
			put(Y::TW_S2, 64 * x + l, n2 - 4 - 4 * (64 * x + l)); // imdct.rs:385-430
	// stages l (imdct.rs:445-477): twiddle A[(8 << l) r], r = the complement of the pair bits below the butterfly's bit
	for (uint32_t y = 0; y < 4; y++)
		for (uint32_t l = 0; l < 64; l++)
			put(Y::TW_L0, 64 * y + l, 8 * (255 - (64 * y + l)));
	for (uint32_t b = 0; b < 2; b++)
		for (uint32_t l = 0; l < 64; l++)
			put(Y::TW_L1, 64 * b + l, 16 * (127 - (64 * b + l)));
	for (uint32_t l = 0; l < 64; l++)
		put(Y::TW_L2, l, 32 * (63 - l));
	for (uint32_t yy = 0; yy < 4; yy++)
		for (uint32_t lo3 = 0; lo3 < 8; lo3++)
			put(Y::TW_L3, 8 * yy + lo3, 64 * (31 - (8 * yy + lo3)));
	for (uint32_t b = 0; b < 2; b++)
		for (uint32_t lo3 = 0; lo3 < 8; lo3++)
			put(Y::TW_L4, 8 * b + lo3, 128 * (15 - (8 * b + lo3)));
	for (uint32_t lo3 = 0; lo3 < 8; lo3++)
		put(Y::TW_L5, lo3, 256 * (7 - lo3));
	f(Y::A2)[0] = A[n8];
	for (uint32_t k4 = 0; k4 < 4; k4++)
		for (uint32_t l = 0; l < 64; l++) {
			const uint32_t mp = 128 * (k4 >> 1) + 2 * l + (k4 & 1), e = 64 * k4 + l;
			for (uint32_t j = 0; j < 4; j++) {
				f(Y::C4)[4 * e + j] = C[4 * mp + j];
				f(Y::B_LO)[4 * e + j] = B[4 * mp + j];
				f(Y::B_HI)[4 * e + j] = B[4 * (511 - mp) + j];
			}
			const uint32_t q[4] = {1023 - 2 * mp, 1022 - 2 * mp, 1 + 2 * mp, 2 * mp};
			for (uint32_t k = 0; k < 4; k++) {
				f(Y::WIN)[8 * e + 2 * k] = W[q[k]];
				f(Y::WIN)[8 * e + 2 * k + 1] = W[n2 - 1 - q[k]];
			}
		}
	// (INV_DB is filled by the runtime: it owns the spec table)
	uint16_t *h = reinterpret_cast<uint16_t *>(sid.data());
	for (size_t fl = 0; fl < s.floors.size(); fl++) {
		const int slot = floor_slot[fl];
		if (slot < 0)
			continue;
		const Floor1 &f1 = s.floors[fl].f1;
		const size_t F = f1.sorted_x.size();
		for (size_t i = 0; i < 64; i++)
			f(Y::XSF)[64 * slot + i] = i < F ? (float)f1.sorted_x[i] : std::numeric_limits<float>::infinity();
		for (uint32_t x = 0; x < 8; x++)
			for (uint32_t l = 0; l < 64; l++)
				for (uint32_t j = 0; j < 4; j++) {
					const uint32_t k = 4 * (64 * x + l) + j;
					size_t sidx = 0; // largest s with xs[s] <= k (xs[0] = 0)
					while (sidx + 1 < F && f1.sorted_x[sidx + 1] <= k)
						sidx++;
					h[((slot * 8 + x) * 64 + l) * 4 + j] = (uint16_t)(16 * sidx);
				}
	}
	if (unit_slot >= 0)
		fill_unit_floor(f(Y::XSF) + 64 * unit_slot, h + (size_t)unit_slot * 8 * 64 * 4, 8 * 64 * 4, n2);
}
} // namespace

void build_blk_plan(const Ident &id, const Setup &s, bool blockflag, const LwFastPlan &fast, LwShortPlan &plan)
{
	plan = LwShortPlan();
	const uint32_t bs = blockflag ? id.bs1 : id.bs0;
	if (blockflag && fast.eligible) {
		plan.why_not = "long blocks run through k_long";
		return;
	}
	if (!blockflag && lw_unified_classes(id, s)) {
		plan.why_not = "blocksize_0 = blocksize_1: served with the long blocks";
		return;
	}
	const bool big = blockflag && bs >= LW_BIG_MIN_BS && bs <= LW_BIG_MAX_BS; // k_big<BS>: a workgroup of n / 32 threads per block
	if ((bs < LW_BLK_MIN_BS || bs > LW_BLK_MAX_BS) && !big) {
		plan.why_not = "block size outside 256 .. 1024 (long blocks: and not 4096 / 8192)";
		return;
	}
	plan.bs = bs;
	plan.lanes = 1u << (bs - 5);
	plan.passes = big ? LW_BIG_MAX_PASSES : plan.lanes == 32 ? 3 : plan.lanes == 16 ? 2 : 1;
	UnitPlan up;
	// (4096 points: k_long12 takes a floor's posts as the lanes of one wave, at most 64 -- a floor of 65 goes on the unit floor)
	if (const char *why = plan_units(id, s, blockflag, big ? (bs == 12 ? 64 : 65) : LW_BLK_MAX_POSTS(plan.lanes), up)) {
		plan.why_not = why;
		return;
	}
	plan.prep = up.prep;
	for (size_t fl = 0; fl < up.floor_slot.size(); fl++)
		if (up.floor_slot[fl] >= 0)
			plan.fl_of[up.floor_slot[fl]] = (uint32_t)fl;
	if (up.prep.unit_slot >= 0) // (k_big reads the posts' x from T.floor_x: the unit floors are rows n_floors + block class there)
		plan.fl_of[up.prep.unit_slot] = (uint32_t)(s.floors.size() + (blockflag ? 1 : 0));
	std::memcpy(plan.short_mode_mask, up.mode_mask, sizeof(plan.short_mode_mask));
	plan.units = up.units;
	plan.n_staged_floors = up.n_staged;
	std::memcpy(plan.staged_floor_F, up.staged_F, sizeof(plan.staged_floor_F));
	if (big) { // (k_big: no LDS image, the tables of the block size stay in HBM / L2)
		if (bs == 12) { // k_long12: a floor's posts are lanes of ONE wave (at most 64), one channel per wave
			bool posts_ok = true;
			for (uint32_t i = 0; i < up.n_staged; i++)
				posts_ok = posts_ok && up.staged_F[i] <= 64;
			split_units(up.units, plan.units_split);
			if (posts_ok && plan.units_split.size() <= LW_FAST_WAVES)
				fill_l12_image(id.tab[1], s, up.floor_slot, up.prep.unit_slot, plan.image, plan.sid12);
			else
				plan.units_split.clear();
		}
		plan.eligible = true;
		return;
	}
	const BlocksizeTables &t = id.tab[blockflag ? 1 : 0];
	if (plan.lanes == 8)
		fill_blk_image<8>(t, s, up.floor_slot, up.prep.unit_slot, plan.image);
	else if (plan.lanes == 16)
		fill_blk_image<16>(t, s, up.floor_slot, up.prep.unit_slot, plan.image);
	else
		fill_blk_image<32>(t, s, up.floor_slot, up.prep.unit_slot, plan.image);
	plan.eligible = true;
}

} // namespace lw
