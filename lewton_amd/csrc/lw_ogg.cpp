// Ogg container either side of the hot path: the page / packet demultiplexer lewton takes from the
// external crate `ogg` 0.8.0 (restated from RFC 3533, the crate is not part of the reference tree) and lewton's own
// `OggStreamReader` (src/inside_ogg.rs:30-314) on top of this library's packet decoder.  Host code only; everything
// below the packet boundary goes through the public C ABI of include/lewton_amd.h.
#include "../../include/lewton_amd.h"

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

void lw_set_device_error(const std::string &msg); // lw_runtime.cpp: thread-local text behind lw_last_device_error()

namespace {

// ---------------------------------------------------------------------------------------------
// byte sources (the reference's `T: Read + Seek`)
// ---------------------------------------------------------------------------------------------
struct Source {
	virtual ~Source() {}
	virtual int64_t read(uint8_t *dst, size_t n) = 0; // bytes read, 0 at the end, < 0 on error
	virtual int64_t seek(int64_t off, int whence) = 0; // new absolute position or < 0
};

struct MemSource : Source {
	const uint8_t *p = nullptr;
	size_t len = 0, pos = 0;
	std::vector<uint8_t> own;
	int64_t read(uint8_t *dst, size_t n) override
	{
		const size_t k = std::min(n, len - pos);
		if (k)
			std::memcpy(dst, p + pos, k);
		pos += k;
		return (int64_t)k;
	}
	int64_t seek(int64_t off, int whence) override
	{
		const int64_t base = whence == 0 ? 0 : whence == 1 ? (int64_t)pos : (int64_t)len;
		const int64_t np = base + off;
		if (np < 0)
			return -1;
		pos = (size_t)std::min<int64_t>(np, (int64_t)len);
		return (int64_t)pos;
	}
};

struct FileSource : Source {
	FILE *f = nullptr;
	~FileSource() override
	{
		if (f)
			std::fclose(f);
	}
	int64_t read(uint8_t *dst, size_t n) override
	{
		const size_t k = std::fread(dst, 1, n, f);
		if (k < n && std::ferror(f))
			return -1;
		return (int64_t)k;
	}
	int64_t seek(int64_t off, int whence) override
	{
		if (fseeko(f, (off_t)off, whence == 0 ? SEEK_SET : whence == 1 ? SEEK_CUR : SEEK_END) != 0)
			return -1;
		return (int64_t)ftello(f);
	}
};

struct IoSource : Source {
	lw_ogg_io io{};
	int64_t read(uint8_t *dst, size_t n) override { return io.read(io.user, dst, n); }
	int64_t seek(int64_t off, int whence) override { return io.seek(io.user, off, whence); }
};

// ---------------------------------------------------------------------------------------------
// CRC of RFC 3533 section 6: polynomial 0x04c11db7, initial value 0, bits not reflected, no final xor
// ---------------------------------------------------------------------------------------------
// Slicing by 8: t[k][b] is the CRC of byte b followed by k zero bytes, so eight input bytes are folded with eight
// independent table reads per step instead of eight dependent ones (the demultiplexer is the sequential stage of the
// single-stream path; bytewise it ran at ~0.3 GB/s).
struct CrcTable {
	uint32_t t[8][256];
	CrcTable()
	{
		for (uint32_t i = 0; i < 256; i++) {
			uint32_t r = i << 24;
			for (int k = 0; k < 8; k++)
				r = (r & 0x80000000u) ? (r << 1) ^ 0x04c11db7u : (r << 1);
			t[0][i] = r;
		}
		for (int k = 1; k < 8; k++)
			for (uint32_t i = 0; i < 256; i++)
				t[k][i] = (t[k - 1][i] << 8) ^ t[0][t[k - 1][i] >> 24];
	}
};
const CrcTable kCrc;

uint32_t crc_update(uint32_t crc, const uint8_t *d, size_t n)
{
	size_t i = 0;
	for (; i + 8 <= n; i += 8) {
		uint32_t a, b;
		std::memcpy(&a, d + i, 4);
		std::memcpy(&b, d + i + 4, 4);
		const uint32_t hi = crc ^ __builtin_bswap32(a), lo = __builtin_bswap32(b); // the CRC shifts the first byte in first
		crc = kCrc.t[7][hi >> 24] ^ kCrc.t[6][(hi >> 16) & 0xff] ^ kCrc.t[5][(hi >> 8) & 0xff] ^ kCrc.t[4][hi & 0xff] ^
		      kCrc.t[3][lo >> 24] ^ kCrc.t[2][(lo >> 16) & 0xff] ^ kCrc.t[1][(lo >> 8) & 0xff] ^ kCrc.t[0][lo & 0xff];
	}
	for (; i < n; i++)
		crc = (crc << 8) ^ kCrc.t[0][((crc >> 24) ^ d[i]) & 0xff];
	return crc;
}

struct Page {
	int64_t offset = 0; // of the capture pattern
	size_t size = 0;    // header + lacing table + body
	bool continued = false, bos = false, eos = false;
	uint64_t absgp = 0;
	uint32_t serial = 0, seq = 0;
	std::vector<uint8_t> lacing, body;
	bool completes_packet() const
	{
		for (uint8_t lv : lacing)
			if (lv < 255)
				return true;
		return false;
	}
};

struct QueuedPacket {
	std::vector<uint8_t> data;
	uint32_t serial = 0;
	uint64_t absgp_page = 0;
	bool first_in_stream = false, last_in_stream = false, first_in_page = false, last_in_page = false;
};

} // namespace

struct lw_ogg_reader {
	std::unique_ptr<Source> src;
	int64_t pos = 0; // byte offset of the next page
	std::unordered_map<uint32_t, std::vector<uint8_t>> partial; // per logical stream: packet continued on the next page
	std::deque<QueuedPacket> queue;
	QueuedPacket current; // storage behind the last lw_ogg_packet handed out
	Page pump_page;       // scratch of pump()

	// reads exactly n bytes at the current source position; 0 ok, 1 clean end before the first byte, < 0 error
	int read_exact(uint8_t *dst, size_t n)
	{
		size_t got = 0;
		while (got < n) {
			const int64_t k = src->read(dst + got, n - got);
			if (k < 0)
				return -1;
			if (k == 0)
				return got == 0 ? 1 : -1;
			got += (size_t)k;
		}
		return 0;
	}

	// the page at byte offset `at`; LW_OK, LW_OGG_EOF (no byte left) or an OggReadError code
	int read_page_at(int64_t at, Page &pg)
	{
		if (src->seek(at, 0) != at)
			return LW_OGG_READ_ERROR;
		uint8_t h[27 + 255];
		int rc = read_exact(h, 27);
		if (rc == 1)
			return LW_OGG_EOF;
		if (rc < 0)
			return LW_OGG_READ_ERROR;
		if (std::memcmp(h, "OggS", 4) != 0)
			return LW_OGG_NO_CAPTURE_PATTERN;
		if (h[4] != 0)
			return LW_OGG_INVALID_STREAM_STRUCT_VER;
		const uint8_t nseg = h[26];
		if (nseg && read_exact(h + 27, nseg) != 0)
			return LW_OGG_READ_ERROR;
		size_t body = 0;
		for (int i = 0; i < nseg; i++)
			body += h[27 + i];
		pg.body.resize(body);
		if (body && read_exact(pg.body.data(), body) != 0)
			return LW_OGG_READ_ERROR;
		uint32_t stored;
		std::memcpy(&stored, h + 22, 4); // little endian host (x86-64, the only host of this library)
		std::memset(h + 22, 0, 4);
		uint32_t crc = crc_update(0, h, 27 + (size_t)nseg);
		crc = crc_update(crc, pg.body.data(), body);
		if (crc != stored)
			return LW_OGG_HASH_MISMATCH;
		pg.offset = at;
		pg.size = 27 + (size_t)nseg + body;
		pg.continued = (h[5] & 1) != 0;
		pg.bos = (h[5] & 2) != 0;
		pg.eos = (h[5] & 4) != 0;
		std::memcpy(&pg.absgp, h + 6, 8);
		std::memcpy(&pg.serial, h + 14, 4);
		std::memcpy(&pg.seq, h + 18, 4);
		pg.lacing.assign(h + 27, h + 27 + nseg);
		return LW_OK;
	}

	// splits the next page into packets (RFC 3533 section 5: a lacing value < 255 ends a packet)
	int pump()
	{
		Page &pg = pump_page; // (its vectors keep their capacity from page to page)
		const int rc = read_page_at(pos, pg);
		if (rc != LW_OK)
			return rc;
		pos = pg.offset + (int64_t)pg.size;
		std::vector<uint8_t> buf;
		bool have_start = true;
		auto it = partial.find(pg.serial);
		if (it != partial.end()) {
			if (pg.continued)
				buf.swap(it->second);
			partial.erase(it); // an unfinished packet is dropped when the next page does not continue it
		} else if (pg.continued) {
			have_start = false; // continuation of a packet whose beginning was not seen (after a seek)
		}
		size_t o = 0, n_done = 0;
		for (size_t li = 0; li < pg.lacing.size(); li++) {
			const uint8_t lv = pg.lacing[li];
			if (have_start) {
				if (buf.empty()) { // one allocation per packet: the segments of the packet that lie on this page
					size_t len = 0;
					for (size_t k = li; k < pg.lacing.size(); k++) {
						len += pg.lacing[k];
						if (pg.lacing[k] < 255)
							break;
					}
					buf.reserve(len);
				}
				buf.insert(buf.end(), pg.body.begin() + o, pg.body.begin() + o + lv);
			}
			o += lv;
			if (lv < 255) {
				if (have_start) {
					QueuedPacket q;
					q.data.swap(buf);
					q.serial = pg.serial;
					q.absgp_page = pg.absgp;
					q.first_in_page = n_done == 0;
					q.first_in_stream = pg.bos && n_done == 0;
					queue.push_back(std::move(q));
					n_done++;
				}
				have_start = true;
				buf.clear();
			}
		}
		if (n_done) {
			queue.back().last_in_page = true;
			queue.back().last_in_stream = pg.eos;
		}
		if (!pg.lacing.empty() && pg.lacing.back() == 255 && have_start)
			partial[pg.serial].swap(buf);
		return LW_OK;
	}

	// the next packet with its storage (the stream layer keeps packets in its look-ahead queue: no second copy)
	int next_owned(QueuedPacket &q)
	{
		while (queue.empty()) {
			const int rc = pump();
			if (rc != LW_OK)
				return rc;
		}
		q = std::move(queue.front());
		queue.pop_front();
		return LW_OK;
	}

	int next(lw_ogg_packet *out)
	{
		while (queue.empty()) {
			const int rc = pump();
			if (rc != LW_OK)
				return rc;
		}
		current = std::move(queue.front());
		queue.pop_front();
		out->data = current.data.data();
		out->len = current.data.size();
		out->stream_serial = current.serial;
		out->absgp_page = current.absgp_page;
		out->first_in_stream = current.first_in_stream;
		out->last_in_stream = current.last_in_stream;
		out->first_in_page = current.first_in_page;
		out->last_in_page = current.last_in_page;
		return LW_OK;
	}

	void forget()
	{
		queue.clear();
		partial.clear();
	}

	// first valid page at or after byte offset `from` and before `limit`; LW_OK / LW_OGG_EOF (none) / read error
	int find_page(int64_t from, int64_t limit, Page &pg)
	{
		std::vector<uint8_t> win(65536 + 3);
		int64_t at = from;
		while (at < limit) {
			if (src->seek(at, 0) != at)
				return LW_OGG_READ_ERROR;
			const size_t want = (size_t)std::min<int64_t>((int64_t)win.size(), limit + 3 - at);
			size_t got = 0;
			while (got < want) {
				const int64_t k = src->read(win.data() + got, want - got);
				if (k < 0)
					return LW_OGG_READ_ERROR;
				if (k == 0)
					break;
				got += (size_t)k;
			}
			if (got < 4)
				return LW_OGG_EOF;
			for (size_t i = 0; i + 4 <= got; i++) {
				if (at + (int64_t)i >= limit)
					return LW_OGG_EOF;
				if (win[i] == 'O' && win[i + 1] == 'g' && win[i + 2] == 'g' && win[i + 3] == 'S') {
					const int rc = read_page_at(at + (int64_t)i, pg);
					if (rc == LW_OK)
						return LW_OK;
				}
			}
			at += (int64_t)got - 3;
			if (got < want)
				return LW_OGG_EOF;
		}
		return LW_OGG_EOF;
	}

	int seek_absgp(bool has_serial, uint32_t serial, uint64_t goal)
	{
		const int64_t size = src->seek(0, 2);
		if (size < 0)
			return LW_OGG_READ_ERROR;
		auto matches = [&](const Page &pg) {
			return (!has_serial || pg.serial == serial) && pg.absgp != ~0ull && pg.completes_packet();
		};
		// invariant: every matching page that starts before `lo` has absgp <= goal (best_end = end of the last of them);
		// every matching page that starts at or after `hi` has absgp > goal.  Granule positions of one logical stream
		// never decrease, which is what makes the bisection valid.
		int64_t lo = 0, hi = size, best_end = 0;
		Page pg;
		while (hi - lo > 65536) {
			const int64_t mid = lo + (hi - lo) / 2;
			int64_t at = mid;
			bool found = false;
			for (;;) {
				const int rc = find_page(at, hi, pg);
				if (rc == LW_OGG_EOF)
					break;
				if (rc != LW_OK)
					return rc;
				if (matches(pg)) {
					found = true;
					break;
				}
				at = pg.offset + (int64_t)pg.size;
			}
			if (!found) {
				hi = mid;
			} else if (pg.absgp <= goal) {
				lo = pg.offset + (int64_t)pg.size;
				best_end = lo;
			} else {
				hi = mid;
			}
		}
		int64_t at = lo;
		for (;;) {
			const int rc = find_page(at, hi, pg);
			if (rc == LW_OGG_EOF)
				break;
			if (rc != LW_OK)
				return rc;
			if (matches(pg)) {
				if (pg.absgp > goal)
					break;
				best_end = pg.offset + (int64_t)pg.size;
			}
			at = pg.offset + (int64_t)pg.size;
		}
		forget();
		pos = best_end;
		return LW_OK;
	}
};

// ---------------------------------------------------------------------------------------------
// OggStreamReader
// ---------------------------------------------------------------------------------------------
struct lw_ogg_stream {
	lw_ogg_reader *rdr = nullptr;
	int device = 0;
	lw_ident *ident = nullptr;
	lw_comment *comment = nullptr;
	lw_setup *setup = nullptr;
	lw_decoder *dec = nullptr; // created at the first decode
	lw_pwr *pwr = nullptr;
	// look-ahead pipeline (lw_ogg_stream_read_dec_packets): a producer thread demultiplexes and entropy-decodes up to
	// pipe_k packets at a time into the staging ring while the caller's thread launches, collects and delivers earlier
	// batches; see "look-ahead pipeline" below
	lw_ring *ring = nullptr;
	size_t ring_cap = 0;
	int ring_fmt = -1;
	bool pipe_active = false;
	int pipe_fmt = -1, pipe_threads = 0;
	bool want_dev_entropy = false; // lw_ogg_stream_set_entropy_on_device: applied whenever the look-ahead pipeline (re)starts
	size_t pipe_k = 0;
	struct PipeSlot {
		std::vector<QueuedPacket> ahead;
		lw_pwr_state saved{};   // host half of the PreviousWindowRight before this batch was staged
		bool launched = false;
	};
	enum { TERM_NONE = 0, TERM_CHAIN, TERM_EOF, TERM_ERROR };
	struct Collected {          // a batch the demultiplexer thread has read, not yet entropy-decoded
		std::vector<QueuedPacket> ahead;
		int term = 0, term_rc = 0; // why no further batch follows this one (TERM_*)
	};
	std::deque<Collected> collected; // guarded by pmu
	Collected *in_stage = nullptr;   // the batch the staging thread is working on (its packets have left `collected`)
	std::vector<QueuedPacket> stage_failed;
	// what the failed batch carried besides its packets: the reason why no batch follows it (a one-shot container error must
	// not be lost when the packets are staged again), and the staging thread's HIP error text (lw_last_device_error is
	// thread-local: the caller's thread re-publishes it when it reports the failure)
	int stage_failed_term = TERM_NONE, stage_failed_term_rc = LW_OK;
	std::string stage_err_text;
	std::thread demuxer;
	std::deque<PipeSlot> staged;  // FIFO image of the ring's busy slots (guarded by pmu)
	int terminal = TERM_NONE, terminal_rc = LW_OK; // why the producer stopped reading (guarded by pmu)
	bool p_stop = false, p_done = false;
	std::thread producer;
	std::mutex pmu;
	std::condition_variable pcv;
	// packets (and one-shot container errors) that were read ahead and handed back by a roll-back: consumed before the
	// demultiplexer is asked again
	struct Requeued {
		QueuedPacket q;
		int rc = LW_OK; // != LW_OK: this container error occurs at this point of the sequence
	};
	std::deque<Requeued> requeue;
	uint32_t serial = 0;
	uint32_t link = 0; // logical streams entered so far minus one
	bool has_absgp = false;
	uint64_t cur_absgp = 0;
	// a packet read ahead by the look-ahead queue that belongs to the next call (chain boundary)
	bool has_pending = false;
	QueuedPacket pending;
	// an audio packet of the current logical stream handed back because the caller's buffer was too small
	bool has_retry = false;
	QueuedPacket retry;
	std::vector<uint8_t> scratch;
	// ---- read-ahead behind the packet-by-packet call (lw_ogg_stream_set_read_ahead): lw_ogg_stream_read_dec_packet hands out the
	// packets of a batch the look-ahead pipeline has decoded, one per call.  `served` = that batch (packets, statuses, sample
	// counts, the granule position as of each packet), served_next = the next one to hand out; the stream's running granule
	// position is that of the END of the batch, the caller sees the one of the last packet handed out (view_*).
	size_t ra_k = 0;
	int ra_threads = 0;
	struct Served {
		QueuedPacket q;
		int32_t status = 0;
		uint32_t n = 0;     // samples per channel handed out (after the truncation of the stream's last packet)
		uint32_t full = 0;  // samples per channel as decoded: the block [ch][full] (or [full][ch]) at element `off`
		size_t off = 0;     // of the batch's PCM in the ring slot's pinned buffer (served_pcm)
		bool has_absgp = false;
		uint64_t absgp = 0; // get_last_absgp() once this packet has been handed out
	};
	std::vector<Served> served;
	size_t served_next = 0;
	int served_fmt = -1;
	// the served batch's samples stay where the GPU's copy put them: its ring slot is held (collected, not released) until the last
	// packet has been handed out -- one copy per packet, straight into the caller's buffer
	const void *served_pcm = nullptr;
	bool slot_held = false;
	std::vector<uint32_t> ra_ns;
	std::vector<int32_t> ra_st;
	bool view_has_absgp = false;
	uint64_t view_absgp = 0;
	// What reproduces the PreviousWindowRight the caller's calls have led to: the last packet that decoded (a decoded packet
	// leaves its own raw right half and nothing of what came before, audio.rs:1125-1138) followed by every packet that has
	// failed since (a failure may or may not have emptied the state, :1083 / :1107-1111: decoding it again does the same);
	// empty = a fresh state.  Kept by every delivery path; used when packets of a served batch go back (unserve).
	std::vector<QueuedPacket> replay;

	bool serving() const { return served_next < served.size(); }

	void release_held()
	{
		if (!slot_held)
			return;
		(void)lw_ring_release(ring);
		std::unique_lock<std::mutex> g(pmu);
		slot_held = false;
		served_pcm = nullptr;
		pcv.notify_all(); // (the staging thread may have been waiting for the slot)
	}

	void note(QueuedPacket &&q, int status)
	{
		if (status == LW_OK)
			replay.clear();
		else if (status != LW_AUDIO_END_OF_PACKET && status != LW_AUDIO_BAD_FORMAT && status != LW_AUDIO_IS_HEADER &&
				status != LW_AUDIO_BUFFER_NOT_ADDRESSABLE)
			return; // (not the packet's doing: a device or argument error changes nothing)
		replay.push_back(std::move(q));
	}

	// the packets of the served batch that have not been handed out go back in front of everything else that was read ahead, and
	// the stream stands where the caller's calls have led it: granule position of the last packet handed out, its
	// PreviousWindowRight (re-made from `replay`: one synchronous decode, plus one per packet that failed since).  No-op otherwise.
	void unserve()
	{
		if (!serving()) {
			served.clear();
			served_next = 0;
			release_held();
			return;
		}
		release_held(); // (before the ring is drained: the pipeline's own bookkeeping never sees a slot of ours)
		rollback();
		for (size_t i = served.size(); i-- > served_next;) {
			Requeued r;
			r.q = std::move(served[i].q);
			requeue.push_front(std::move(r));
		}
		served.clear();
		served_next = 0;
		has_absgp = view_has_absgp;
		cur_absgp = view_absgp;
		std::vector<QueuedPacket> rp;
		rp.swap(replay);
		if (pwr)
			lw_pwr_reset(pwr);
		for (QueuedPacket &q : rp)
			(void)decode_discard(q); // (notes the packet again: `replay` is what it was)
	}

	// every entry point but the two reading calls: back to exactly what the caller has been handed
	void settle()
	{
		if (serving())
			unserve();
		else
			rollback();
	}

	void drop_context()
	{
		release_held();
		rollback();
		served.clear();
		served_next = 0;
		replay.clear();
		if (ring)
			lw_ring_destroy(ring);
		ring = nullptr;
		ring_cap = 0;
		if (pwr)
			lw_pwr_free(pwr);
		pwr = nullptr;
		if (dec)
			lw_decoder_destroy(dec);
		dec = nullptr;
		if (ident)
			lw_ident_free(ident);
		if (comment)
			lw_comment_free(comment);
		if (setup)
			lw_setup_free(setup);
		ident = nullptr;
		comment = nullptr;
		setup = nullptr;
	}

	int ensure_decoder()
	{
		if (!dec) {
			int e = 0;
			dec = lw_decoder_create(ident, setup, device, &e);
			if (!dec)
				return e ? e : LW_ERR_DEVICE;
		}
		if (!pwr) {
			pwr = lw_pwr_new(dec);
			if (!pwr)
				return LW_ERR_DEVICE;
		}
		return LW_OK;
	}

	void reset_pwr() // `self.pwr = PreviousWindowRight::new()`
	{
		if (pwr)
			lw_pwr_reset(pwr);
		replay.clear();
	}

	static void take(const lw_ogg_packet &k, QueuedPacket &q)
	{
		q.data.assign(k.data, k.data + k.len);
		q.serial = k.stream_serial;
		q.absgp_page = k.absgp_page;
		q.first_in_stream = k.first_in_stream;
		q.last_in_stream = k.last_in_stream;
		q.first_in_page = k.first_in_page;
		q.last_in_page = k.last_in_page;
	}

	// the three header packets of a logical stream whose ident packet is `first` (inside_ogg.rs:30-49, :123-131)
	int read_header_set(const QueuedPacket &first, bool skip_foreign, lw_ident **id, lw_comment **cm, lw_setup **st,
			uint32_t *ser)
	{
		int e = 0;
		*id = lw_read_header_ident(first.data.data(), first.data.size(), &e);
		if (!*id)
			return e;
		lw_ogg_packet k;
		auto next_of_stream = [&]() {
			for (;;) {
				const int rc = lw_ogg_read_packet_expected(rdr, &k);
				if (rc != LW_OK)
					return rc;
				if (!skip_foreign || k.stream_serial == first.serial)
					return (int)LW_OK;
			}
		};
		int rc = next_of_stream();
		if (rc == LW_OK) {
			*cm = lw_read_header_comment(k.data, k.len, &e);
			if (!*cm)
				rc = e;
		}
		if (rc == LW_OK)
			rc = next_of_stream();
		if (rc == LW_OK) {
			lw_ident_info info;
			lw_ident_get_info(*id, &info);
			*st = lw_read_header_setup(k.data, k.len, info.audio_channels, info.blocksize_0, info.blocksize_1, &e);
			if (!*st)
				rc = e;
			*ser = k.stream_serial;
		}
		if (rc != LW_OK) {
			if (*id)
				lw_ident_free(*id);
			if (*cm)
				lw_comment_free(*cm);
			*id = nullptr;
			*cm = nullptr;
		}
		return rc;
	}

	// decode one packet of the current logical stream into `out` (cap_elems elements); planar blocks are packed [ch][m].
	// A buffer that cannot hold a full block of the CURRENT logical stream (the stream may just have changed at a
	// chain boundary) hands the packet back: LW_ERR_CAPACITY, nothing consumed.
	int decode(QueuedPacket &q, int fmt, void *out, size_t cap_elems, size_t *m)
	{
		lw_ident_info info;
		lw_ident_get_info(ident, &info);
		const size_t cap = cap_elems / std::max<size_t>(info.audio_channels, 1);
		if (cap < ((size_t)1 << info.blocksize_1)) {
			retry = std::move(q);
			has_retry = true;
			return LW_ERR_CAPACITY;
		}
		if (int rc = ensure_decoder())
			return rc;
		const int rc = lw_read_audio_packet(dec, q.data.data(), q.data.size(), pwr, fmt, out, cap, m);
		note(QueuedPacket(q), rc);
		return rc;
	}

	int decode_discard(const QueuedPacket &q) // "read the first audio packet to prime the pwr and discard the packet"
	{
		lw_ident_info info;
		lw_ident_get_info(ident, &info);
		const size_t cap = (size_t)1 << info.blocksize_1;
		scratch.resize((size_t)info.audio_channels * cap * 2);
		size_t m = 0;
		if (int rc = ensure_decoder())
			return rc;
		const int rc = lw_read_audio_packet(dec, q.data.data(), q.data.size(), pwr, LW_FMT_I16_PLANAR, scratch.data(), cap, &m);
		note(QueuedPacket(q), rc);
		return rc;
	}

	// next packet of the physical stream: what a roll-back of the look-ahead pipeline handed back first, then the
	// demultiplexer.  LW_OK + q, LW_OGG_EOF, or an OggReadError code.
	int next_raw(QueuedPacket &q)
	{
		if (!requeue.empty()) {
			Requeued r = std::move(requeue.front());
			requeue.pop_front();
			if (r.rc != LW_OK)
				return r.rc;
			q = std::move(r.q);
			return LW_OK;
		}
		return rdr->next_owned(q);
	}

	// read_next_audio_packet, inside_ogg.rs:114-160.  LW_OK + q, LW_OGG_EOF, or an error.
	int next_audio(QueuedPacket &q)
	{
		if (has_retry) {
			q = std::move(retry);
			has_retry = false;
			return LW_OK;
		}
		for (;;) {
			if (has_pending && requeue.empty()) { // (what was handed back by a roll-back precedes the pending packet)
				q = std::move(pending);
				has_pending = false;
				if (q.serial == serial)
					return LW_OK;
				return chain(q);
			}
			const int rc = next_raw(q);
			if (rc != LW_OK)
				return rc;
			if (q.serial == serial)
				return LW_OK;
			if (q.first_in_stream)
				return chain(q);
			// every packet with a mismatching stream serial is ignored
		}
	}

	// chained file: q is the ident packet of the next logical stream (inside_ogg.rs:120-151)
	int chain(QueuedPacket &q)
	{
		lw_ident *id = nullptr;
		lw_comment *cm = nullptr;
		lw_setup *st = nullptr;
		uint32_t ser = 0;
		if (int rc = read_header_set(q, false, &id, &cm, &st, &ser))
			return rc;
		drop_context();
		ident = id;
		comment = cm;
		setup = st;
		serial = ser;
		link++;
		has_absgp = false;
		lw_ogg_packet k;
		int rc = lw_ogg_read_packet(rdr, &k);
		if (rc != LW_OK)
			return rc; // LW_OGG_EOF = Ok(None)
		take(k, q);
		if ((rc = decode_discard(q)) != LW_OK)
			return rc;
		has_absgp = true;
		cur_absgp = q.absgp_page;
		rc = lw_ogg_read_packet(rdr, &k);
		if (rc != LW_OK)
			return rc;
		take(k, q); // returned as is, whatever its serial (the reference does the same)
		return LW_OK;
	}

	// ---- look-ahead pipeline ----------------------------------------------------------------------------------
	// Three batches are in the pipeline at a time (ring of three slots): the producer thread demultiplexes and
	// entropy-decodes batch k+2 (lw_ring_stage) while the GPU works on batch k+1 (launched by the caller's thread as soon
	// as batch k has come back) and the caller's thread copies batch k out of pinned memory.  The producer only reads
	// packets of the current logical stream and stops in front of a chain boundary, at the end of the file and at a
	// container error, exactly where the sequential code stopped.  Everything lewton's API can do between two batched
	// calls (read_dec_packet, skip_samples_linear, seek_absgp_pg, into_inner) first ROLLS the pipeline BACK: the producer
	// is stopped, GPU work in flight is waited for and dropped, every packet that was read ahead but not delivered goes
	// back in front of the demultiplexer, and the PreviousWindowRight returns to its state after the last delivered packet
	// (its host half was saved before each batch was staged; of the undelivered batches at most ONE was launched, and a
	// launch writes the other of the state's two device buffers, so the device half is still there).
	// demultiplexer thread: batches of up to pipe_k packets of the current logical stream, stopping in front of a chain
	// boundary (this thread owns the demultiplexer, `requeue` and `pending` while the pipeline is active)
	void demux_main()
	{
		for (;;) {
			{
				std::unique_lock<std::mutex> g(pmu);
				pcv.wait(g, [&]() { return p_stop || collected.size() < 2; });
				if (p_stop)
					return;
			}
			Collected c;
			while (c.ahead.size() < pipe_k) {
				QueuedPacket q;
				if (has_pending && requeue.empty()) {
					if (pending.serial != serial) {
						c.term = TERM_CHAIN;
						break;
					}
					q = std::move(pending);
					has_pending = false;
				} else {
					const int rc = next_raw(q);
					if (rc == LW_OGG_EOF) {
						c.term = TERM_EOF;
						break;
					}
					if (rc != LW_OK) {
						c.term = TERM_ERROR;
						c.term_rc = rc;
						break;
					}
					if (q.serial != serial) {
						if (q.first_in_stream) {
							pending = std::move(q);
							has_pending = true;
							c.term = TERM_CHAIN;
							break;
						}
						continue; // every packet with a mismatching stream serial is ignored
					}
				}
				c.ahead.push_back(std::move(q));
			}
			const bool last = c.term != TERM_NONE;
			std::unique_lock<std::mutex> g(pmu);
			collected.push_back(std::move(c));
			pcv.notify_all();
			if (last)
				return;
		}
	}

	// staging thread: host entropy stage of the collected batches into the ring (lw_ring_stage runs the bit-serial decode
	// on pipe_threads host threads, this one included), in order
	void producer_main()
	{
		std::vector<lw_packet> pk;
		for (;;) {
			Collected c;
			{
				std::unique_lock<std::mutex> g(pmu);
				// (three ring slots: the batches staged or in flight, plus the one whose packets are being handed out one by one)
				pcv.wait(g, [&]() { return p_stop || (!collected.empty() && staged.size() + (slot_held ? 1 : 0) < 3); });
				if (p_stop)
					break;
				c = std::move(collected.front());
				// (the entry stays at the head of `collected`, emptied, until its packets sit in `staged`: a roll-back that
				// comes in between finds every packet in exactly one of the two queues or in `in_stage`)
				in_stage = &c;
			}
			PipeSlot ps;
			ps.ahead = std::move(c.ahead);
			int term = c.term, term_rc = c.term_rc;
			if (!ps.ahead.empty()) {
				lw_pwr_get_state(pwr, &ps.saved);
				pk.resize(ps.ahead.size());
				for (size_t i = 0; i < pk.size(); i++)
					pk[i] = lw_packet{ps.ahead[i].data.data(), ps.ahead[i].data.size(), pwr};
				const int rc = lw_ring_stage(ring, pk.data(), pk.size(), pipe_threads);
				if (rc != LW_OK) { // (cannot happen for a well-formed call: the packets go back, the error is reported)
					lw_pwr_set_state(pwr, &ps.saved);
					stage_failed = std::move(ps.ahead);
					stage_failed_term = c.term; // the batch's own end marker travels with its packets (rollback)
					stage_failed_term_rc = c.term_rc;
					if (const char *t = lw_last_device_error())
						stage_err_text = t;
					ps.ahead.clear();
					term = TERM_ERROR;
					term_rc = rc;
				}
			}
			std::unique_lock<std::mutex> g(pmu);
			in_stage = nullptr;
			collected.pop_front();
			if (!ps.ahead.empty())
				staged.push_back(std::move(ps));
			if (term != TERM_NONE) {
				terminal = term;
				terminal_rc = term_rc;
			}
			pcv.notify_all();
			if (term != TERM_NONE)
				break;
		}
		std::unique_lock<std::mutex> g(pmu);
		p_done = true;
		pcv.notify_all();
	}

	int pipeline_start(int fmt, size_t k, int n_threads)
	{
		if (int rc = ensure_decoder())
			return rc;
		if (!ring || ring_cap < k || ring_fmt != fmt) {
			if (ring)
				lw_ring_destroy(ring);
			int e = 0;
			ring = lw_ring_create(dec, 3, k, fmt, &e);
			if (!ring)
				return e ? e : LW_ERR_DEVICE;
			ring_cap = k;
			ring_fmt = fmt;
		}
		// entropy stage on the device when asked for and the current logical stream is eligible (else the host stage)
		// (LW_ERR_UNSUPPORTED = not eligible: expected, the host stage stays; anything else is a device failure)
		if (const int rc = lw_ring_set_entropy_on_device(ring, want_dev_entropy ? 1 : 0))
			if (rc != LW_ERR_UNSUPPORTED)
				return rc;
		pipe_fmt = fmt;
		pipe_k = k;
		pipe_threads = n_threads;
		terminal = TERM_NONE;
		terminal_rc = LW_OK;
		p_stop = p_done = false;
		pipe_active = true;
		demuxer = std::thread([this]() { demux_main(); });
		producer = std::thread([this]() { producer_main(); });
		return LW_OK;
	}

	// stop reading ahead and give back everything that was not delivered (see above); no-op when the pipeline is idle
	void rollback()
	{
		if (!pipe_active)
			return;
		{
			std::unique_lock<std::mutex> g(pmu);
			p_stop = true;
			pcv.notify_all();
		}
		if (producer.joinable())
			producer.join();
		if (demuxer.joinable())
			demuxer.join();
		(void)lw_ring_drain(ring);
		if (!staged.empty())
			lw_pwr_set_state(pwr, &staged.front().saved);
		// everything read ahead, in stream order: delivered-next batches first, then what was only demultiplexed; a container
		// error sits behind the packets it followed (the demultiplexer is only asked once `requeue` is empty)
		std::deque<Requeued> back;
		auto give_back = [&](std::vector<QueuedPacket> &v) {
			for (QueuedPacket &q : v) {
				Requeued r;
				r.q = std::move(q);
				back.push_back(std::move(r));
			}
		};
		for (PipeSlot &ps : staged)
			give_back(ps.ahead);
		give_back(stage_failed);
		stage_failed.clear();
		bool error_queued = false;
		if (stage_failed_term == TERM_ERROR) { // the container error that ended the batch whose staging failed: behind its packets
			Requeued r;
			r.rc = stage_failed_term_rc;
			back.push_back(std::move(r));
			error_queued = true;
		}
		stage_failed_term = TERM_NONE;
		for (Collected &c : collected) {
			give_back(c.ahead);
			if (c.term == TERM_ERROR) {
				Requeued r;
				r.rc = c.term_rc;
				back.push_back(std::move(r));
				error_queued = true;
			}
		}
		if (terminal == TERM_ERROR && !error_queued) {
			Requeued r;
			r.rc = terminal_rc;
			back.push_back(std::move(r));
		}
		for (Requeued &r : requeue)
			back.push_back(std::move(r));
		requeue.swap(back);
		staged.clear();
		collected.clear();
		terminal = TERM_NONE;
		pipe_active = false;
	}

	static void truncate(int fmt, int ch, void *out, size_t m, size_t target)
	{
		if (fmt == LW_FMT_I16_INTERLEAVED || target >= m)
			return; // interleaved: the first target * ch elements are already in place
		const size_t es = fmt == LW_FMT_F32_PLANAR ? 4 : 2;
		for (int c = 1; c < ch; c++)
			std::memmove((char *)out + (size_t)c * target * es, (char *)out + (size_t)c * m * es, target * es);
	}

	// granule bookkeeping of dec_packet_generic (inside_ogg.rs:219-230) for a packet that decoded to m samples
	size_t account(const QueuedPacket &q, int fmt, void *out, size_t m)
	{
		lw_ident_info info;
		lw_ident_get_info(ident, &info);
		if (has_absgp && q.last_in_stream) {
			const uint64_t target = q.absgp_page >= cur_absgp ? q.absgp_page - cur_absgp : 0; // saturating_sub
			if (target < m) {
				if (out)
					truncate(fmt, info.audio_channels, out, m, (size_t)target);
				m = (size_t)target;
			}
		}
		if (q.last_in_page) {
			has_absgp = true;
			cur_absgp = q.absgp_page;
		} else if (has_absgp) {
			cur_absgp += m;
		}
		return m;
	}
};

extern "C" {

uint32_t lw_ogg_crc32(const uint8_t *data, size_t len, uint32_t crc)
{
	return data ? crc_update(crc, data, len) : crc;
}

lw_ogg_reader *lw_ogg_reader_open_memory(const uint8_t *data, size_t len, int copy)
{
	if (!data && len)
		return nullptr;
	auto m = std::make_unique<MemSource>();
	if (copy) {
		m->own.assign(data, data + len);
		m->p = m->own.data();
	} else {
		m->p = data;
	}
	m->len = len;
	auto *r = new lw_ogg_reader();
	r->src = std::move(m);
	return r;
}

lw_ogg_reader *lw_ogg_reader_open_file(const char *path, int *err)
{
	FILE *f = path ? std::fopen(path, "rb") : nullptr;
	if (!f) {
		if (err)
			*err = path ? (int)LW_OGG_READ_ERROR : (int)LW_ERR_NULL_ARG;
		return nullptr;
	}
	auto s = std::make_unique<FileSource>();
	s->f = f;
	auto *r = new lw_ogg_reader();
	r->src = std::move(s);
	return r;
}

lw_ogg_reader *lw_ogg_reader_open_io(const lw_ogg_io *io)
{
	if (!io || !io->read || !io->seek)
		return nullptr;
	auto s = std::make_unique<IoSource>();
	s->io = *io;
	auto *r = new lw_ogg_reader();
	r->src = std::move(s);
	return r;
}

void lw_ogg_reader_close(lw_ogg_reader *r)
{
	delete r;
}

int lw_ogg_read_packet(lw_ogg_reader *r, lw_ogg_packet *out)
{
	if (!r || !out)
		return LW_ERR_NULL_ARG;
	return r->next(out);
}

int lw_ogg_read_packet_expected(lw_ogg_reader *r, lw_ogg_packet *out)
{
	const int rc = lw_ogg_read_packet(r, out);
	return rc == LW_OGG_EOF ? LW_OGG_READ_ERROR : rc;
}

void lw_ogg_delete_unread_packets(lw_ogg_reader *r)
{
	if (r)
		r->forget();
}

int lw_ogg_seek_absgp(lw_ogg_reader *r, int has_serial, uint32_t serial, uint64_t absgp)
{
	if (!r)
		return LW_ERR_NULL_ARG;
	return r->seek_absgp(has_serial != 0, serial, absgp);
}

lw_ogg_stream *lw_ogg_stream_open(lw_ogg_reader *r, int device, int *err)
{
	int dummy;
	if (!err)
		err = &dummy;
	*err = LW_OK;
	if (!r) {
		*err = LW_ERR_NULL_ARG;
		return nullptr;
	}
	auto *s = new lw_ogg_stream();
	s->rdr = r;
	s->device = device;
	lw_ogg_packet k;
	int rc = lw_ogg_read_packet_expected(r, &k);
	if (rc == LW_OK) {
		QueuedPacket first;
		lw_ogg_stream::take(k, first);
		rc = s->read_header_set(first, true, &s->ident, &s->comment, &s->setup, &s->serial);
	}
	if (rc != LW_OK) {
		*err = rc;
		lw_ogg_stream_close(s);
		return nullptr;
	}
	lw_ogg_delete_unread_packets(r);
	return s;
}

void lw_ogg_stream_close(lw_ogg_stream *s)
{
	if (!s)
		return;
	s->drop_context();
	lw_ogg_reader_close(s->rdr);
	delete s;
}

lw_ogg_reader *lw_ogg_stream_into_inner(lw_ogg_stream *s)
{
	if (!s)
		return nullptr;
	s->settle();
	lw_ogg_reader *r = s->rdr; // (packets read ahead are dropped with the stream object, like the reference's queue)
	s->rdr = nullptr;
	lw_ogg_stream_close(s);
	return r;
}

const lw_ident *lw_ogg_stream_ident(const lw_ogg_stream *s) { return s ? s->ident : nullptr; }
const lw_comment *lw_ogg_stream_comment(const lw_ogg_stream *s) { return s ? s->comment : nullptr; }
const lw_setup *lw_ogg_stream_setup(const lw_ogg_stream *s) { return s ? s->setup : nullptr; }
uint32_t lw_ogg_stream_serial(const lw_ogg_stream *s) { return s ? s->serial : 0; }
uint32_t lw_ogg_stream_link_index(const lw_ogg_stream *s) { return s ? s->link : 0; }

int lw_ogg_stream_last_absgp(const lw_ogg_stream *s, uint64_t *absgp)
{
	if (!s)
		return 0;
	// (packets of a served batch waiting to be handed out: the position of the last one the caller has, not the batch's end)
	const bool has = s->serving() ? s->view_has_absgp : s->has_absgp;
	if (has && absgp)
		*absgp = s->serving() ? s->view_absgp : s->cur_absgp;
	return has ? 1 : 0;
}

static int read_batch(lw_ogg_stream *s, int fmt, size_t max_packets, int n_threads, void *out, size_t cap_elems, uint32_t *n_samples,
		int32_t *status, size_t *n_packets, bool serve);

// memcpy on up to n_threads threads (0 = four), pieces of at least 2 MB
static void copy_out(char *dst, const char *src, size_t bytes, int n_threads)
{
	const size_t piece = (size_t)2 << 20;
	size_t t = (size_t)std::min(std::max(n_threads > 0 ? n_threads : 4, 1), 8);
	t = std::min(t, bytes / piece);
	if (t <= 1) {
		if (bytes)
			std::memcpy(dst, src, bytes);
		return;
	}
	const size_t per = ((bytes / t) + 63) & ~(size_t)63;
	std::vector<std::thread> th;
	for (size_t k = 1; k < t; k++) {
		const size_t a = k * per, b = k + 1 == t ? bytes : std::min(bytes, (k + 1) * per);
		if (a < b)
			th.emplace_back([=]() { std::memcpy(dst + a, src + a, b - a); });
	}
	std::memcpy(dst, src, std::min(per, bytes));
	for (std::thread &x : th)
		x.join();
}

int lw_ogg_stream_set_read_ahead(lw_ogg_stream *s, size_t max_packets, int n_threads)
{
	if (!s)
		return LW_ERR_NULL_ARG;
	if (max_packets > 65536)
		return LW_ERR_CAPACITY; // (a batch's samples are held in host memory: 65 536 stereo packets are 0.4 GB already)
	if (max_packets != s->ra_k || n_threads != s->ra_threads)
		s->settle(); // what was read ahead under the old setting goes back; the stream stands where the caller's calls have led it
	s->ra_k = max_packets;
	s->ra_threads = n_threads;
	return LW_OK;
}

// lw_ogg_stream_read_dec_packet with the read-ahead on: the next packet of the served batch, fetching a batch when there is none.
// Returns 1 when the call has to go the sequential way (in front of a chain boundary), else 0 with *rc = the call's result.
static int read_ahead_packet(lw_ogg_stream *s, int fmt, void *out, size_t cap_elems, size_t *n_samples, int *rc)
{
	for (;;) {
		if (s->serving() && s->served_fmt != fmt)
			s->unserve();
		lw_ident_info info;
		lw_ident_get_info(s->ident, &info);
		const size_t ch = std::max<size_t>(info.audio_channels, 1);
		if (s->has_retry)
			return 1; // (a packet the sequential path handed back for a larger buffer comes first, through that path)
		if (cap_elems / ch < ((size_t)1 << info.blocksize_1)) { // the packet-by-packet call's own rule; nothing consumed
			*rc = LW_ERR_CAPACITY;
			return 0;
		}
		if (s->serving()) {
			lw_ogg_stream::Served &e = s->served[s->served_next++];
			const size_t es = fmt == LW_FMT_F32_PLANAR ? 4 : 2;
			const int st = e.status;
			s->view_has_absgp = e.has_absgp;
			s->view_absgp = e.absgp;
			if (st == LW_OK) {
				const char *src = (const char *)s->served_pcm + e.off * es;
				if (fmt == LW_FMT_I16_INTERLEAVED || e.n == e.full) {
					std::memcpy(out, src, (size_t)e.n * ch * es);
				} else { // the stream's last packet, truncated (inside_ogg.rs:219-227): the first n samples of every channel
					for (size_t c = 0; c < ch; c++)
						std::memcpy((char *)out + c * e.n * es, src + c * e.full * es, (size_t)e.n * es);
				}
				*n_samples = e.n;
			}
			s->note(std::move(e.q), st);
			if (!s->serving()) {
				s->served.clear();
				s->served_next = 0;
				s->release_held();
			}
			*rc = st;
			return 0;
		}
		// a batch through the look-ahead pipeline; its samples stay in the ring slot (served_pcm)
		s->ra_ns.resize(s->ra_k);
		s->ra_st.resize(s->ra_k);
		size_t np = 0;
		s->served_fmt = fmt;
		const int r = read_batch(s, fmt, s->ra_k, s->ra_threads, nullptr, 0, s->ra_ns.data(), s->ra_st.data(), &np, true);
		if (r != LW_OK) {
			*rc = r;
			return 0;
		}
		if (np == 0)
			return 1; // in front of a chain boundary: the sequential call crosses it
	}
}

int lw_ogg_stream_read_dec_packet(lw_ogg_stream *s, int fmt, void *out, size_t cap_elems, size_t *n_samples)
{
	if (!s || !out || !n_samples)
		return LW_ERR_NULL_ARG;
	if (s->ra_k && fmt >= 0 && fmt <= 2) {
		int rc = LW_OK;
		if (!read_ahead_packet(s, fmt, out, cap_elems, n_samples, &rc))
			return rc;
	}
	s->settle();
	QueuedPacket q;
	if (int rc = s->next_audio(q))
		return rc;
	size_t m = 0;
	if (int rc = s->decode(q, fmt, out, cap_elems, &m))
		return rc;
	*n_samples = s->account(q, fmt, out, m);
	return LW_OK;
}

int lw_ogg_stream_set_entropy_on_device(lw_ogg_stream *s, int on)
{
	if (!s)
		return LW_ERR_NULL_ARG;
	if (s->want_dev_entropy != (on != 0)) {
		s->settle(); // the look-ahead restarts in the other mode from what the caller has been handed
		s->want_dev_entropy = on != 0;
	}
	return LW_OK;
}

int lw_ogg_stream_read_dec_packets(lw_ogg_stream *s, int fmt, size_t max_packets, int n_threads, void *out,
		size_t cap_elems, uint32_t *n_samples, int32_t *status, size_t *n_packets)
{
	if (!s || !out || !n_samples || !status || !n_packets || max_packets == 0 || fmt < 0 || fmt > 2)
		return LW_ERR_NULL_ARG;
	if (s->serving())
		s->unserve(); // (the two reading calls mixed with the read-ahead on: this one continues behind the last packet handed out)
	return read_batch(s, fmt, max_packets, n_threads, out, cap_elems, n_samples, status, n_packets, false);
}

// one batch of the look-ahead pipeline, delivered into `out`; serve: its packets are kept in s->served for
// lw_ogg_stream_read_dec_packet to hand out one by one (else they count as handed out by this call)
static int read_batch(lw_ogg_stream *s, int fmt, size_t max_packets, int n_threads, void *out, size_t cap_elems, uint32_t *n_samples,
		int32_t *status, size_t *n_packets, bool serve)
{
	*n_packets = 0;
	if (s->pipe_active && (s->pipe_fmt != fmt || s->pipe_k != max_packets || s->pipe_threads != n_threads))
		s->rollback(); // other batch geometry: what was read ahead is staged again under the new one
	if (!s->pipe_active)
		if (int rc = s->pipeline_start(fmt, max_packets, n_threads))
			return rc;
	// the oldest batch in the pipeline, or the reason why there is none
	lw_ogg_stream::PipeSlot *ps = nullptr;
	{
		std::unique_lock<std::mutex> g(s->pmu);
		s->pcv.wait(g, [&]() { return !s->staged.empty() || s->p_done; });
		if (!s->staged.empty())
			ps = &s->staged.front(); // (deque: stays valid while the producer appends at the back)
	}
	if (!ps) {
		const int term = s->terminal, rc = s->terminal_rc;
		s->terminal = lw_ogg_stream::TERM_NONE; // reported by this call
		if (!s->stage_err_text.empty()) { // a failure on the staging thread: its HIP error text, on the thread that reports it
			lw_set_device_error(s->stage_err_text);
			s->stage_err_text.clear();
		}
		s->rollback(); // both threads have stopped or stop now; nothing is staged; what the demultiplexer still holds goes back
		if (term == lw_ogg_stream::TERM_ERROR)
			return rc;
		if (term == lw_ogg_stream::TERM_EOF)
			return LW_OGG_EOF;
		return LW_OK; // in front of a chain boundary: *n_packets == 0, lw_ogg_stream_read_dec_packet crosses it
	}
	if (!ps->launched) {
		if (int rc = lw_ring_launch(s->ring))
			return rc;
		ps->launched = true;
	}
	const lw_packet_result *res = nullptr;
	const void *pcm = nullptr;
	size_t n = 0, total = 0;
	if (int rc = lw_ring_collect(s->ring, &res, &n, &pcm, &total)) {
		// LW_ERR_DEVICE: the batch's samples are void and the batches behind it were planned on window states it never produced.
		// Nothing of it is handed out: the stream goes back to what the caller has been given -- packets re-queued in order,
		// PreviousWindowRight restored from the snapshot taken before this batch, ring drained -- and the next call stages and
		// decodes the same packets again.  Never wrong samples under LW_OK.
		if (rc == LW_ERR_DEVICE)
			s->rollback();
		return rc;
	}
	if (!serve && total > cap_elems)
		return LW_ERR_CAPACITY; // nothing consumed: the batch stays at the head of the pipeline for a larger buffer
	// the next batch goes to the GPU now, while this one is copied out (at most one launched batch is ever undelivered)
	{
		lw_ogg_stream::PipeSlot *next = nullptr;
		{
			std::unique_lock<std::mutex> g(s->pmu);
			if (s->staged.size() >= 2 && !s->staged[1].launched)
				next = &s->staged[1];
		}
		// (a failure here is not lost: the slot stays un-launched and the next call launches it again and reports the error)
		if (next && lw_ring_launch(s->ring) == LW_OK)
			next->launched = true;
	}
	// per packet: truncation of the stream's last packet and granule bookkeeping, blocks packed back to back
	lw_ident_info info;
	lw_ident_get_info(s->ident, &info);
	const size_t es = fmt == LW_FMT_F32_PLANAR ? 4 : 2;
	size_t w = 0; // write cursor in elements
	if (serve) {
		s->served.clear();
		s->served.reserve(n);
		s->served_next = 0;
		s->view_has_absgp = s->has_absgp; // (as of the last packet handed out: nothing of this batch yet)
		s->view_absgp = s->cur_absgp;
	}
	// the batched call: the whole batch out of pinned memory with one copy, on several threads when it is large (a single
	// thread's memcpy, ~8 GB/s, was the bound of one stream: 4 KB per stereo packet); the blocks are already back to back, only a
	// truncated packet (the last of its stream) makes what follows move up
	if (!serve)
		copy_out((char *)out, (const char *)pcm, total * es, n_threads);
	for (size_t i = 0; i < n; i++) {
		status[i] = res[i].status;
		size_t m = 0;
		const size_t full = res[i].status == LW_OK ? (size_t)res[i].n_samples * info.audio_channels : 0;
		char *dst = serve ? nullptr : (char *)out + w * es;
		if (!serve && full && w != res[i].out_offset)
			std::memmove(dst, (const char *)out + res[i].out_offset * es, full * es);
		if (res[i].status == LW_OK)
			m = s->account(ps->ahead[i], fmt, dst, res[i].n_samples);
		n_samples[i] = (uint32_t)m;
		if (serve) {
			lw_ogg_stream::Served e;
			e.q = std::move(ps->ahead[i]);
			e.status = res[i].status;
			e.n = (uint32_t)m;
			e.full = res[i].status == LW_OK ? res[i].n_samples : 0;
			e.off = res[i].out_offset;
			e.has_absgp = s->has_absgp;
			e.absgp = s->cur_absgp;
			s->served.push_back(std::move(e));
		}
		w += m * info.audio_channels;
	}
	if (!serve) { // handed out by this call: what the PreviousWindowRight now derives from (see `replay`)
		size_t from = 0;
		for (size_t i = n; i-- > 0;)
			if (res[i].status == LW_OK) {
				from = i;
				break;
			}
		for (size_t i = from; i < n; i++)
			s->note(std::move(ps->ahead[i]), res[i].status);
	}
	*n_packets = n;
	if (serve && n) { // the slot stays ours until its last packet has been handed out (release_held)
		std::unique_lock<std::mutex> g(s->pmu);
		s->served_pcm = pcm;
		s->slot_held = true;
		s->staged.pop_front();
		s->pcv.notify_all();
		return LW_OK;
	}
	(void)lw_ring_release(s->ring);
	{
		std::unique_lock<std::mutex> g(s->pmu);
		s->staged.pop_front();
		s->pcv.notify_all();
	}
	return LW_OK;
}

int lw_ogg_stream_skip_samples_linear(lw_ogg_stream *s, size_t to_skip, int fmt, void *out, size_t cap_elems,
		size_t *n_samples, size_t *left, int *got_packet)
{
	if (!s || !out || !n_samples || !left || !got_packet)
		return LW_ERR_NULL_ARG;
	*got_packet = 0;
	*n_samples = 0;
	s->settle();
	bool have_last = false;
	QueuedPacket last, next;
	for (;;) {
		const int rc = s->next_audio(next);
		if (rc == LW_OGG_EOF) {
			*left = to_skip;
			return LW_OK;
		}
		if (rc != LW_OK)
			return rc;
		size_t cnt = 0;
		if (int e = lw_get_decoded_sample_count(s->ident, s->setup, next.data.data(), next.data.size(), &cnt))
			return e;
		if (s->has_absgp && next.last_in_stream) {
			have_last = false;
			const uint64_t target = next.absgp_page >= s->cur_absgp ? next.absgp_page - s->cur_absgp : 0;
			cnt = (size_t)std::min<uint64_t>(cnt, target);
		}
		if (to_skip < cnt) {
			if (have_last) {
				s->reset_pwr();
				if (int e = s->decode_discard(last))
					return e;
			}
			size_t m = 0;
			*left = to_skip; // on LW_ERR_CAPACITY: call again with this many samples and a larger buffer
			if (int e = s->decode(next, fmt, out, cap_elems, &m))
				return e;
			*n_samples = s->account(next, fmt, out, m);
			*got_packet = 1;
			return LW_OK;
		}
		to_skip -= cnt;
		if (s->has_absgp)
			s->cur_absgp += cnt;
		last = std::move(next);
		have_last = true;
	}
}

int lw_ogg_stream_seek_absgp_pg(lw_ogg_stream *s, uint64_t absgp)
{
	if (!s)
		return LW_ERR_NULL_ARG;
	s->settle();
	s->requeue.clear(); // everything read ahead is void after a seek
	s->has_pending = false;
	s->has_retry = false;
	if (int rc = lw_ogg_seek_absgp(s->rdr, 0, 0, absgp))
		return rc;
	s->has_absgp = false;
	s->reset_pwr();
	return LW_OK;
}

} // extern "C"
