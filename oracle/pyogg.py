"""CPU restatement of the Ogg layer either side of the hot path.

TEST INFRASTRUCTURE ONLY (like the rest of oracle/): nothing under lewton_amd/ imports this module.

Three parts:

* Ogg page / packet demultiplexing.  lewton delegates this to the external crate `ogg 0.8.0`
  (Cargo.lock; not vendored under /root/reference), so the restatement follows the published
  format, RFC 3533 section 6 (page header, lacing) and its CRC (polynomial 0x04c11db7, initial value 0,
  no reflection, no final xor, CRC field zeroed while summing), and the packet attributes lewton's call
  sites consume: `stream_serial()`, `first_in_stream()`, `last_in_stream()`, `last_in_page()`,
  `absgp_page()` (inside_ogg.rs:32-49, 116-151, 219-227).
* `read_header_comment` (header.rs:309-355), which the C oracle does not have.
* `OggStreamReader` (inside_ogg.rs:66-314): header bootstrap, chained streams, truncation of the last
  packet to the final granule position, `skip_samples_linear`, `seek_absgp_pg`, over the oracle's
  packet decoder (oracle/pyoracle.py).

Pure-Python loops: meant for the small streams of the tests.
"""
import struct

from . import pyoracle as po

CAPTURE = b"OggS"


def _crc_table():
    t = []
    for i in range(256):
        r = i << 24
        for _ in range(8):
            r = ((r << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if r & 0x80000000 else (r << 1) & 0xFFFFFFFF
        t.append(r)
    return t


_CRC = _crc_table()


def crc32_ogg(data):
    r = 0
    for b in data:
        r = ((r << 8) & 0xFFFFFFFF) ^ _CRC[((r >> 24) ^ b) & 0xFF]
    return r


class OggError(Exception):
    """kind in {"NoCapturePatternFound", "InvalidStreamStructVer", "HashMismatch", "ReadError", "InvalidData"}."""

    def __init__(self, kind, detail=""):
        super().__init__("%s %s" % (kind, detail))
        self.kind = kind


class Page:
    __slots__ = ("offset", "size", "continued", "bos", "eos", "absgp", "serial", "seq", "lacing", "body")


def parse_page(data, off):
    """Page at byte offset `off` (must start with the capture pattern); returns Page or None at a clean end."""
    if off == len(data):
        return None
    if len(data) - off < 27:
        raise OggError("ReadError", "truncated page header")
    if data[off:off + 4] != CAPTURE:
        raise OggError("NoCapturePatternFound")
    ver, flags, absgp, serial, seq, crc, nseg = struct.unpack_from("<BBQIIIB", data, off + 4)
    if ver != 0:
        raise OggError("InvalidStreamStructVer", str(ver))
    if len(data) - off < 27 + nseg:
        raise OggError("ReadError", "truncated lacing table")
    lacing = list(data[off + 27:off + 27 + nseg])
    size = 27 + nseg + sum(lacing)
    if len(data) - off < size:
        raise OggError("ReadError", "truncated page body")
    raw = bytearray(data[off:off + size])
    raw[22:26] = b"\0\0\0\0"
    if crc32_ogg(raw) != crc:
        raise OggError("HashMismatch")
    p = Page()
    p.offset, p.size = off, size
    p.continued, p.bos, p.eos = bool(flags & 1), bool(flags & 2), bool(flags & 4)
    p.absgp, p.serial, p.seq, p.lacing = absgp, serial, seq, lacing
    p.body = bytes(data[off + 27 + nseg:off + size])
    return p


class OggPacket:
    __slots__ = ("data", "serial", "first_in_stream", "last_in_stream", "first_in_page", "last_in_page", "absgp_page")

    def stream_serial(self):
        return self.serial


class PacketReader:
    """Packets of all logical streams of a physical stream in the order they are completed."""

    def __init__(self, data):
        self.data = bytes(data)
        self.pos = 0
        self.partial = {}   # serial -> bytearray of a packet continued on the next page
        self.queue = []

    def _read_page(self):
        p = parse_page(self.data, self.pos)
        if p is None:
            return False
        self.pos = p.offset + p.size
        buf = self.partial.pop(p.serial, None)
        if buf is not None and not p.continued:
            buf = None  # the unfinished packet is dropped when the next page does not continue it
        skipping = buf is None and p.continued  # continuation of a packet whose start was not seen
        if buf is None:
            buf = bytearray()
        done = []
        o = 0
        for lv in p.lacing:
            buf += p.body[o:o + lv]
            o += lv
            if lv < 255:
                if skipping:
                    skipping = False
                else:
                    done.append(bytes(buf))
                buf = bytearray()
        if p.lacing and p.lacing[-1] == 255:
            if not skipping:
                self.partial[p.serial] = buf
        for i, d in enumerate(done):
            k = OggPacket()
            k.data, k.serial = d, p.serial
            k.first_in_page, k.last_in_page = i == 0, i == len(done) - 1
            k.first_in_stream = p.bos and i == 0
            k.last_in_stream = p.eos and i == len(done) - 1
            k.absgp_page = p.absgp
            self.queue.append(k)
        return True

    def read_packet(self):
        while not self.queue:
            if not self._read_page():
                return None
        return self.queue.pop(0)

    def read_packet_expected(self):
        k = self.read_packet()
        if k is None:
            raise OggError("ReadError", "UnexpectedEof")
        return k

    def delete_unread_packets(self):
        self.queue = []
        self.partial = {}

    def seek_bytes_to_page(self, off):
        """Continue reading at the first valid page at or after byte offset `off`."""
        self.delete_unread_packets()
        while True:
            i = self.data.find(CAPTURE, off)
            if i < 0:
                self.pos = len(self.data)
                return
            try:
                if parse_page(self.data, i) is not None:
                    self.pos = i
                    return
            except OggError:
                pass
            off = i + 1

    def seek_absgp(self, serial, goal):
        """Per-page seek: reading resumes at the page after the LAST page of the stream (any stream if serial is
        None) whose granule position is <= goal and on which a packet ends; at the first page if there is none."""
        off, best_end = 0, 0
        while True:
            i = self.data.find(CAPTURE, off)
            if i < 0:
                break
            try:
                p = parse_page(self.data, i)
            except OggError:
                off = i + 1
                continue
            if p is None:
                break
            if (serial is None or p.serial == serial) and p.absgp != 0xFFFFFFFFFFFFFFFF and p.absgp <= goal \
                    and any(lv < 255 for lv in p.lacing):
                best_end = p.offset + p.size
            off = p.offset + p.size
        self.delete_unread_packets()
        self.pos = best_end
        return True


class VorbisError(Exception):
    """kind "BadAudio" / "BadHeader" / "OggError" with the inner code or kind (lib.rs:120-157)."""

    def __init__(self, kind, inner):
        super().__init__("%s(%s)" % (kind, inner))
        self.kind, self.inner = kind, inner


def _hdr(fn, *a):
    try:
        return fn(*a)
    except po.OracleError as e:
        raise VorbisError("BadHeader", e.code)


def read_header_comment(packet):
    """header.rs:309-355 (+ read_header_begin_body :131-153): (vendor, [(key, value)]) or OracleError(HeaderReadError code).
    Comments that are not UTF-8 or have no '=' are skipped, the key is everything before the FIRST '='; the framing byte
    must be exactly 1.  Every short read is EndOfPacket (From<io::Error>, :80-87)."""
    pos = 0

    def take(n):
        nonlocal pos
        if len(packet) - pos < n:
            raise po.OracleError(po.HDR_END_OF_PACKET)
        b = bytes(packet[pos:pos + n])
        pos += n
        return b

    hd_id = take(1)[0]
    if hd_id & 1 == 0:
        raise po.OracleError(po.HDR_IS_AUDIO)
    for want in b"vorbis":          # `&&` chain: reading stops at the first mismatch
        if take(1)[0] != want:
            raise po.OracleError(po.HDR_NOT_VORBIS)
    if hd_id != 3:
        raise po.OracleError(po.HDR_BAD_TYPE)
    try:
        vendor = take(struct.unpack("<I", take(4))[0]).decode("utf-8")
    except UnicodeDecodeError:
        raise po.OracleError(po.HDR_UTF8)
    comments = []
    for _ in range(struct.unpack("<I", take(4))[0]):
        raw = take(struct.unpack("<I", take(4))[0])
        try:
            text = raw.decode("utf-8")
        except UnicodeDecodeError:
            continue
        if "=" not in text:
            continue
        key, val = text.split("=", 1)
        comments.append((key, val))
    if take(1)[0] != 1:
        raise po.OracleError(po.HDR_BAD_FORMAT)
    return vendor, comments


def read_headers(rdr):
    """inside_ogg.rs:30-49"""
    try:
        pck = rdr.read_packet_expected()
        ident = _hdr(po.Ident, pck.data)
        serial = pck.serial
        pck = rdr.read_packet_expected()
        while pck.serial != serial:
            pck = rdr.read_packet_expected()
        comment_packet = pck.data
        _hdr(read_header_comment, comment_packet)
        pck = rdr.read_packet_expected()
        while pck.serial != serial:
            pck = rdr.read_packet_expected()
        setup = _hdr(po.Setup, pck.data, ident)
    except OggError as e:
        raise VorbisError("OggError", e.kind)
    rdr.delete_unread_packets()
    return (ident, comment_packet, setup), pck.serial


class OggStreamReader:
    def __init__(self, data, fmt="i16"):
        self.rdr = PacketReader(data)
        (self.ident_hdr, self.comment_packet, self.setup_hdr), self.stream_serial = read_headers(self.rdr)
        self.pwr = po.Pwr()
        self.cur_absgp = None
        self.fmt = fmt

    def _decode(self, data, pwr=None):
        try:
            return po.read_audio_packet(self.ident_hdr, self.setup_hdr, data, pwr or self.pwr, self.fmt)
        except po.OracleError as e:
            raise VorbisError("BadAudio", e.code)

    def _read_packet(self, expected=False):
        try:
            return self.rdr.read_packet_expected() if expected else self.rdr.read_packet()
        except OggError as e:
            raise VorbisError("OggError", e.kind)

    def read_next_audio_packet(self):  # inside_ogg.rs:114-160
        while True:
            pck = self._read_packet()
            if pck is None:
                return None
            if pck.serial == self.stream_serial:
                return pck
            if pck.first_in_stream:
                ident = _hdr(po.Ident, pck.data)
                pck = self._read_packet(True)
                self.comment_packet = pck.data
                _hdr(read_header_comment, pck.data)
                pck = self._read_packet(True)
                setup = _hdr(po.Setup, pck.data, ident)
                self.pwr = po.Pwr()
                self.ident_hdr, self.setup_hdr = ident, setup
                self.stream_serial = pck.serial
                self.cur_absgp = None
                pck = self._read_packet()
                if pck is None:
                    return None
                self._decode(pck.data)
                self.cur_absgp = pck.absgp_page
                return self._read_packet()

    @staticmethod
    def _num_samples(dec, fmt, ch):
        return len(dec) // ch if fmt == "i16_itl" else dec.shape[1]

    def _truncate(self, dec, n):
        ch = self.ident_hdr.audio_channels
        if self.fmt == "i16_itl":
            return dec[: n * ch]
        return dec[:, :n]

    def dec_packet(self, pck):  # inside_ogg.rs:208-231
        dec = self._decode(pck.data)
        ch = self.ident_hdr.audio_channels
        if self.cur_absgp is not None and pck.last_in_stream:
            target = max(0, pck.absgp_page - self.cur_absgp)
            if target < self._num_samples(dec, self.fmt, ch):
                dec = self._truncate(dec, target)
        if pck.last_in_page:
            self.cur_absgp = pck.absgp_page
        elif self.cur_absgp is not None:
            self.cur_absgp += self._num_samples(dec, self.fmt, ch)
        return dec

    def read_dec_packet(self):  # inside_ogg.rs:167-206
        pck = self.read_next_audio_packet()
        if pck is None:
            return None
        return self.dec_packet(pck)

    def skip_samples_linear(self, to_skip):  # inside_ogg.rs:244-283
        last = None
        while True:
            nxt = self.read_next_audio_packet()
            if nxt is None:
                return None, to_skip
            try:
                cnt = po.get_decoded_sample_count(self.ident_hdr, self.setup_hdr, nxt.data)
            except po.OracleError as e:
                raise VorbisError("BadAudio", e.code)
            if self.cur_absgp is not None and nxt.last_in_stream:
                last = None
                cnt = min(cnt, max(0, nxt.absgp_page - self.cur_absgp))
            if to_skip < cnt:
                if last is not None:
                    self.pwr = po.Pwr()
                    self._decode(last.data)
                return self.dec_packet(nxt), to_skip
            to_skip -= cnt
            if self.cur_absgp is not None:
                self.cur_absgp += cnt
            last = nxt

    def get_last_absgp(self):
        return self.cur_absgp

    def seek_absgp_pg(self, absgp):  # inside_ogg.rs:307-313
        self.rdr.seek_absgp(None, absgp)
        self.cur_absgp = None
        self.pwr = po.Pwr()
