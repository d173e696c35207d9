"""Synthetic Vorbis I stream generator (test / bench input tooling).

The reference's own media are downloaded at test time (dev/cmp/src/lib.rs:238-674) and are not
available offline, so every BASELINE.json config is driven by streams built here: real ident /
comment / setup header packets (bit layout as parsed by src/header.rs:221-259, :309-355,
:673-1154) and audio packets whose bits are emitted in exactly the order the decoder consumes them
(src/audio.rs:919-986).  Symbols are drawn at random; the decoded signal is
noise-like with a realistic spectral envelope.

This module contains no decoding logic and is independent of both the oracle and the product.
"""
from __future__ import annotations

import heapq
import struct
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------------------------
# bit writer: LSb-first, little-endian fields (Vorbis I spec section 2.1; src/bitpacking.rs:93-161)
# --------------------------------------------------------------------------------------------


class BitWriter:
    __slots__ = ("acc", "nbits")

    def __init__(self):
        self.acc = 0
        self.nbits = 0

    def write(self, value: int, width: int):
        if width:
            self.acc |= (int(value) & ((1 << width) - 1)) << self.nbits
            self.nbits += width

    def bytes(self) -> bytes:
        return self.acc.to_bytes((self.nbits + 7) // 8, "little")


def ilog(v: int) -> int:
    return int(v).bit_length()


def float32_pack(x: float) -> int:
    """Inverse of src/bitpacking.rs:304-314 for values with a <=21-bit mantissa."""
    if x == 0:
        return 788 << 21
    sgn = 0x80000000 if x < 0 else 0
    m = abs(float(x))
    e = 0
    while m != int(m):
        m *= 2.0
        e -= 1
    mant = int(m)
    while mant >= (1 << 21):
        assert mant % 2 == 0, "value not representable"
        mant //= 2
        e += 1
    exp = e + 788
    assert 0 <= exp < 1024
    return sgn | (exp << 21) | mant


# --------------------------------------------------------------------------------------------
# Huffman: lengths from probabilities; codewords by the Vorbis rule (spec 3.2.1: each entry, in
# entry order, takes the leftmost free leaf at its depth; src/huffman_tree.rs:66-123)
# --------------------------------------------------------------------------------------------


def huffman_lengths(probs: Sequence[float], max_len: int = 24) -> List[int]:
    probs = np.asarray(probs, np.float64)
    n = len(probs)
    if n == 1:
        return [1]
    floor_p = 0.0
    while True:
        p = np.maximum(probs, floor_p)
        heap = [(float(p[i]), i, (i,)) for i in range(n)]
        heapq.heapify(heap)
        lengths = [0] * n
        cnt = n
        while len(heap) > 1:
            a = heapq.heappop(heap)
            b = heapq.heappop(heap)
            for i in a[2] + b[2]:
                lengths[i] += 1
            heapq.heappush(heap, (a[0] + b[0], cnt, a[2] + b[2]))
            cnt += 1
        if max(lengths) <= max_len:
            return lengths
        floor_p = max(floor_p * 4.0, probs.sum() * 2.0 ** (-max_len + 2))


def assign_codewords(lengths: Sequence[int]) -> List[Optional[Tuple[int, int]]]:
    """Returns per entry (bits_lsb_first, length) or None for unused entries.

    bits_lsb_first has the root decision in bit 0 (the order the decoder reads them)."""
    # explicit binary tree with "full" marks; insertion = depth-first, left (0) before right (1)
    child = [[-1, -1]]
    leaf = [False]
    full = [False]

    def insert(node, depth, path, plen):
        if leaf[node]:
            return None
        if depth == 0:
            if child[node][0] >= 0 or child[node][1] >= 0:
                return None
            leaf[node] = True
            full[node] = True
            return path
        if full[node]:
            return None
        for side in (0, 1):
            c = child[node][side]
            if c < 0:
                c = len(child)
                child.append([-1, -1])
                leaf.append(False)
                full.append(False)
                child[node][side] = c
            if not full[c]:
                r = insert(c, depth - 1, path | (side << plen), plen + 1)
                if r is not None:
                    l, rr = child[node]
                    full[node] = l >= 0 and rr >= 0 and full[l] and full[rr]
                    return r
        return None

    out: List[Optional[Tuple[int, int]]] = []
    used = [i for i, l in enumerate(lengths) if l > 0]
    for i, l in enumerate(lengths):
        if l == 0:
            out.append(None)
            continue
        r = insert(0, l, 0, 0)
        if r is None:
            raise ValueError("overspecified Huffman lengths")
        out.append((r, l))
    if len(used) == 1:
        # single-entry book: one bit, either value decodes it (src/huffman_tree.rs:202-217)
        assert lengths[used[0]] == 1
    return out


# --------------------------------------------------------------------------------------------
# setup model (mirrors the fields of src/header.rs:363-481)
# --------------------------------------------------------------------------------------------


@dataclass
class Codebook:
    dims: int
    lengths: List[int]
    lookup_type: int = 0
    minimum: float = 0.0
    delta: float = 1.0
    value_bits: int = 1
    sequence_p: bool = False
    multiplicands: List[int] = field(default_factory=list)
    ordered: bool = False
    sparse: Optional[bool] = None
    _cw: Optional[list] = None

    @property
    def entries(self):
        return len(self.lengths)

    def codewords(self):
        if self._cw is None:
            self._cw = assign_codewords(self.lengths)
        return self._cw

    def used_entries(self):
        return [i for i, l in enumerate(self.lengths) if l > 0]


@dataclass
class Floor1:
    partition_class: List[int]
    class_dim: List[int]
    class_sub: List[int]
    class_master: List[int]
    sub_books: List[List[int]]  # -1 = no book
    multiplier: int
    rangebits: int
    x_rest: List[int]  # x_list[2:]

    @property
    def x_list(self):
        return [0, 1 << self.rangebits] + list(self.x_rest)


@dataclass
class Floor0:
    order: int
    rate: int
    bark_map_size: int
    amplitude_bits: int
    amplitude_offset: int
    book_list: List[int]
    amp_max: int = 0   # generator only: largest amplitude drawn (0 = the full range of amplitude_bits)


@dataclass
class Residue:
    type: int
    begin: int
    end: int
    partition_size: int
    classifications: int
    classbook: int
    books: List[List[int]]  # per class: 8 entries, -1 = none (pass 7 can never be coded, header.rs:450)


@dataclass
class Mapping:
    coupling: List[Tuple[int, int]]
    mux: List[int]
    submap_floor: List[int]
    submap_residue: List[int]


@dataclass
class Mode:
    blockflag: int
    mapping: int


@dataclass
class StreamSetup:
    channels: int
    sample_rate: int
    bs0: int
    bs1: int
    codebooks: List[Codebook]
    floors: list
    residues: List[Residue]
    mappings: List[Mapping]
    modes: List[Mode]

    # ---- header packets ----
    def ident_packet(self) -> bytes:
        return (b"\x01vorbis" + struct.pack("<IBIiii", 0, self.channels, self.sample_rate, 0, 112000, 0)
                + bytes([(self.bs1 << 4) | self.bs0, 1]))

    def comment_packet(self, vendor=b"lewton_amd streamgen", comments=(b"TITLE=synthetic",)) -> bytes:
        out = b"\x03vorbis" + struct.pack("<I", len(vendor)) + vendor + struct.pack("<I", len(comments))
        for c in comments:
            out += struct.pack("<I", len(c)) + c
        return out + b"\x01"

    def setup_packet(self) -> bytes:
        w = BitWriter()
        for b in b"\x05vorbis":
            w.write(b, 8)
        w.write(len(self.codebooks) - 1, 8)
        for cb in self.codebooks:
            _write_codebook(w, cb)
        w.write(0, 6)  # one time-domain transform placeholder
        w.write(0, 16)
        w.write(len(self.floors) - 1, 6)
        for fl in self.floors:
            _write_floor(w, fl)
        w.write(len(self.residues) - 1, 6)
        for rs in self.residues:
            _write_residue(w, rs)
        w.write(len(self.mappings) - 1, 6)
        for m in self.mappings:
            _write_mapping(w, m, self.channels)
        w.write(len(self.modes) - 1, 6)
        for md in self.modes:
            w.write(md.blockflag, 1)
            w.write(0, 16)
            w.write(0, 16)
            w.write(md.mapping, 8)
        w.write(1, 1)
        return w.bytes()

    def headers(self):
        return self.ident_packet(), self.comment_packet(), self.setup_packet()


def _write_codebook(w: BitWriter, cb: Codebook):
    w.write(0x564342, 24)
    w.write(cb.dims, 16)
    w.write(cb.entries, 24)
    if cb.ordered:
        # lengths must be non-decreasing
        w.write(1, 1)
        ls = cb.lengths
        assert all(ls[i] <= ls[i + 1] for i in range(len(ls) - 1)) and ls[0] >= 1
        cur, i = ls[0], 0
        w.write(cur - 1, 5)
        while i < len(ls):
            num = 0
            while i + num < len(ls) and ls[i + num] == cur:
                num += 1
            w.write(num, ilog(len(ls) - i))
            i += num
            cur += 1
    else:
        w.write(0, 1)
        sparse = cb.sparse if cb.sparse is not None else any(l == 0 for l in cb.lengths)
        w.write(1 if sparse else 0, 1)
        for l in cb.lengths:
            if sparse:
                if l:
                    w.write(1, 1)
                    w.write(l - 1, 5)
                else:
                    w.write(0, 1)
            else:
                assert l >= 1
                w.write(l - 1, 5)
    w.write(cb.lookup_type, 4)
    if cb.lookup_type:
        w.write(float32_pack(cb.minimum), 32)
        w.write(float32_pack(cb.delta), 32)
        w.write(cb.value_bits - 1, 4)
        w.write(1 if cb.sequence_p else 0, 1)
        for m in cb.multiplicands:
            w.write(m, cb.value_bits)


def _write_floor(w: BitWriter, fl):
    if isinstance(fl, Floor0):
        w.write(0, 16)
        w.write(fl.order, 8)
        w.write(fl.rate, 16)
        w.write(fl.bark_map_size, 16)
        w.write(fl.amplitude_bits, 6)
        w.write(fl.amplitude_offset, 8)
        w.write(len(fl.book_list) - 1, 4)
        for b in fl.book_list:
            w.write(b, 8)
        return
    w.write(1, 16)
    w.write(len(fl.partition_class), 5)
    for c in fl.partition_class:
        w.write(c, 4)
    for c in range(max(fl.partition_class) + 1 if fl.partition_class else 0):
        w.write(fl.class_dim[c] - 1, 3)
        w.write(fl.class_sub[c], 2)
        if fl.class_sub[c]:
            w.write(fl.class_master[c], 8)
        for j in range(1 << fl.class_sub[c]):
            w.write(fl.sub_books[c][j] + 1, 8)
    w.write(fl.multiplier - 1, 2)
    w.write(fl.rangebits, 4)
    for x in fl.x_rest:
        w.write(x, fl.rangebits)


def _write_residue(w: BitWriter, rs: Residue):
    w.write(rs.type, 16)
    w.write(rs.begin, 24)
    w.write(rs.end, 24)
    w.write(rs.partition_size - 1, 24)
    w.write(rs.classifications - 1, 6)
    w.write(rs.classbook, 8)
    casc = []
    for c in range(rs.classifications):
        bits = 0
        for p in range(8):
            if rs.books[c][p] >= 0:
                bits |= 1 << p
        casc.append(bits)
        w.write(bits & 7, 3)
        if bits >> 3:
            w.write(1, 1)
            w.write(bits >> 3, 5)
        else:
            w.write(0, 1)
    for c in range(rs.classifications):
        for p in range(7):  # pass 7 is never read from the header (src/header.rs:450)
            if casc[c] & (1 << p):
                w.write(rs.books[c][p], 8)


def _write_mapping(w: BitWriter, m: Mapping, channels: int):
    w.write(0, 16)
    nsub = len(m.submap_floor)
    if nsub > 1:
        w.write(1, 1)
        w.write(nsub - 1, 4)
    else:
        w.write(0, 1)
    if m.coupling:
        w.write(1, 1)
        w.write(len(m.coupling) - 1, 8)
        for mag, ang in m.coupling:
            w.write(mag, ilog(channels - 1))
            w.write(ang, ilog(channels - 1))
    else:
        w.write(0, 1)
    w.write(0, 2)
    if nsub > 1:
        for c in range(channels):
            w.write(m.mux[c], 4)
    for s in range(nsub):
        w.write(0, 8)
        w.write(m.submap_floor[s], 8)
        w.write(m.submap_residue[s], 8)


# --------------------------------------------------------------------------------------------
# codebook factories
# --------------------------------------------------------------------------------------------


def _laplace_probs(values, scale):
    v = np.abs(np.asarray(values, np.float64))
    p = np.exp(-v / scale)
    return p / p.sum()


def scalar_book(entries: int, scale: float = 3.0, max_len: int = 16) -> Codebook:
    """Entry-number book (floor Y values / class words): short codes for small numbers."""
    p = np.exp(-np.arange(entries) / scale)
    return Codebook(dims=1, lengths=huffman_lengths(p, max_len))


def uniform_book(entries: int) -> Codebook:
    return Codebook(dims=1, lengths=huffman_lengths(np.ones(entries), 24))


def class_book(nclass: int, words: int, scale: float = 6.0) -> Codebook:
    """Residue class book: `dims` = class words per code word, entries = nclass^words."""
    cb = scalar_book(nclass ** words, scale)
    cb.dims = words
    return cb


def vq_lattice_book(dims: int, values: Sequence[int], scale: float = 1.5, delta: float = 1.0,
                    max_len: int = 20, sequence_p: bool = False) -> Codebook:
    """Lookup-type-1 book over the lattice values^dims (src/header.rs:495-531)."""
    values = list(values)
    nv = len(values)
    entries = nv ** dims
    vmin = min(values)
    mult = [int(round((v - vmin) / delta)) for v in values]
    assert all(vmin + m * delta == v for m, v in zip(mult, values))
    # entry e -> element d = values[(e // nv^d) % nv]
    e = np.arange(entries)
    l1 = np.zeros(entries)
    for d in range(dims):
        l1 += np.abs(np.asarray(values, np.float64))[(e // (nv ** d)) % nv]
    p = np.exp(-l1 / scale)
    return Codebook(dims=dims, lengths=huffman_lengths(p, max_len), lookup_type=1, minimum=float(vmin), delta=float(delta),
                    value_bits=max(1, ilog(max(mult))), sequence_p=sequence_p, multiplicands=mult)


def vq_table_book(dims: int, table: np.ndarray, delta: float = 0.5, max_len: int = 16) -> Codebook:
    """Lookup-type-2 book with an explicit entries x dims table (multiples of delta)."""
    table = np.asarray(table, np.float64)
    vmin = float(table.min())
    mult = np.rint((table - vmin) / delta).astype(int).reshape(-1)
    p = np.exp(-np.abs(table).sum(axis=1) / 2.0)
    return Codebook(dims=dims, lengths=huffman_lengths(p, max_len), lookup_type=2, minimum=vmin, delta=delta,
                    value_bits=max(1, ilog(int(mult.max()))), multiplicands=[int(m) for m in mult])


# --------------------------------------------------------------------------------------------
# stream setups for the BASELINE.json configs (shapes follow SURVEY.md section 8d)
# --------------------------------------------------------------------------------------------

LONG_X = [93, 23, 372, 6, 46, 186, 750, 14, 33, 65, 130, 260, 556, 3, 10, 18, 28, 39, 55, 79, 111, 158, 220, 312, 464, 650, 850]
SHORT_X = [14, 4, 58, 2, 8, 28, 90]


def _scale_x(base, base_half, half):
    """Scale a post list made for `base_half` spectral lines to `half` lines, keeping values distinct and in (0, half)."""
    out, seen = [], {0, half}
    for x in base:
        v = x * half // base_half
        if v in seen or v <= 0 or v >= half:
            continue
        seen.add(v)
        out.append(v)
    return out


def _std_books():
    """Shared codebook set. Returns (list, index dict)."""
    books, idx = [], {}

    def add(name, cb):
        idx[name] = len(books)
        books.append(cb)

    add("y16", scalar_book(16, 3.0))            # floor Y, values 0..15
    add("y32", scalar_book(32, 5.0))            # floor Y, values 0..31
    add("master8", uniform_book(8))             # floor class master book, cbits=1 cdim=3
    add("class16", class_book(4, 2))            # residue classbook: 4 classes, 2 class words per code word
    add("vq2", vq_lattice_book(2, range(-8, 9), 2.0))     # 289 entries
    add("vq4", vq_lattice_book(4, range(-2, 3), 1.2))     # 625 entries
    add("vq8", vq_lattice_book(8, range(-1, 2), 0.9))     # 6561 entries
    add("vq2fine", vq_lattice_book(2, [-1.0, -0.5, 0.0, 0.5, 1.0], 0.6, delta=0.5))
    return books, idx


def _floor1(xs, rangebits, mult, y_book, master_book, alt_book):
    nrest = len(xs)
    # classes: 0 = dim 3 no subclasses; 1 = dim 3, one subclass bit (sub-book 0 = none -> Y = 0, sub-book 1 = alt_book)
    pcs, remaining = [], nrest
    k = 0
    while remaining >= 3:
        pcs.append(k % 2)
        remaining -= 3
        k += 1
    dims = [3, 3]
    if remaining:
        pcs.append(2)
        dims.append(remaining)
    return Floor1(partition_class=pcs, class_dim=dims, class_sub=[0, 1] + ([0] if remaining else []),
                  class_master=[0, master_book] + ([0] if remaining else []),
                  sub_books=[[y_book], [-1, alt_book]] + ([[y_book]] if remaining else []),
                  multiplier=mult, rangebits=rangebits, x_rest=list(xs))


def _res_books(ix):
    none = [-1] * 8
    return [
        list(none),                                                    # class 0: silent partition
        [ix["vq4"]] + [-1] * 7,                                        # class 1: one pass
        [ix["vq2"], ix["vq2fine"]] + [-1] * 6,                         # class 2: two passes
        [ix["vq8"], ix["vq4"], ix["vq2fine"]] + [-1] * 5,              # class 3: three passes
    ]


def stereo_setup(sample_rate: int = 44100, bs0: int = 8, bs1: int = 11, residue_type: int = 2,
                 single_entry_book: bool = False) -> StreamSetup:
    """2 ch, bs 8/11 (the sizes of lewton's ident-header test, src/header.rs:264-276).
    single_entry_book: the second pass of class 2 uses a one-entry codebook (one bit per codeword, either value,
    src/huffman_tree.rs:202-217)."""
    books, ix = _std_books()
    if single_entry_book:
        ix["vq2fine"] = len(books)
        books.append(vq_lattice_book(2, [0.5], 1.0, delta=0.5))
    n0h, n1h = (1 << bs0) // 2, (1 << bs1) // 2
    fl_short = _floor1(_scale_x(SHORT_X, 128, n0h), bs0 - 1, 4, ix["y16"], ix["master8"], ix["y16"])
    fl_long = _floor1(_scale_x(LONG_X, 1024, n1h), bs1 - 1, 2, ix["y16"], ix["master8"], ix["y32"])
    mulch = 2 if residue_type == 2 else 1
    rs_short = Residue(residue_type, 0, mulch * (n0h * 13 // 16), 16, 4, ix["class16"], _res_books(ix))
    rs_long = Residue(residue_type, 0, mulch * 800 * n1h // 1024, 32, 4, ix["class16"], _res_books(ix))
    maps = [Mapping([(0, 1)], [0, 0], [0], [0]), Mapping([(0, 1)], [0, 0], [1], [1])]
    return StreamSetup(2, sample_rate, bs0, bs1, books, [fl_short, fl_long], [rs_short, rs_long], maps,
                       [Mode(0, 0), Mode(1, 1)])


def surround51_setup(sample_rate: int = 48000, bs0: int = 8, bs1: int = 11) -> StreamSetup:
    """6 ch: submap 0 = channels 0-4 (floor A, type-2 residue), submap 1 = LFE (own floor, type-1 residue);
    coupling steps (0,2),(3,4) applied in reverse order at decode (BASELINE config 4)."""
    books, ix = _std_books()
    n0h, n1h = (1 << bs0) // 2, (1 << bs1) // 2
    fl_short = _floor1(_scale_x(SHORT_X, 128, n0h), bs0 - 1, 4, ix["y16"], ix["master8"], ix["y16"])
    fl_long = _floor1(_scale_x(LONG_X, 1024, n1h), bs1 - 1, 2, ix["y16"], ix["master8"], ix["y32"])
    fl_lfe_s = _floor1([16, 64, 32], bs0 - 1, 1, ix["y32"], ix["master8"], ix["y16"])
    fl_lfe_l = _floor1([64, 16, 256, 128, 32, 512], bs1 - 1, 1, ix["y32"], ix["master8"], ix["y16"])
    rs = [
        Residue(2, 0, 5 * (n0h * 13 // 16), 16, 4, ix["class16"], _res_books(ix)),  # short main
        Residue(2, 0, 5 * 800 * n1h // 1024, 32, 4, ix["class16"], _res_books(ix)),  # long main
        Residue(1, 0, n0h // 4, 8, 4, ix["class16"], _res_books(ix)),               # short LFE
        Residue(0, 0, n1h // 8, 32, 4, ix["class16"], _res_books(ix)),              # long LFE (type 0)
    ]
    mux = [0, 0, 0, 0, 0, 1]
    maps = [Mapping([(0, 2), (3, 4)], mux, [0, 2], [0, 2]), Mapping([(0, 2), (3, 4)], mux, [1, 3], [1, 3])]
    return StreamSetup(6, sample_rate, bs0, bs1, books, [fl_short, fl_long, fl_lfe_s, fl_lfe_l], rs, maps,
                       [Mode(0, 0), Mode(1, 1)])


def multichannel_setup(channels: int = 12, sample_rate: int = 48000, bs0: int = 8, bs1: int = 11) -> StreamSetup:
    """`channels` (> 8, e.g. 7.1.4 = 12) channels: submap 0 = all but the last channel (one floor, type-2 residue interleaving
    channels - 1 vectors), submap 1 = the last channel (own floor, type-1 residue); coupling steps on pairs (0,1), (2,3), ...
    of the main channels.  blocksize_1 * channels stays below 65536 (the reference's u16 product, audio.rs:745)."""
    assert 2 < channels <= 16 and (1 << bs1) * channels < 65536
    books, ix = _std_books()
    n0h, n1h = (1 << bs0) // 2, (1 << bs1) // 2
    main = channels - 1
    fl_short = _floor1(_scale_x(SHORT_X, 128, n0h), bs0 - 1, 4, ix["y16"], ix["master8"], ix["y16"])
    fl_long = _floor1(_scale_x(LONG_X, 1024, n1h), bs1 - 1, 2, ix["y16"], ix["master8"], ix["y32"])
    fl_lfe_s = _floor1([16, 64, 32], bs0 - 1, 1, ix["y32"], ix["master8"], ix["y16"])
    fl_lfe_l = _floor1([64, 16, 256, 128, 32, 512], bs1 - 1, 1, ix["y32"], ix["master8"], ix["y16"])
    rs = [
        Residue(2, 0, main * (n0h * 13 // 16), 16, 4, ix["class16"], _res_books(ix)),   # short main
        Residue(2, 0, main * 600 * n1h // 1024, 32, 4, ix["class16"], _res_books(ix)),  # long main
        Residue(1, 0, n0h // 4, 8, 4, ix["class16"], _res_books(ix)),                   # short last channel
        Residue(1, 0, n1h // 8, 32, 4, ix["class16"], _res_books(ix)),                  # long last channel
    ]
    mux = [0] * main + [1]
    coupling = [(c, c + 1) for c in range(0, main - 1, 2)]
    maps = [Mapping(coupling, mux, [0, 2], [0, 2]), Mapping(coupling, mux, [1, 3], [1, 3])]
    return StreamSetup(channels, sample_rate, bs0, bs1, books, [fl_short, fl_long, fl_lfe_s, fl_lfe_l], rs, maps,
                       [Mode(0, 0), Mode(1, 1)])


def mono_setup(bs0: int = 6, bs1: int = 9, sample_rate: int = 8000) -> StreamSetup:
    """1 ch, small blocks, residue types 0 and 1, a begin offset, dims not dividing the partition size."""
    books, ix = _std_books()
    n0h, n1h = (1 << bs0) // 2, (1 << bs1) // 2
    fl_short = _floor1([n0h // 2, n0h // 4, 3 * n0h // 4], bs0 - 1, 3, ix["y16"], ix["master8"], ix["y16"])
    fl_long = _floor1([n1h // 2, n1h // 4, 3 * n1h // 4, n1h // 8, 5, 100], bs1 - 1, 1, ix["y32"], ix["master8"], ix["y16"])
    rb = _res_books(ix)
    rs = [Residue(0, 4, n0h - 4, 8, 4, ix["class16"], rb), Residue(1, 8, n1h + 40, 24, 4, ix["class16"], rb)]
    maps = [Mapping([], [0], [0], [0]), Mapping([], [0], [1], [1])]
    return StreamSetup(1, sample_rate, bs0, bs1, books, [fl_short, fl_long], rs, maps, [Mode(0, 0), Mode(1, 1)])


def floor0_setup(bs0: int = 7, bs1: int = 10, sample_rate: int = 22050, mixed: bool = False) -> StreamSetup:
    """2 ch with floor type 0 (LSP, src/audio.rs:109-212): even order for short blocks, odd order and a two-book list for
    long blocks.  `mixed`: short blocks use a floor 1 instead, so both floor kinds occur in one stream."""
    books, ix = _std_books()
    rng = np.random.default_rng(77)
    # LSP angles: ascending within a vector, the last element advances the running offset (audio.rs:131-147), so the
    # angles spread over (0, pi) for the orders used below and p + q stays away from zero
    ix["lsp2"] = len(books)
    books.append(vq_table_book(2, np.round(np.sort(rng.uniform(0.3, 0.8, (32, 2)), axis=1) * 64) / 64, delta=1 / 64))
    ix["lsp3"] = len(books)
    books.append(vq_table_book(3, np.round(np.sort(rng.uniform(0.3, 1.0, (64, 3)), axis=1) * 64) / 64, delta=1 / 64))
    n0h, n1h = (1 << bs0) // 2, (1 << bs1) // 2
    if mixed:
        fl_short = _floor1([n0h // 2, n0h // 4, 3 * n0h // 4], bs0 - 1, 2, ix["y16"], ix["master8"], ix["y16"])
    else:
        fl_short = Floor0(order=8, rate=sample_rate, bark_map_size=n0h // 2, amplitude_bits=4, amplitude_offset=12,
                          book_list=[ix["lsp2"]], amp_max=2)
    fl_long = Floor0(order=9, rate=sample_rate, bark_map_size=n1h // 4, amplitude_bits=5, amplitude_offset=15,
                     book_list=[ix["lsp3"], ix["lsp2"]], amp_max=2)
    rs_short = Residue(2, 0, 2 * (n0h * 13 // 16), 16, 4, ix["class16"], _res_books(ix))
    rs_long = Residue(1, 0, 800 * n1h // 1024, 32, 4, ix["class16"], _res_books(ix))
    maps = [Mapping([(0, 1)], [0, 0], [0], [0]), Mapping([(0, 1)], [0, 0], [1], [1])]
    return StreamSetup(2, sample_rate, bs0, bs1, books, [fl_short, fl_long], [rs_short, rs_long], maps,
                       [Mode(0, 0), Mode(1, 1)])


# --------------------------------------------------------------------------------------------
# audio packet writer -- emits symbols in decode order (src/audio.rs:921-986)
# --------------------------------------------------------------------------------------------


class PacketWriter:
    def __init__(self, setup: StreamSetup, seed: int = 0, y01_range=(45, 95), p_floor_unused: float = 0.0,
                 p_zero_y: float = 0.7, class_probs=(0.35, 0.3, 0.2, 0.15), vq_scale: float = 1.0):
        self.s = setup
        self.rng = np.random.default_rng(seed)
        self.y01_range = y01_range
        self.p_unused = p_floor_unused
        self.p_zero_y = p_zero_y
        self.class_probs = np.asarray(class_probs, np.float64)
        self.vq_scale = vq_scale
        self._vq_cdf = {}

    # -- helpers
    def _huff(self, w: BitWriter, book: int, entry: int):
        cw = self.s.codebooks[book].codewords()[entry]
        assert cw is not None, (book, entry)
        w.write(cw[0], cw[1])

    def _draw_vq_entries(self, book: int, count: int):
        """Entries drawn with probability 2^-len (what the code lengths were built for)."""
        if book not in self._vq_cdf:
            cb = self.s.codebooks[book]
            used = np.array(cb.used_entries())
            p = 2.0 ** (-np.array([cb.lengths[i] for i in used], np.float64) * self.vq_scale)
            self._vq_cdf[book] = (used, np.cumsum(p / p.sum()))
        used, cdf = self._vq_cdf[book]
        return used[np.searchsorted(cdf, self.rng.random(count), side="right").clip(0, len(used) - 1)]

    def _draw_y(self, book: int):
        cb = self.s.codebooks[book]
        if self.rng.random() < self.p_zero_y:
            return 0
        return int(min(cb.entries - 1, self.rng.integers(1, 13)))

    def _floor1(self, w: BitWriter, fl: Floor1) -> bool:
        if self.rng.random() < self.p_unused:
            w.write(0, 1)
            return False
        w.write(1, 1)
        rng_ = [256, 128, 86, 64][fl.multiplier - 1]
        b = ilog(rng_ - 1)
        lo, hi = self.y01_range
        scale = rng_ / 128.0  # y01_range is quoted for multiplier 2 (range 128)
        for _ in range(2):
            w.write(int(np.clip(self.rng.integers(lo, hi + 1) * scale, 0, rng_ - 1)), b)
        for c in fl.partition_class:
            cdim, cbits = fl.class_dim[c], fl.class_sub[c]
            cval = 0
            if cbits:
                mb = fl.class_master[c]
                cval = int(self.rng.choice(self.s.codebooks[mb].used_entries()))
                self._huff(w, mb, cval)
            for _ in range(cdim):
                book = fl.sub_books[c][cval & ((1 << cbits) - 1)]
                cval >>= cbits
                if book >= 0:
                    self._huff(w, book, self._draw_y(book))
        return True

    def _floor0(self, w: BitWriter, fl: Floor0) -> bool:
        if self.rng.random() < self.p_unused:
            w.write(0, fl.amplitude_bits)
            return False
        hi = 1 << min(fl.amplitude_bits, 30)
        w.write(int(self.rng.integers(1, min(hi, fl.amp_max + 1) if fl.amp_max else hi)), fl.amplitude_bits)
        bn = int(self.rng.integers(0, len(fl.book_list)))
        w.write(bn, ilog(len(fl.book_list)))
        cb_i = fl.book_list[bn]
        cb = self.s.codebooks[cb_i]
        n = 0
        while n < fl.order:
            self._huff(w, cb_i, int(self._draw_vq_entries(cb_i, 1)[0]))
            n += cb.dims
        return True

    def _residue(self, w: BitWriter, rs: Residue, n: int, dnd: List[bool]):
        ch = len(dnd)
        if rs.type == 2:
            if all(dnd):
                return
            n = n * ch
            dnd = [False]
            ch = 1
        size = n // 2
        begin, end = min(rs.begin, size), min(rs.end, size)
        parts = (end - begin) // rs.partition_size
        if end - begin == 0:
            return
        classbook = self.s.codebooks[rs.classbook]
        cpc = classbook.dims
        # the classbook's `dims` is the number of class words per code word (src/audio.rs:634)
        ncls = rs.classifications
        cls = np.zeros((ch, parts + cpc), np.int64)
        for p_ in range(8):
            pc = 0
            while pc < parts:
                if p_ == 0:
                    for j in range(ch):
                        if dnd[j]:
                            continue
                        cw = self.rng.choice(ncls, size=cpc, p=self.class_probs[:ncls] / self.class_probs[:ncls].sum())
                        sym = 0
                        for c in cw:
                            sym = sym * ncls + int(c)
                        sym = min(sym, classbook.entries - 1)
                        # decode exactly what the decoder will see
                        t = sym
                        for i in range(cpc - 1, -1, -1):
                            cls[j, pc + i] = t % ncls
                            t //= ncls
                        self._huff(w, rs.classbook, sym)
                for _ in range(cpc):
                    if pc >= parts:
                        break
                    for j in range(ch):
                        if dnd[j]:
                            continue
                        book = rs.books[int(cls[j, pc])][p_]
                        if book >= 0:
                            cb = self.s.codebooks[book]
                            offs = begin + pc * rs.partition_size
                            if rs.type == 0:
                                cnt = rs.partition_size // cb.dims
                            else:
                                # one code word per `dims` values until the partition is covered; the word that
                                # would run past the end of the vector is still read (src/audio.rs:602-608)
                                cnt, i = 0, 0
                                while i < rs.partition_size:
                                    cnt += 1
                                    if i + cb.dims > size - offs:
                                        break
                                    i += cb.dims
                            for e in self._draw_vq_entries(book, cnt):
                                self._huff(w, book, int(e))
                    pc += 1

    def packet(self, mode: int, prev_flag: int = 1, next_flag: int = 1) -> bytes:
        s = self.s
        w = BitWriter()
        w.write(0, 1)
        w.write(mode, ilog(len(s.modes) - 1))
        md = s.modes[mode]
        mp = s.mappings[md.mapping]
        n = 1 << (s.bs1 if md.blockflag else s.bs0)
        if md.blockflag:
            w.write(prev_flag, 1)
            w.write(next_flag, 1)
        used = []
        for c in range(s.channels):
            fl = s.floors[mp.submap_floor[mp.mux[c]]]
            used.append(self._floor0(w, fl) if isinstance(fl, Floor0) else self._floor1(w, fl))
        no_res = [not u for u in used]
        for mag, ang in mp.coupling:
            if not (no_res[mag] and no_res[ang]):
                no_res[mag] = no_res[ang] = False
        for sm in range(len(mp.submap_floor)):
            dnd = [no_res[c] for c in range(s.channels) if mp.mux[c] == sm]
            self._residue(w, s.residues[mp.submap_residue[sm]], n, dnd)
        return w.bytes()


def block_sequence(pattern: str, count: int) -> List[Tuple[int, int, int]]:
    """Expand a cyclic pattern of 'L'/'S' into `count` (blockflag, prev_flag, next_flag) triples whose
    window flags are consistent with the neighbouring blocks (the stream is treated as cyclic)."""
    seq = [pattern[i % len(pattern)] for i in range(count)]
    out = []
    for i, b in enumerate(seq):
        if b == "S":
            out.append((0, 0, 0))
        else:
            prv = seq[i - 1] if i > 0 else "L"
            nxt = seq[i + 1] if i + 1 < count else "L"
            out.append((1, 1 if prv == "L" else 0, 1 if nxt == "L" else 0))
    return out


def make_stream(setup: StreamSetup, pattern: str, count: int, seed: int = 0, **kw) -> List[bytes]:
    """Audio packets of one logical stream. Mode 0 = short, mode 1 = long by convention of the setups above."""
    pw = PacketWriter(setup, seed, **kw)
    short_mode = next(i for i, m in enumerate(setup.modes) if not m.blockflag)
    long_mode = next(i for i, m in enumerate(setup.modes) if m.blockflag)
    return [pw.packet(long_mode if bf else short_mode, pf, nf) for bf, pf, nf in block_sequence(pattern, count)]


# --------------------------------------------------------------------------------------------
# Ogg encapsulation (RFC 3533) -- used for BASELINE config 1 plumbing and the inside_ogg row
# --------------------------------------------------------------------------------------------

_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = []
        for i in range(256):
            r = i << 24
            for _ in range(8):
                r = ((r << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if r & 0x80000000 else (r << 1) & 0xFFFFFFFF
            t.append(r)
        _CRC_TABLE = t
    return _CRC_TABLE


def ogg_crc(data: bytes) -> int:
    t = _crc_table()
    crc = 0
    for b in data:
        crc = ((crc << 8) & 0xFFFFFFFF) ^ t[((crc >> 24) & 0xFF) ^ b]
    return crc


def ogg_page(packets: Sequence[bytes], serial: int, seq: int, granule: int, bos=False, eos=False) -> bytes:
    segs = []
    for p in packets:
        n = len(p)
        segs += [255] * (n // 255) + [n % 255]
    assert len(segs) <= 255
    hdr = bytearray(b"OggS\x00" + bytes([(2 if bos else 0) | (4 if eos else 0)]) +
                    struct.pack("<qIII", granule, serial, seq, 0) + bytes([len(segs)]) + bytes(segs))
    body = b"".join(packets)
    crc = ogg_crc(bytes(hdr) + body)
    hdr[22:26] = struct.pack("<I", crc)
    return bytes(hdr) + body


def ogg_stream(setup: StreamSetup, audio_packets: Sequence[bytes], sample_counts: Sequence[int], serial: int = 0x4C57,
               packets_per_page: int = 8, final_trim: int = 0) -> bytes:
    """Wrap headers + audio packets into Ogg pages; `sample_counts[i]` = per-channel samples packet i yields."""
    idp, cmt, stp = setup.headers()
    out = [ogg_page([idp], serial, 0, 0, bos=True), ogg_page([cmt, stp], serial, 1, 0)]
    seq, gp = 2, 0
    for i in range(0, len(audio_packets), packets_per_page):
        chunk = audio_packets[i:i + packets_per_page]
        gp += sum(sample_counts[i:i + packets_per_page])
        last = i + packets_per_page >= len(audio_packets)
        out.append(ogg_page(chunk, serial, seq, gp - (final_trim if last else 0), eos=last))
        seq += 1
    return b"".join(out)


# --------------------------------------------------------------------------------------------
# random setup headers (round 6): draws from the whole space src/header.rs accepts -- floors :771-918 (2 .. 65 posts, any
# distinct x list, multipliers 1-4, class / subclass structures with and without master books, floor 0 mixed in), residues
# :922-981 (types 0/1/2, begin > 0, end below / above the vector, partition sizes that the books' dimensions do not divide,
# class books of 1-4 words, sparse cascades), mappings :985-1058 (any coupling list incl. a channel in several steps and
# chains, 1-3 submaps), modes :1060-1080 (several long / short modes with their own mappings), codebooks :673-768 (ordered
# and sparse length lists, lookup types 1 and 2, sequence_p, entry counts that are no power of the lookup values, one-entry
# books).  Every draw is legal: a setup the product's parser rejects is a finding, not noise.
# --------------------------------------------------------------------------------------------


def _rand_lengths(rng, weights, max_len):
    n = len(weights)
    if n == 1:
        return [1]
    need = int(np.ceil(np.log2(n))) + 1
    return huffman_lengths(np.asarray(weights, np.float64) + 1e-300, max(max_len, need))


def _finish_book(rng, cb: Codebook, allow_sparse=True) -> Codebook:
    """Randomise how the length list is coded: ordered (lengths sorted ascending -- any order of a complete set of lengths is a
    complete tree), sparse with unused entries, or plain."""
    r = rng.random()
    n = len(cb.lengths)
    if r < 0.15 and n >= 2:
        cb.lengths = sorted(cb.lengths)
        cb.ordered = True
    elif r < 0.25 and allow_sparse:
        cb.sparse = True          # the sparse flag with every entry used
    return cb


def random_scalar_book(rng, entries: int, max_len: int = 0, sparse_ok: bool = False) -> Codebook:
    """A book that is only read for its entry NUMBER (floor-1 Y values, floor class master books, residue class words)."""
    entries = max(1, int(entries))
    max_len = max_len or int(rng.choice([6, 10, 16, 24, 32]))
    used = np.ones(entries, bool)
    if sparse_ok and entries >= 4 and rng.random() < 0.3:
        used = rng.random(entries) < 0.7
        used[int(rng.integers(0, entries))] = True
        if used.sum() < 2:
            used[:2] = True
    shape = rng.random()
    if shape < 0.6:
        w = np.exp(-np.arange(entries) / max(0.7, entries * float(rng.uniform(0.08, 0.5))))
    elif shape < 0.85:
        w = np.ones(entries)
    else:
        w = rng.random(entries) ** 4 + 1e-4
    ls = np.zeros(entries, int)
    ls[used] = _rand_lengths(rng, w[used], max_len)
    cb = Codebook(dims=int(rng.choice([1, 1, 1, 2, 3])), lengths=[int(x) for x in ls])
    if not used.all():
        cb.sparse = True
        return cb
    return _finish_book(rng, cb)


def random_vq_book(rng, small: bool = False) -> Codebook:
    """A book with a vector lookup (residue passes, floor-0 coefficients)."""
    max_len = int(rng.choice([8, 12, 16, 20, 32]))
    if rng.random() < 0.04:                              # one entry: one bit per codeword, either value (huffman_tree.rs:202-217)
        dims = int(rng.choice([1, 2, 4, 8]))
        return Codebook(dims=dims, lengths=[1], lookup_type=2, minimum=-1.0, delta=0.5, value_bits=3,
                        multiplicands=[int(x) for x in rng.integers(0, 5, dims)])
    cap = 600 if small else 4096
    if rng.random() < 0.7:
        dims = int(rng.choice([1, 2, 2, 2, 4, 4, 4, 8, 8, 3, 5, 6, 12]))
        nv_max = max(1, int(np.floor(cap ** (1.0 / dims) + 1e-9)))
        while (nv_max + 1) ** dims <= cap:
            nv_max += 1
        nv = int(rng.integers(max(1, min(2, nv_max)), min(nv_max, 17) + 1))
        entries = nv ** dims
        if rng.random() < 0.25 and nv >= 2 and (nv + 1) ** dims - entries > 1:   # more entries than lookup_values ^ dims: the index wraps (header.rs:504)
            entries += int(rng.integers(1, min(9, (nv + 1) ** dims - entries)))
        delta = float(rng.choice([1.0, 1.0, 0.5, 0.25, 2.0]))
        order = np.arange(nv)
        if rng.random() < 0.2:
            rng.shuffle(order)
        seq = rng.random() < 0.12
        vmin = -delta * (nv // 2) if not seq else -delta * float(rng.integers(0, 3))
        if rng.random() < 0.15:
            vmin += delta * float(rng.integers(-2, 3))
        vals = vmin + order * delta
        e = np.arange(entries)
        l1 = np.zeros(entries)
        for d in range(dims):
            l1 += np.abs(vals[(e // (nv ** d)) % nv])
        w = np.exp(-l1 / (delta * float(rng.uniform(0.6, 2.5))))
        cb = Codebook(dims=dims, lengths=_rand_lengths(rng, w, max_len), lookup_type=1, minimum=float(vmin), delta=delta,
                      value_bits=max(1, ilog(int(order.max()))) + int(rng.integers(0, 3)), sequence_p=seq,
                      multiplicands=[int(x) for x in order])
        return _finish_book(rng, cb)
    dims = int(rng.choice([1, 2, 3, 4, 8]))
    entries = int(rng.integers(2, 257 if not small else 65))
    delta = float(rng.choice([1.0, 0.5, 0.25]))
    table = rng.integers(-8, 9, (entries, dims)) * delta
    if rng.random() < 0.5:
        table = np.where(rng.random((entries, dims)) < 0.5, 0.0, table)
    vmin = float(table.min())
    mult = np.rint((table - vmin) / delta).astype(int).reshape(-1)
    used = np.ones(entries, bool)
    if entries >= 4 and rng.random() < 0.25:
        used = rng.random(entries) < 0.75
        used[:2] = True
    w = np.exp(-np.abs(table).sum(axis=1) / (delta * 3.0))
    ls = np.zeros(entries, int)
    ls[used] = _rand_lengths(rng, w[used], max_len)
    cb = Codebook(dims=dims, lengths=[int(x) for x in ls], lookup_type=2, minimum=vmin, delta=delta,
                  value_bits=max(1, ilog(int(mult.max()))), multiplicands=[int(m) for m in mult])
    if not used.all():
        cb.sparse = True
        return cb
    return _finish_book(rng, cb)


def _random_x_rest(rng, count: int, rangebits: int) -> List[int]:
    hi = (1 << rangebits) - 1            # legal values 1 .. hi, all distinct (header.rs:885-900)
    assert count <= hi
    if count == 0:
        return []
    if rng.random() < 0.5 or count * 3 > hi:
        xs = rng.choice(np.arange(1, hi + 1), size=count, replace=False)
    else:                                # dense at the low end, like an encoder's lists
        seen = set()
        while len(seen) < count:
            v = int(np.exp(rng.uniform(0.0, np.log(hi + 0.999))))
            if 1 <= v <= hi:
                seen.add(v)
        xs = np.array(sorted(seen))
        rng.shuffle(xs)
    return [int(x) for x in xs]


def random_floor1(rng, books: List[Codebook], bs: int, posts: Optional[int] = None) -> Floor1:
    """`books` is appended to.  `posts`: total post count F (2 .. 65); default random."""
    if posts is None:
        r = rng.random()
        posts = int(2 if r < 0.04 else rng.integers(3, 9) if r < 0.2 else rng.integers(9, 33) if r < 0.65 else
                    rng.integers(33, 65) if r < 0.93 else 65)
    rest = posts - 2
    dims_list = []
    while rest > 0:
        lo = max(1, rest - 8 * (30 - len(dims_list)))      # at most 31 partitions (5-bit count)
        d = int(rng.integers(lo, min(8, rest) + 1))
        dims_list.append(d)
        rest -= d
    mult = int(rng.integers(1, 5))
    range_ = [256, 128, 86, 64][mult - 1]
    # classes: one or more per distinct dimension, at most 16
    class_dim, class_sub, class_master, sub_books = [], [], [], []
    by_dim = {}
    for d in sorted(set(dims_list)):
        for _ in range(int(rng.integers(1, 3)) if len(class_dim) + len(set(dims_list)) < 14 else 1):
            if len(class_dim) == 16:
                break
            by_dim.setdefault(d, []).append(len(class_dim))
            cbits = int(rng.choice([0, 0, 1, 1, 2, 3]))
            class_dim.append(d)
            class_sub.append(cbits)
            if cbits:
                ent = int(rng.choice([1 << min(8, cbits * d), int(rng.integers(1, 40))]))
                class_master.append(len(books))
                books.append(random_scalar_book(rng, max(1, ent), sparse_ok=True))
            else:
                class_master.append(0)
            sb = []
            for _j in range(1 << cbits):
                if rng.random() < 0.25:
                    sb.append(-1)
                else:
                    r = rng.random()
                    ent = int(rng.choice([4, 8, 16, 32]) if r < 0.6 else range_ if r < 0.9 else rng.integers(range_, 2 * range_ + 40))
                    sb.append(len(books))
                    books.append(random_scalar_book(rng, ent, sparse_ok=True))
            sub_books.append(sb)
    perm = list(range(len(class_dim)))
    rng.shuffle(perm)                                        # class numbers in no particular order
    inv = {old: new for new, old in enumerate(perm)}
    class_dim = [class_dim[o] for o in perm]
    class_sub = [class_sub[o] for o in perm]
    class_master = [class_master[o] for o in perm]
    sub_books = [sub_books[o] for o in perm]
    pcs = [inv[int(rng.choice(by_dim[d]))] for d in dims_list]
    rb_min = max(1, ilog(posts - 2)) if posts > 2 else 0
    r = rng.random()
    rangebits = int(bs - 1 if r < 0.6 else rng.integers(rb_min, 16) if r < 0.8 else max(rb_min, bs - 2) if r < 0.9 else min(15, bs))
    rangebits = max(rangebits, rb_min)
    return Floor1(partition_class=pcs, class_dim=class_dim, class_sub=class_sub, class_master=class_master,
                  sub_books=sub_books, multiplier=mult, rangebits=rangebits, x_rest=_random_x_rest(rng, posts - 2, rangebits))


def random_floor0(rng, books: List[Codebook], bs: int, sample_rate: int) -> Floor0:
    """Floor type 0 with coefficient books made for its order: element i of a vector is (i + 1 +- 0.3) steps of pi / (order + 1),
    and the last element of a vector advances the running offset (audio.rs:131-147), so the LSP angles ascend, stay inside
    (0, pi) and the roots of P and Q interlace -- p + q stays away from zero and the curve finite, as for an encoder's streams."""
    order = int(rng.integers(2, 17))
    step = np.pi / (order + 1)
    n_books = int(rng.integers(1, 4))
    bl = []
    for _ in range(n_books):
        dims = int(rng.choice([1, 2, 3, 4]))
        ent = int(rng.integers(4, 65))
        tab = np.round((np.arange(1, dims + 1)[None, :] + rng.uniform(-0.3, 0.3, (ent, dims))) * step * 256) / 256
        bl.append(len(books))
        books.append(vq_table_book(dims, tab, delta=1 / 256, max_len=int(rng.choice([8, 16]))))
    half = (1 << bs) // 2
    return Floor0(order=order, rate=sample_rate, bark_map_size=int(rng.choice([max(2, half // 4), max(2, half // 2), half, 2 * half, 37])),
                  amplitude_bits=int(rng.integers(4, 9)), amplitude_offset=int(rng.integers(8, 21)), book_list=bl,
                  amp_max=int(rng.integers(1, 3)))


def random_residue(rng, books: List[Codebook], vq_pool: List[int], bs: int, ch_sub: int) -> Residue:
    rtype = int(rng.choice([0, 1, 2, 2]))
    size = ((1 << bs) // 2) * (ch_sub if rtype == 2 else 1)
    r = rng.random()
    begin = 0 if r < 0.6 else int(rng.integers(1, max(2, size // 4))) if r < 0.95 else size + int(rng.integers(0, 50))
    r = rng.random()
    end = (int(size * rng.uniform(0.4, 1.0)) if r < 0.6 else size if r < 0.7 else size + int(rng.integers(1, 5000)) if r < 0.85
           else int(rng.integers(0, size + 1)) if r < 0.97 else (1 << 24) - 1)
    end = max(end, begin)
    if rng.random() < 0.03:
        end = begin
    r = rng.random()
    psize = int(rng.choice([4, 8, 16, 32, 64]) if r < 0.6 else rng.integers(1, 65) if r < 0.93 else rng.integers(65, 400))
    ncls = int(rng.choice([1, 2, 3, 4, 4, 5, 8, 10, 16]))
    words = int(rng.choice([1, 2, 2, 3, 4]))
    while ncls ** words > 4096:
        words -= 1
    ent = ncls ** words
    r = rng.random()
    if r < 0.15:
        ent += int(rng.integers(1, 20))                    # entries above classifications ^ words: the digits still come out in range
    elif r < 0.25 and ent > 2:
        ent -= int(rng.integers(1, max(2, ent // 3)))
    cbk = random_scalar_book(rng, ent)
    cbk.dims = words
    classbook = len(books)
    books.append(cbk)
    rows = []
    p_bit = float(rng.choice([0.15, 0.3, 0.5]))
    for _c in range(ncls):
        row = [-1] * 8
        if rng.random() > 0.2:                             # (a class without any book: a silent partition)
            for p in range(7):
                if rng.random() < p_bit / (1 + 0.5 * p):
                    row[p] = int(rng.choice(vq_pool))
            if rng.random() < 0.04 and books[0].lookup_type:
                row[7] = 0                                 # the cascade's bit 7: no book number is read for it, book 0 is used (header.rs:449-466)
        rows.append(row)
    return Residue(rtype, begin, end, psize, ncls, classbook, rows)


def random_coupling(rng, channels: int) -> List[Tuple[int, int]]:
    if channels < 2:
        return []
    r = rng.random()
    if r < 0.2:
        return []
    if r < 0.6:                                            # disjoint pairs, some channels left out
        chans = list(range(channels))
        rng.shuffle(chans)
        n = int(rng.integers(1, channels // 2 + 1))
        steps = [(chans[2 * i], chans[2 * i + 1]) for i in range(n)]
        rng.shuffle(steps)
        return [(int(m), int(a)) for m, a in steps]
    if r < 0.75 and channels >= 3:                         # a chain: 0-1, 1-2, ...
        n = int(rng.integers(2, channels))
        return [(i, i + 1) if rng.random() < 0.5 else (i + 1, i) for i in range(n)]
    steps = []
    for _ in range(int(rng.integers(1, channels + 2))):   # anything legal: repeated pairs, a channel in several steps
        m = int(rng.integers(0, channels))
        a = int(rng.integers(0, channels - 1))
        steps.append((m, a if a < m else a + 1))
    return steps


_RANDOM_BS = [(8, 11)] * 7 + [(8, 10), (9, 10), (8, 9), (9, 12), (10, 12), (8, 12), (8, 13), (6, 13), (7, 7), (6, 9), (6, 6), (11, 11),
                              (10, 10), (9, 11), (10, 11), (7, 10), (6, 8), (13, 13), (12, 13), (7, 12)]


def random_setup(rng, channels: Optional[int] = None, blocksizes: Optional[Tuple[int, int]] = None, allow_floor0: bool = True
                 ) -> StreamSetup:
    """One draw from the space of legal ident + setup headers (see the section comment).  `rng`: numpy Generator."""
    while True:
        st = _random_setup_once(rng, channels, blocksizes, allow_floor0)
        if len(st.codebooks) <= 256:       # (the codebook count is one byte, header.rs:1096; a draw with more books is drawn again)
            return st


def _random_setup_once(rng, channels, blocksizes, allow_floor0) -> StreamSetup:
    bs0, bs1 = blocksizes if blocksizes else _RANDOM_BS[int(rng.integers(0, len(_RANDOM_BS)))]
    if channels is None:
        channels = int(rng.choice([1, 2, 2, 2, 2, 3, 4, 5, 6, 6, 7, 8]))
    if channels * (1 << bs1) >= 65536:                     # audio.rs:745: blocksize * channels as u16 (type-2 residues)
        channels = 65535 // (1 << bs1)
    sample_rate = int(rng.choice([8000, 16000, 22050, 44100, 48000]))
    books: List[Codebook] = []
    vq_pool = []
    for _ in range(int(rng.integers(3, 9))):
        vq_pool.append(len(books))
        books.append(random_vq_book(rng))
    n_modes = int(rng.choice([1, 2, 2, 2, 2, 3, 3, 4, 6]))
    flags = [int(rng.integers(0, 2)) for _ in range(n_modes)]
    if n_modes >= 2 and rng.random() < 0.85:
        flags[0], flags[1] = (0, 1) if rng.random() < 0.7 else (1, 0)
    if bs0 == bs1 and rng.random() < 0.5:
        flags = [0] * n_modes                              # what libvorbis writes for equal block sizes
    floors, residues, mappings, modes = [], [], [], []
    floor_of_class = {0: [], 1: []}
    res_of_class = {0: [], 1: []}

    def new_floor(bf):
        bs = bs1 if bf else bs0
        if allow_floor0 and rng.random() < 0.06:
            fl = random_floor0(rng, books, bs, sample_rate)
        else:
            fl = random_floor1(rng, books, bs)
        floors.append(fl)
        floor_of_class[bf].append(len(floors) - 1)
        return len(floors) - 1

    def new_residue(bf, ch_sub):
        residues.append(random_residue(rng, books, vq_pool, bs1 if bf else bs0, max(1, ch_sub)))
        res_of_class[bf].append(len(residues) - 1)
        return len(residues) - 1

    map_of_class = {0: [], 1: []}
    for md in range(n_modes):
        bf = flags[md]
        if map_of_class[bf] and rng.random() < 0.45:       # several modes may share a mapping
            modes.append(Mode(bf, int(rng.choice(map_of_class[bf]))))
            continue
        nsub = int(rng.choice([1, 1, 1, 2, 2, 3]))
        nsub = min(nsub, 16)
        mux = [int(rng.integers(0, nsub)) for _ in range(channels)] if nsub > 1 else [0] * channels
        sf, sr = [], []
        for sm in range(nsub):
            ch_sub = sum(1 for c in mux if c == sm)
            reuse_f = floor_of_class[bf] and rng.random() < 0.4 and len(floors) < 12
            any_f = floors and rng.random() < 0.08        # (a floor made for the other block size: legal)
            sf.append(int(rng.choice(floor_of_class[bf])) if reuse_f else int(rng.integers(0, len(floors))) if any_f else new_floor(bf))
            reuse_r = res_of_class[bf] and rng.random() < 0.3
            any_r = residues and rng.random() < 0.08
            sr.append(int(rng.choice(res_of_class[bf])) if reuse_r else int(rng.integers(0, len(residues))) if any_r else new_residue(bf, ch_sub))
        mappings.append(Mapping(random_coupling(rng, channels), mux, sf, sr))
        map_of_class[bf].append(len(mappings) - 1)
        modes.append(Mode(bf, len(mappings) - 1))
    return StreamSetup(channels, sample_rate, bs0, bs1, books, floors, residues, mappings, modes)


class RandomPacketWriter(PacketWriter):
    """PacketWriter for arbitrary setups: symbols are drawn from the USED entries of whatever book the setup names."""

    def __init__(self, setup: StreamSetup, seed: int = 0, p_floor_unused: float = 0.05, p_zero_y: float = 0.6, loud: float = 1.0):
        super().__init__(setup, seed, p_floor_unused=p_floor_unused, p_zero_y=p_zero_y)
        self.loud = loud

    def _draw_y(self, book: int):
        cb = self.s.codebooks[book]
        used = cb.used_entries()
        if self.rng.random() < self.p_zero_y and cb.lengths[0] > 0:
            return 0
        if self.rng.random() < 0.9:
            small = [e for e in used if e <= 12]
            if small:
                return int(self.rng.choice(small))
        return int(self.rng.choice(used))

    def _floor1(self, w: BitWriter, fl: Floor1) -> bool:
        if self.rng.random() < self.p_unused:
            w.write(0, 1)
            return False
        w.write(1, 1)
        rng_ = [256, 128, 86, 64][fl.multiplier - 1]
        b = ilog(rng_ - 1)
        r = self.rng.random()
        for _ in range(2):
            if r < 0.9:
                v = int(np.clip(self.rng.integers(45, 96) * rng_ / 128.0 * self.loud, 0, rng_ - 1))
            else:
                v = int(self.rng.integers(0, 1 << b))       # anything the field can hold (86 .. 127 with range 86: clamped at the end)
            w.write(v, b)
        for c in fl.partition_class:
            cdim, cbits = fl.class_dim[c], fl.class_sub[c]
            cval = 0
            if cbits:
                mb = fl.class_master[c]
                cval = int(self.rng.choice(self.s.codebooks[mb].used_entries()))
                self._huff(w, mb, cval)
            for _ in range(cdim):
                book = fl.sub_books[c][cval & ((1 << cbits) - 1)]
                cval >>= cbits
                if book >= 0:
                    self._huff(w, book, self._draw_y(book))
        return True

    def _residue(self, w: BitWriter, rs: Residue, n: int, dnd: List[bool]):
        ch = len(dnd)
        if rs.type == 2:
            if all(dnd):
                return
            n = (n * ch) & 0xFFFF                           # audio.rs:745
            dnd = [False]
            ch = 1
        size = n // 2
        begin, end = min(rs.begin, size), min(rs.end, size)
        if end - begin == 0:
            return
        parts = (end - begin) // rs.partition_size
        classbook = self.s.codebooks[rs.classbook]
        cpc = classbook.dims
        ncls = rs.classifications
        cls = np.zeros((max(1, ch), parts + cpc), np.int64)
        for p_ in range(8):
            pc = 0
            while pc < parts:
                if p_ == 0:
                    for j in range(ch):
                        if dnd[j]:
                            continue
                        sym = int(self._draw_vq_entries(rs.classbook, 1)[0])
                        t = sym
                        for i in range(cpc - 1, -1, -1):
                            cls[j, pc + i] = t % ncls
                            t //= ncls
                        self._huff(w, rs.classbook, sym)
                for _ in range(cpc):
                    if pc >= parts:
                        break
                    for j in range(ch):
                        if dnd[j]:
                            continue
                        book = rs.books[int(cls[j, pc])][p_]
                        if book >= 0:
                            cb = self.s.codebooks[book]
                            offs = begin + pc * rs.partition_size
                            if rs.type == 0:
                                cnt = rs.partition_size // cb.dims
                            else:
                                cnt, i = 0, 0
                                while i < rs.partition_size:
                                    cnt += 1
                                    if i + cb.dims > size - offs:
                                        break
                                    i += cb.dims
                            for e in self._draw_vq_entries(book, cnt):
                                self._huff(w, book, int(e))
                    pc += 1


def random_stream(setup: StreamSetup, rng, count: int, seed: int = 0, p_floor_unused: float = 0.05, p_odd_flags: float = 0.03,
                  p_damage: float = 0.0) -> List[bytes]:
    """`count` audio packets of one stream of a random_setup(): runs of short and long blocks with window flags that fit the
    neighbours (sometimes not: legal), the mode drawn among the setup's modes of that block flag; `p_damage`: share of packets
    truncated or bit-flipped afterwards."""
    pw = RandomPacketWriter(setup, seed, p_floor_unused=p_floor_unused)
    by_flag = {0: [i for i, m in enumerate(setup.modes) if not m.blockflag], 1: [i for i, m in enumerate(setup.modes) if m.blockflag]}
    kinds = [f for f in (0, 1) if by_flag[f]]
    seq = []
    while len(seq) < count:
        f = int(rng.choice(kinds))
        seq += [f] * int(rng.choice([1, 1, 2, 3, 5, 8, 9, 17]))
    seq = seq[:count]
    out = []
    for i, f in enumerate(seq):
        mode = int(rng.choice(by_flag[f]))
        if not f:
            out.append(pw.packet(mode))
            continue
        pf = 1 if (i == 0 or seq[i - 1]) else 0
        nf = 1 if (i + 1 >= count or seq[i + 1]) else 0
        if rng.random() < p_odd_flags:
            pf, nf = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        out.append(pw.packet(mode, pf, nf))
    for i in range(count):
        r = rng.random()
        if r < p_damage / 2 and len(out[i]) > 2:
            out[i] = out[i][: int(rng.integers(1, len(out[i])))]
        elif r < p_damage and len(out[i]):
            p = bytearray(out[i])
            p[int(rng.integers(0, len(p)))] ^= 1 << int(rng.integers(0, 8))
            out[i] = bytes(p)
    return out
