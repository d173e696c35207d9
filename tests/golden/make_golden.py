#!/usr/bin/env python3
"""Extract the reference's own known-answer vectors for the audio-packet path.

Reads the `#[test]` bodies and test arrays in /root/reference/src (read-only) and writes
tests/golden/reference_vectors.json.  Only DATA is extracted (numbers asserted by the
reference's tests); no reference code is copied.  Run in the build container (the reference
tree does not exist on the GPU box); the JSON it produces is committed.

Sources:
  src/imdct_test.rs:11-980      IMDCT input/output arrays 1..3
  src/header_cached.rs:113-127  bit-reverse table for bs=8
  src/audio.rs:295-389          low/high neighbour and render_point answers
  src/lib.rs:164-172            ilog
  src/header.rs:262-276,651-669 ident header bytes + fields, lookup1_values
  src/bitpacking.rs:316-356,488-589  float32_unpack, bit reader vectors
  src/huffman_tree.rs:396-486   Huffman trees (valid / invalid, codeword paths)
"""
import json
import os
import re
import sys

REF = os.environ.get("LEWTON_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def read(rel):
    with open(os.path.join(REF, rel), "r", encoding="utf-8") as f:
        return f.read()


def fn_body(src, name):
    """Text of `fn name(...) { ... }` (brace matched)."""
    m = re.search(r"fn\s+" + re.escape(name) + r"\s*\(", src)
    assert m, name
    i = src.index("{", m.end())
    depth, j = 0, i
    while True:
        if src[j] == "{":
            depth += 1
        elif src[j] == "}":
            depth -= 1
            if depth == 0:
                return src[i + 1:j]
        j += 1


def strip_comments(s):
    s = re.sub(r"/\*.*?\*/", "", s, flags=re.S)
    return re.sub(r"//[^\n]*", "", s)


def ints(s):
    out = []
    for tok in re.findall(r"0x[0-9a-fA-F]+|0b[01]+|-?\d+", s):
        if tok.startswith("0x"):
            out.append(int(tok, 16))
        elif tok.startswith("0b"):
            out.append(int(tok, 2))
        else:
            out.append(int(tok))
    return out


def main():
    g = {}

    # ---- imdct_test.rs
    src = read("src/imdct_test.rs")
    imdct = {}
    for m in re.finditer(r"pub static (IMDCT_(?:INPUT|OUTPUT)_TEST_ARR_\d) :\[f32; (\d+)\] =\s*\[(.*?)\];", src, flags=re.S):
        name, n, body = m.group(1), int(m.group(2)), m.group(3)
        vals = [float(x) for x in re.findall(r"-?\d+\.\d+(?:e-?\d+)?", body)]
        assert len(vals) == n, (name, len(vals), n)
        imdct[name] = vals
    assert len(imdct) == 6
    g["imdct"] = imdct
    g["imdct_tolerance"] = 0.00005  # src/imdct.rs:833-846 fuzzy_compare_array epsilon, 0 mismatches allowed

    # ---- header_cached.rs: compute_bitreverse(8)
    src = read("src/header_cached.rs")
    body = strip_comments(fn_body(src, "test_compute_bitreverse"))
    arr = re.search(r"cmp_arr = &\[(.*?)\];", body, flags=re.S).group(1)
    g["bitreverse_bs8"] = ints(arr)
    assert len(g["bitreverse_bs8"]) == 32

    # ---- audio.rs
    src = read("src/audio.rs")
    rp = []
    for m in re.finditer(r"assert_eq!\(render_point\(([^)]*)\),\s*(\d+)\);", fn_body(src, "test_render_point")):
        rp.append({"args": ints(m.group(1)), "want": int(m.group(2))})
    assert len(rp) == 17
    g["render_point"] = rp
    nb = []
    for fn in ("test_low_neighbor", "test_high_neighbor", "test_high_neighbor_ex"):
        body = strip_comments(fn_body(src, fn))
        v = ints(re.search(r"let v = \[(.*?)\];", body, flags=re.S).group(1))
        for m in re.finditer(r"assert_eq!\((low|high)_neighbor\(&v, (\d+)\), \((\d+), (\d+)\)\);", body):
            nb.append({"kind": m.group(1), "v": v, "x": int(m.group(2)), "idx": int(m.group(3)), "val": int(m.group(4))})
    for fn, kind in (("test_high_neighbor_panic", "high"), ("test_low_neighbor_panic", "low")):
        m = re.search(r"(low|high)_neighbor\(&\[(.*?)\], (\d+)\)", fn_body(src, fn))
        nb.append({"kind": m.group(1), "v": ints(m.group(2)), "x": int(m.group(3)), "panics": True})
    g["neighbors"] = nb
    assert len(nb) == 5 + 3 + 17 + 2

    # ---- lib.rs: ilog
    src = read("src/lib.rs")
    g["ilog"] = [[int(a), int(b)] for a, b in re.findall(r"assert_eq!\(ilog\((\d+)\), (\d+)\);", fn_body(src, "test_ilog"))]
    assert len(g["ilog"]) == 6

    # ---- header.rs
    src = read("src/header.rs")
    body = strip_comments(fn_body(src, "test_read_header_ident"))
    pkt = ints(re.search(r"let test_arr = &\[(.*?)\];", body, flags=re.S).group(1))
    fields = {}
    for m in re.finditer(r"assert_eq!\(hdr\.(\w+), (0x[0-9a-fA-F]+|\d+)\);", body):
        fields[m.group(1)] = int(m.group(2), 0)
    g["ident_header"] = {"packet": pkt, "fields": fields}
    assert len(pkt) == 30 and len(fields) == 7
    body = strip_comments(fn_body(src, "test_read_hdr_begin"))
    g["ident_header_bad_capture"] = ints(re.search(r"let test_arr = &\[(.*?)\];", body, flags=re.S).group(1))
    l1 = []
    for m in re.finditer(r"assert_eq!\(lookup1_values\((\d+), (\d+)\), (std::u32::MAX|\d+)\);", fn_body(src, "test_lookup1_values")):
        want = 0xFFFFFFFF if "MAX" in m.group(3) else int(m.group(3))
        l1.append([int(m.group(1)), int(m.group(2)), want])
    g["lookup1_values"] = l1
    assert len(l1) == 11

    # ---- bitpacking.rs
    src = read("src/bitpacking.rs")
    fu = []
    for fn in ("test_float_32_unpack", "test_float_32_unpack_issue_24"):
        for m in re.finditer(r"assert_eq!\(float32_unpack\((\d+)\),\s*(-?[\d.]+)\);", fn_body(src, fn)):
            fu.append([int(m.group(1)), float(m.group(2))])
    g["float32_unpack"] = fu
    assert len(fu) == 27
    # bit reader: each case = bytes + a sequence of (width, expected value); widths from the method names
    width_of = {"read_u1": 1, "read_u2": 2, "read_u3": 3, "read_u4": 4, "read_u5": 5, "read_u6": 6, "read_u7": 7,
                "read_u8": 8, "read_u13": 13, "read_u16": 16, "read_u24": 24, "read_u32": 32, "read_i8": 8}
    cases = []
    for fn in ("test_bitpacking_reader_static", "test_bitpacking_reader_dynamic", "test_bitpacking_reader_empty",
               "test_bitpacking_reader_byte_aligned", "test_capture_pattern_nonaligned"):
        body = strip_comments(fn_body(src, fn))
        # split at each new array
        parts = re.split(r"let (?:test_arr|capture_pattern_arr) = &\[", body)[1:]
        for part in parts:
            data = ints(part[:part.index("]")])
            reads = []
            for m in re.finditer(r"(assert_eq!\()?cur\.(read_\w+)\((\d*)\)\.unwrap\(\)(?:,\s*(0x[0-9a-fA-F]+|\d+)\))?", part):
                meth, arg, want = m.group(2), m.group(3), m.group(4)
                width = int(arg) if meth.startswith("read_dyn") else width_of[meth]
                reads.append([width, None if want is None else int(want, 0)])
            cases.append({"test": fn, "data": data, "reads": reads})
    g["bitreader"] = cases
    assert len(cases) == 8, len(cases)

    # ---- huffman_tree.rs
    src = read("src/huffman_tree.rs")
    hf = []
    for fn in ("test_huffman_tree", "test_issue_8", "test_under_over_spec", "test_single_entry_huffman_tree",
               "test_unordered_huffman_tree", "test_extracted_huffman_tree"):
        body = strip_comments(fn_body(src, fn))
        # statements: load_from_array(&[...]) optionally followed by .unwrap() / is_err assertion, then iter_test lines
        pos = 0
        for m in re.finditer(r"load_from_array\(&\[(.*?)\]\)(\.unwrap\(\))?;", body, flags=re.S):
            raw = m.group(1)
            rep = re.match(r"\s*(\d+)\s*;\s*(\d+)\s*$", raw)
            lengths = [int(rep.group(1))] * int(rep.group(2)) if rep else ints(raw)
            tail = body[m.end():]
            nxt = re.search(r"load_from_array", tail)
            seg = tail[:nxt.start()] if nxt else tail
            entry = {"test": fn, "lengths": lengths}
            if m.group(2):
                entry["valid"] = True
            elif "is_err" in seg.split(";")[0] + seg[:40]:
                entry["valid"] = False
            else:
                entry["valid"] = None  # only "must not panic" (test_issue_8)
            entry["codewords"] = [[int(p, 2), int(l), int(v)] for p, l, v in
                                  re.findall(r"iter_test\(0b([01]+), (\d+), (\d+)\);", seg)]
            hf.append(entry)
    g["huffman"] = hf
    assert len(hf) == 2 + 1 + 3 + 3 + 1 + 1, len(hf)

    out = os.path.join(HERE, "reference_vectors.json")
    with open(out, "w") as f:
        json.dump(g, f, indent=0, separators=(",", ":"))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    sys.exit(main())
