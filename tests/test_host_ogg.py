"""Stream layer (lw_ogg.cpp = OggStreamReader of inside_ogg.rs:66-313) in the CPU suite: tests/san/ogg_stream_host.cpp links
the PRODUCT sources against stand-ins for the HIP runtime (tests/san/hip_standins.inc; sample values are zero, everything the
host decides is real) and prints a trace of sample counts, serials, chain links and granule positions.  The trace is compared
with the oracle's OggStreamReader (oracle/pyogg.py); the harness is built with ASan + UBSan, and mutated files are run
through it as well.  The sample VALUES of the same calls are checked on the GPU (tests/test_gpu_ogg.py)."""
import os
import subprocess

import numpy as np
import pytest

from common import ROOT
from oracle import pyogg
from test_ogg import GOLDEN, _vorbis_stream

CS = os.path.join(ROOT, "lewton_amd", "csrc")
SRC = [os.path.join(ROOT, "tests", "san", "ogg_stream_host.cpp")] + [
    os.path.join(CS, n) for n in ("lw_ogg.cpp", "lw_ring.cpp", "lw_runtime.cpp", "lw_batch.cpp", "lw_packet.cpp", "lw_pool.cpp", "lw_dev_entropy.cpp", "lw_entropy.cpp", "lw_headers.cpp", "lw_fast.cpp")]
HIP_INC = "/opt/rocm/include"


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if not os.path.isdir(os.path.join(HIP_INC, "hip")):
        pytest.skip("HIP headers not installed")
    exe = str(tmp_path_factory.mktemp("hostogg") / "ogg_stream_host")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I" + HIP_INC] + SRC + ["-lpthread", "-o", exe])
    return exe


def _chained():
    parts = []
    for k, (name, pattern, count, trim) in enumerate([("stereo", "LSL", 9, 11), ("surround51", "L", 6, 0),
                                                      ("mono_small", "SLL", 7, 3)]):
        parts.append(_vorbis_stream(name, pattern, count, seed=20 + k, serial=0x100 + k, per_page=2, trim=trim)[2].bytes())
    return b"".join(parts)


def _files():
    return {
        "golden": open(GOLDEN, "rb").read(),
        "trim": _vorbis_stream("stereo", "LSSL", 24, per_page=5, trim=700)[2].bytes(),
        "surround": _vorbis_stream("surround51", "LLSSSL", 18, per_page=4, trim=37)[2].bytes(),
        "mono_pages": _vorbis_stream("mono_small", "SL", 15, per_page=1, trim=5)[2].bytes(),
        "chained": _chained(),
    }


def _run(exe, tmp_path, data, *mode, dev_entropy=False, read_ahead=0, go_on=False):
    path = str(tmp_path / "in.ogg")
    with open(path, "wb") as f:
        f.write(data)
    env = dict(os.environ)
    if dev_entropy:
        env["LW_OSH_DEVICE_ENTROPY"] = "1"
    if read_ahead:
        env["LW_OSH_READ_AHEAD"] = str(read_ahead)   # lw_ogg_stream_set_read_ahead(s, K, 2) right after the open
    if go_on:
        env["LW_OSH_GO_ON"] = "1"                    # a drain goes on behind a BadAudio packet (trace line P 0 <code> ...)
    out = subprocess.run([exe, path] + [str(m) for m in mode], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-4000:]
    return [l.split() for l in out.stdout.splitlines()]


def _gp(v):
    return "-" if v is None else str(v)


def _oracle_trace(o):
    """P lines of a drain of the oracle reader + the terminating line."""
    rows, link, last_serial = [], 0, o.stream_serial
    while True:
        try:
            d = o.read_dec_packet()
        except pyogg.VorbisError as e:
            rows.append(["E", e])
            return rows
        if d is None:
            rows.append(["EOF"])
            return rows
        if o.stream_serial != last_serial:
            link, last_serial = link + 1, o.stream_serial
        rows.append(["P", str(d.shape[1]), "0", str(o.stream_serial), str(link), _gp(o.get_last_absgp())])


@pytest.mark.parametrize("name", ["golden", "trim", "surround", "mono_pages", "chained"])
def test_packet_by_packet_trace_matches_oracle(harness, tmp_path, name):
    data = _files()[name]
    got = _run(harness, tmp_path, data, "seq")
    want = _oracle_trace(pyogg.OggStreamReader(data))
    assert want[-1] == ["EOF"] and got == want
    if name == "chained":
        assert [r[4] for r in got if r[0] == "P"][-1] == "2"


@pytest.mark.parametrize("k", [1, 4, 64])
@pytest.mark.parametrize("name", ["golden", "trim", "chained"])
def test_look_ahead_queue_trace(harness, tmp_path, name, k):
    """read_dec_packets: the same sample counts in the same order, stops in front of a chain boundary, the granule
    position after every batch equals the packet-by-packet one at that point."""
    data = _files()[name]
    got = _run(harness, tmp_path, data, "ahead", k)
    assert _run(harness, tmp_path, data, "ahead", k, dev_entropy=True) == got
    want = [r for r in _oracle_trace(pyogg.OggStreamReader(data)) if r[0] == "P"]
    counts, at = [], []
    i = 0
    while i < len(got):
        r = got[i]
        if r[0] == "B":
            n = int(r[1])
            assert n <= k
            qs = got[i + 1:i + 1 + n]
            assert all(q[0] == "Q" and q[2] == "0" for q in qs)
            counts += [q[1] for q in qs]
            pos = got[i + 1 + n]
            if n == 0:                      # the single-packet call that crossed the chain boundary
                counts.append(pos[1])
            at.append((len(counts), pos[3], pos[5]))
            i += n + 2
        else:
            assert r == ["EOF"] and i == len(got) - 1
            i += 1
    assert counts == [w[1] for w in want]
    for n_done, serial, gp in at:
        assert (serial, gp) == (want[n_done - 1][3], want[n_done - 1][5])


@pytest.mark.parametrize("to_skip", [0, 1, 127, 128, 1500, 9000, 10 ** 7])
def test_skip_samples_linear_trace(harness, tmp_path, to_skip):
    data = _vorbis_stream("stereo", "LSSLL", 30, per_page=4, trim=100)[2].bytes()
    got = _run(harness, tmp_path, data, "skip", to_skip)
    o = pyogg.OggStreamReader(data)
    dec, left = o.skip_samples_linear(to_skip)
    want = [["S", "0" if dec is None else "1", "0" if dec is None else str(dec.shape[1]), str(left)]]
    if dec is not None:
        want.append(["P", str(dec.shape[1]), "0", str(o.stream_serial), "0", _gp(o.get_last_absgp())])
    assert got == want + _oracle_trace(o)


@pytest.mark.parametrize("k,singles,skip,goal", [(4, 2, 900, -1), (7, 0, 1, 5000), (3, 5, 3000, 0), (64, 1, 128, 20000),
                                                  (1, 1, 0, -1), (16, 3, 10 ** 7, -1)])
def test_calls_between_batched_calls_roll_the_pipeline_back(harness, tmp_path, k, singles, skip, goal):
    """read_dec_packet, skip_samples_linear and seek_absgp_pg between read_dec_packets calls (inside_ogg.rs:167-313): the
    look-ahead pipeline has read and entropy-decoded up to three batches ahead by then; every call must see the stream
    exactly where the packet-by-packet reader stands after the same calls (sample counts, granule positions, EOF)."""
    data = _vorbis_stream("stereo", "LLSLLLSSL", 70, per_page=4, trim=123)[2].bytes()
    got = _run(harness, tmp_path, data, "mix", k, singles, skip, goal)
    o = pyogg.OggStreamReader(data)
    want, stop = [], [False]

    def pos():
        return ["P", "0", "0", str(o.stream_serial), "0", _gp(o.get_last_absgp())]

    def batch():
        rows = []
        for _ in range(k):
            d = o.read_dec_packet()
            if d is None:
                break
            rows.append(["Q", str(d.shape[1]), "0"])
        if not rows:
            want.append(["EOF"])
            stop[0] = True
            return
        want.extend(rows)
        want.append(pos())

    batch()
    for _ in range(singles):
        if stop[0]:
            break
        d = o.read_dec_packet()
        if d is None:
            want.append(["EOF"])
            stop[0] = True
        else:
            want.append(["P", str(d.shape[1]), "0", str(o.stream_serial), "0", _gp(o.get_last_absgp())])
    if not stop[0]:
        batch()
    if not stop[0] and skip:
        dec, left = o.skip_samples_linear(skip)
        want.append(["S", "0" if dec is None else "1", "0" if dec is None else str(dec.shape[1]), str(left)])
        if dec is not None:
            want.append(["P", str(dec.shape[1]), "0", str(o.stream_serial), "0", _gp(o.get_last_absgp())])
    if not stop[0]:
        batch()
    if not stop[0] and goal >= 0:
        o.seek_absgp_pg(goal)
        want.append(["K", "0"])
        batch()
    if not stop[0]:
        want += _oracle_trace(o)
    assert got == want


@pytest.mark.parametrize("goal", [0, 3000, 12345, 10 ** 9])
def test_seek_absgp_pg_trace(harness, tmp_path, goal):
    data = _vorbis_stream("stereo", "LLSL", 40, per_page=3)[2].bytes()
    got = _run(harness, tmp_path, data, "seek", goal)
    o = pyogg.OggStreamReader(data)
    o.seek_absgp_pg(goal)
    assert got == [["K", "0"]] + _oracle_trace(o)


def test_mutated_files_under_sanitizers(harness, tmp_path):
    """Bit flips, truncations and duplicated pages of real and synthetic files: the stream layer may fail in any documented
    way, it may not trip ASan/UBSan; where the oracle reads the stream to a clean end the product must do so too, packet
    for packet."""
    rng = np.random.default_rng(8)
    files = _files()
    ran = agree = 0
    for name in ("golden", "trim", "chained"):
        base = files[name]
        for trial in range(14):
            d = bytearray(base)
            kind = trial % 4
            if kind == 0:
                for _ in range(int(rng.integers(1, 4))):
                    d[int(rng.integers(0, len(d)))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                d = d[: int(rng.integers(28, len(d)))]
            elif kind == 2:
                a = int(rng.integers(0, len(d) - 64))
                d[a:a + int(rng.integers(1, 64))] = b""
            else:
                a = bytes(d).find(b"OggS", int(rng.integers(0, len(d) // 2)))
                b = bytes(d).find(b"OggS", a + 4)
                if a >= 0 and b > a:
                    d[a:a] = d[a:b]          # a page twice
            for mode in (("seq",), ("ahead", 5)):
                got = _run(harness, tmp_path, bytes(d), *mode)
                ran += 1
                if mode[0] == "ahead":   # the same with the look-ahead batches in device-entropy mode (packets copied into the
                    #                      pinned pool instead of decoded): identical trace, nothing for the sanitizers
                    assert _run(harness, tmp_path, bytes(d), *mode, dev_entropy=True) == got
                if mode == ("seq",):
                    try:
                        want = _oracle_trace(pyogg.OggStreamReader(bytes(d)))
                    except pyogg.VorbisError:
                        assert got and got[0][0] == "E"
                        continue
                    if want[-1] == ["EOF"]:
                        assert got == want
                        agree += 1
                    else:
                        assert got[: len(want) - 1] == want[:-1] and got[len(want) - 1][0] == "E"
    assert ran == 84 and agree >= 5


def test_zero_length_packets_are_empty_packets_not_null_arguments(harness, tmp_path):
    """A zero-length Ogg packet (lacing value 0) is legal; lewton's reader fails on its first bit: BadAudio(EndOfPacket)
    for an audio packet (audio.rs:921), EndOfPacket for a header.  (The C ABI takes such a packet as (NULL, 0).)"""
    from common import SETUPS, sg
    from lewton_amd import ogg
    from oracle import pyoracle as po
    setup = SETUPS["stereo"]()
    idp, cmt, stp = setup.headers()
    pk = sg.make_stream(setup, "LSL", 9, seed=5)
    pk[4] = b""
    w = ogg.PageWriter(0x51)
    w.add_packet(idp, 0, flush=True)
    w.add_packet(cmt, 0)
    w.add_packet(stp, 0, flush=True)
    for i, p in enumerate(pk):
        w.add_packet(p, 1000 * i, flush=(i % 3 == 2), eos=(i == len(pk) - 1))
    data = w.bytes()
    want = _oracle_trace(pyogg.OggStreamReader(data))
    assert want[-1][0] == "E" and want[-1][1].kind == "BadAudio" and want[-1][1].inner == po.AUDIO_END_OF_PACKET and len(want) == 5
    got = _run(harness, tmp_path, data, "seq")
    assert got[:-1] == want[:-1] and got[-1] == ["E", str(po.AUDIO_END_OF_PACKET)]
    # the look-ahead queue reports it per packet
    got = _run(harness, tmp_path, data, "ahead", 16)
    qs = [r for r in got if r[0] == "Q"]
    assert [q[2] for q in qs][:5] == ["0"] * 4 + [str(po.AUDIO_END_OF_PACKET)]
    # an empty comment header: read_header_comment fails on its first byte (header.rs:309-313)
    w = ogg.PageWriter(0x52)
    w.add_packet(idp, 0, flush=True)
    w.add_packet(b"", 0)
    w.add_packet(stp, 0, flush=True)
    data = w.bytes()
    with pytest.raises(pyogg.VorbisError) as e:
        pyogg.OggStreamReader(data)
    assert (e.value.kind, e.value.inner) == ("BadHeader", po.HDR_END_OF_PACKET)
    got = _run(harness, tmp_path, data, "seq")
    assert got == [["E", str(po.HDR_END_OF_PACKET)]]


# ---- lw_ogg_stream_set_read_ahead: the packet-by-packet call served from batches decoded ahead.  Nothing the caller can observe
#      may change: every trace above, with the read-ahead on, is the trace without it (= the oracle's).
@pytest.mark.parametrize("k", [1, 3, 64])
@pytest.mark.parametrize("name", ["golden", "trim", "surround", "mono_pages", "chained"])
def test_read_ahead_packet_by_packet_trace_matches_oracle(harness, tmp_path, name, k):
    data = _files()[name]
    got = _run(harness, tmp_path, data, "seq", read_ahead=k)
    want = _oracle_trace(pyogg.OggStreamReader(data))
    assert want[-1] == ["EOF"] and got == want          # sample counts, serials, links AND the granule position after every call
    assert _run(harness, tmp_path, data, "seq", read_ahead=k, dev_entropy=True) == got


@pytest.mark.parametrize("k", [2, 5, 64])
@pytest.mark.parametrize("first,skip,more,goal", [(3, 900, 4, -1), (1, 1, 0, 5000), (7, 3000, 2, 0), (10, 0, 0, 20000), (0, 128, 9, -1),
                                                   (4, 10 ** 7, 3, -1), (66, 5000, 1, 100)])
def test_read_ahead_other_calls_in_the_middle_of_a_served_batch(harness, tmp_path, k, first, skip, more, goal):
    """skip_samples_linear and seek_absgp_pg (and the end of the stream) while packets of a served batch wait to be handed out: those
    packets go back, the PreviousWindowRight is re-made as of the last packet handed out, and the stream stands exactly where the
    packet-by-packet reader stands after the same calls (inside_ogg.rs:167-313)."""
    data = _vorbis_stream("stereo", "LLSLLLSSL", 70, per_page=4, trim=123)[2].bytes()
    got = _run(harness, tmp_path, data, "hop", first, skip, more, goal, read_ahead=k)
    assert got == _run(harness, tmp_path, data, "hop", first, skip, more, goal)      # the sequential product path
    o = pyogg.OggStreamReader(data)
    want, stop = [], [False]

    def singles(cnt):
        for _ in range(cnt):
            if stop[0]:
                return
            d = o.read_dec_packet()
            if d is None:
                want.append(["EOF"])
                stop[0] = True
            else:
                want.append(["P", str(d.shape[1]), "0", str(o.stream_serial), "0", _gp(o.get_last_absgp())])

    singles(first)
    if not stop[0] and skip:
        dec, left = o.skip_samples_linear(skip)
        want.append(["S", "0" if dec is None else "1", "0" if dec is None else str(dec.shape[1]), str(left)])
        if dec is not None:
            want.append(["P", str(dec.shape[1]), "0", str(o.stream_serial), "0", _gp(o.get_last_absgp())])
    singles(more)
    if not stop[0] and goal >= 0:
        o.seek_absgp_pg(goal)
        want.append(["K", "0"])
    if not stop[0]:
        want += _oracle_trace(o)
    assert got == want


@pytest.mark.parametrize("k,singles,skip,goal", [(4, 2, 900, -1), (7, 0, 1, 5000), (3, 5, 3000, 0), (16, 3, 10 ** 7, -1)])
def test_read_ahead_mixed_with_the_batched_call(harness, tmp_path, k, singles, skip, goal):
    """the batched call and the served packet-by-packet call on one stream (a batched call continues behind the last packet handed
    out, whatever was read ahead): the `mix` trace with the read-ahead on is the one without"""
    data = _vorbis_stream("stereo", "LLSLLLSSL", 70, per_page=4, trim=123)[2].bytes()
    want = _run(harness, tmp_path, data, "mix", k, singles, skip, goal)
    for ra in (2, 9):
        assert _run(harness, tmp_path, data, "mix", k, singles, skip, goal, read_ahead=ra) == want


def test_read_ahead_mutated_files_under_sanitizers(harness, tmp_path):
    """the mutated files of the test above through the served packet-by-packet call: the same trace as the sequential call, error
    for error (audio errors at their packets, container errors behind the packets that precede them), nothing for ASan / UBSan"""
    rng = np.random.default_rng(9)
    files = _files()
    ran = 0
    for name in ("golden", "trim", "chained"):
        base = files[name]
        for trial in range(12):
            d = bytearray(base)
            kind = trial % 4
            if kind == 0:
                for _ in range(int(rng.integers(1, 4))):
                    d[int(rng.integers(0, len(d)))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                d = d[: int(rng.integers(28, len(d)))]
            elif kind == 2:
                a = int(rng.integers(0, len(d) - 64))
                d[a:a + int(rng.integers(1, 64))] = b""
            else:
                a = bytes(d).find(b"OggS", int(rng.integers(0, len(d) // 2)))
                b = bytes(d).find(b"OggS", a + 4)
                if a >= 0 and b > a:
                    d[a:a] = d[a:b]
            want = _run(harness, tmp_path, bytes(d), "seq")
            assert _run(harness, tmp_path, bytes(d), "seq", read_ahead=int(rng.choice([1, 4, 50]))) == want
            ran += 1
    assert ran == 36


def _damaged(seed, count=60, p_bad=0.15):
    """a stereo stream with damaged audio packets (cut, header bit set, flipped bits): several fail with an AudioReadError"""
    from common import sg
    from lewton_amd import ogg
    from oracle import pyoracle as po
    setup, pk, _ = _vorbis_stream("stereo", "LLSLLLSSL", count, seed=seed)
    rng = np.random.default_rng(seed)
    idp, cmt, stp = setup.headers()
    o_id = po.Ident(idp)
    o_st = po.Setup(stp, o_id)
    w = ogg.PageWriter(0x77)
    w.add_packet(idp, 0, flush=True)
    w.add_packet(cmt, 0)
    w.add_packet(stp, 0, flush=True)
    gp = 0
    for i, p in enumerate(pk):
        if i and i + 1 < len(pk) and rng.random() < p_bad:
            kind = int(rng.integers(0, 3))
            if kind == 0:
                p = p[: max(1, len(p) // int(rng.integers(2, 9)))]
            elif kind == 1:
                p = bytes([p[0] | 1]) + p[1:]
            else:
                q = bytearray(p)
                for _ in range(3):
                    q[int(rng.integers(0, len(q)))] ^= 1 << int(rng.integers(0, 8))
                p = bytes(q)
        try:
            gp += po.get_decoded_sample_count(o_id, o_st, p) if i else 0
        except po.OracleError:
            pass
        w.add_packet(p, gp, flush=(i % 5 == 4), eos=(i == len(pk) - 1))
    return w.bytes()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_read_ahead_damaged_packets_fail_at_their_call_and_the_stream_goes_on(harness, tmp_path, seed):
    """AudioReadErrors in the middle of served batches: the code at the call of its packet, no samples, the granule position
    untouched, the next call the next packet (sample counts as the oracle's reader gives them when it goes on behind an error)"""
    data = _damaged(seed)
    o = pyogg.OggStreamReader(data)
    want, n_bad = [], 0
    while True:
        try:
            d = o.read_dec_packet()
        except pyogg.VorbisError as e:
            assert e.kind == "BadAudio"
            want.append(["P", "0", str(e.inner), str(o.stream_serial), "0", _gp(o.get_last_absgp())])
            n_bad += 1
            continue
        if d is None:
            want.append(["EOF"])
            break
        want.append(["P", str(d.shape[1]), "0", str(o.stream_serial), "0", _gp(o.get_last_absgp())])
    assert n_bad >= 2
    assert _run(harness, tmp_path, data, "seq", go_on=True) == want
    for k in (1, 4, 25):
        assert _run(harness, tmp_path, data, "seq", go_on=True, read_ahead=k) == want


@pytest.fixture(scope="module")
def harness_tsan(tmp_path_factory):
    if not os.path.isdir(os.path.join(HIP_INC, "hip")):
        pytest.skip("HIP headers not installed")
    exe = str(tmp_path_factory.mktemp("hostogg_tsan") / "ogg_stream_host_tsan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__",
                           "-I" + HIP_INC] + SRC + ["-lpthread", "-o", exe])
    return exe


def test_read_ahead_and_look_ahead_under_thread_sanitizer(harness, harness_tsan, tmp_path):
    """the stream's two library threads (demultiplexer, entropy staging) against the caller's thread -- batches served packet by
    packet, roll-backs by skip / seek / the batched call in the middle of them, chain boundaries, damaged packets: the same traces,
    and no report from ThreadSanitizer"""
    mix = _vorbis_stream("stereo", "LLSLLLSSL", 70, per_page=4, trim=123)[2].bytes()
    cases = [(_damaged(2), ("seq",), dict(go_on=True)), (mix, ("hop", 5, 900, 4, 5000), {}), (mix, ("mix", 4, 2, 900, 3000), {}),
             (_files()["chained"], ("seq",), {}), (mix, ("ahead", 6), {})]
    for data, mode, kw in cases:
        for k in (1, 7):
            path = str(tmp_path / "in.ogg")
            with open(path, "wb") as f:
                f.write(data)
            env = dict(os.environ, LW_OSH_READ_AHEAD=str(k), TSAN_OPTIONS="halt_on_error=0")
            if kw.get("go_on"):
                env["LW_OSH_GO_ON"] = "1"
            out = subprocess.run([harness_tsan, path] + [str(m) for m in mode], capture_output=True, text=True, timeout=600, env=env)
            assert out.returncode == 0 and "ThreadSanitizer" not in out.stderr, out.stderr[-3000:]
            assert [l.split() for l in out.stdout.splitlines()] == _run(harness, tmp_path, data, *mode, read_ahead=k, **kw)
