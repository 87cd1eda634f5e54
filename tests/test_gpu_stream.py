"""The io::Read surface with read-ahead on the GPU (zgpu_streaming_*, zstd-rs_amd/csrc/zg_stream.h / zg_stream.cpp) against the oracle's
FrameDecoder driven by the reference's own read loop (StreamingDecoder::read, streaming_decoder.rs:119-155): for every sequence of read
sizes the same bytes, the same return value of every read() call, an error in the same call and with the same leaf, the same checksums.
Fixtures: the reference's decode corpus and dictionary fixtures, mutated corpus frames, hand-made frames whose offsets reach beyond the
window (what the reference does with those depends on what its caller has drained: the read-ahead must fall back to its schedule)."""
import hashlib
import io
import os
import random
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "zstd-rs_amd"))
import oracle
from golden_io import read_manifest, read_pack
from test_exact_cpu import K, frame, lit_block, raw_block, seq_block

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import zgpu
    c = zgpu.Context(0)
    yield c
    c.close()


def oracle_reads(z, reads, dict_raw=None):
    """ruzstd's StreamingDecoder::read over the oracle's FrameDecoder: [(n, bytes)] per call, ('err', status) ends the list"""
    o = oracle.FrameDecoder()
    if dict_raw is not None:
        o.add_dict(dict_raw)
    st, pos, _, _ = o.init(z)
    if st:
        return [("init", st)], o
    res = []
    for cap in reads:
        if o.is_finished() and o.can_collect() == 0:
            res.append((0, b""))
            continue
        err = 0
        while o.can_collect() < cap and not o.is_finished():
            st, used, fin = o.decode_blocks(z[pos:pos + (8 << 20) + cap], oracle.STRAT_UPTO_BYTES, cap - o.can_collect())
            pos += used
            if st:
                err = st
                break
        if err:
            res.append(("err", err))
            break
        d = o.read(cap)
        res.append((len(d), d))
    return res, o


def zgpu_reads(ctx, z, reads, callback, **kw):
    import zgpu
    try:
        s = zgpu.CStreamingDecoder(ctx, io.BytesIO(z), **kw) if callback else zgpu.CStreamingDecoder(ctx, data=z, **kw)
    except zgpu.ZgpuError as e:
        return [("init", e.status)], None
    res = []
    for cap in reads:
        try:
            d = s.read(cap)
        except zgpu.ZgpuError as e:
            res.append(("err", e.status))
            break
        res.append((len(d), d))
    return res, s


MODES = [dict(), dict(read_ahead=1), dict(pipe_after=1, read_ahead=4 << 20), dict(pipe_after=1 << 40, first_run_blocks=1)]


def patterns(rng, total):
    yield [total + 100, 10]
    yield [8192] * (total // 8192 + 3)
    r, done = [], 0
    while done < total + 1000:
        c = rng.choice([0, 1, 7, 4096, 8192, 65536, K, K + 1, 1 << 20, 3 << 20])
        r.append(c)
        done += c
    yield r + [5, 5]


def same(want, got, what):
    assert len(got) == len(want), (what, len(got), len(want), got[-1][:1], want[-1][:1])
    for i, (w, g) in enumerate(zip(want, got)):
        assert g[0] == w[0], (what, i, g[0], w[0])
        assert g[1] == w[1], (what, i, "bytes / status differ", g[1] if isinstance(g[1], int) else len(g[1]), w[1] if isinstance(w[1], int) else len(w[1]))


def test_corpus_through_every_mode(ctx):
    pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
    rng = random.Random(61)
    names = sorted(man)
    for k, name in enumerate(names):
        z = pack[name]
        size = man[name]["size"]
        pats = list(patterns(rng, size))
        pat = pats[k % 3]
        want, o = oracle_reads(z, pat)
        assert want[-1] == (0, b"") or size == 0
        for mi, kw in enumerate(MODES):
            if k % 4 != mi and k % 9:           # every frame through one mode, every ninth through all
                continue
            for callback in (False, True):
                got, s = zgpu_reads(ctx, z + b"bytes of the next frame", pat, callback, **kw)
                same(want, got, (name, kw, callback))
                assert s.is_finished() and s.get_calculated_checksum() == o.calculated_checksum() == s.get_checksum_from_data()
                assert s.blocks_decoded() == o.blocks_decoded() and s.bytes_read_from_source() == o.bytes_read_from_source() == len(z)
                if not callback:
                    assert s.source_position() == len(z)             # nothing behind the frame is taken from the source
                if kw.get("read_ahead") == 1:
                    assert s.stats()["mode"] == 2 and s.stats()["runs"] == 0
                s.close()


def test_dictionary_frames(ctx):
    pack, man = read_pack("dict_tests.pack"), read_manifest("dict_tests.json")
    raw = pack["dictionary"]
    ctx.add_dict(raw)
    rng = random.Random(62)
    names = sorted(n for n in man if n.endswith(".zst"))[::6]
    for k, name in enumerate(names):
        z = pack[name]
        pat = list(patterns(rng, man[name]["size"]))[k % 3]
        want, o = oracle_reads(z, pat, raw)
        for kw in (MODES[k % 4], MODES[(k + 1) % 4]):
            got, s = zgpu_reads(ctx, z, pat, k % 2 == 0, **kw)
            same(want, got, (name, kw))
            assert s.get_calculated_checksum() == o.calculated_checksum()
            s.close()


def test_mutated_frames_fail_in_the_same_read_call(ctx):
    """one to three bit flips per frame: the same bytes in front of the defect, the error in the same read() call, the same leaf —
    whether the defect is met block by block or inside a run that was decoded ahead (and then dropped)"""
    pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
    rng = random.Random(63)
    names = sorted(man)
    nerr = nok = 0
    leaves = set()
    for it in range(140):
        name = names[rng.randrange(len(names))]
        m = bytearray(pack[name])
        for _ in range(rng.choice([1, 1, 2, 3])):
            i = rng.randrange(4, len(m))
            m[i] ^= 1 << rng.randrange(8)
        z = bytes(m)
        pat = list(patterns(rng, man[name]["size"]))[it % 3]
        want, o = oracle_reads(z, pat)
        if want[-1][0] in ("err", "init"):
            nerr += 1
            leaves.add(want[-1][1])
        else:
            nok += 1
        for kw in (MODES[it % 4], MODES[0]):
            got, s = zgpu_reads(ctx, z, pat, it % 2 == 1, **kw)
            same(want, got, (name, it, kw))
            if s:
                s.close()
    assert nerr > 40 and nok > 10 and len(leaves) >= 5, (nerr, nok, leaves)


def test_offsets_beyond_the_window_follow_the_readers_drains(ctx):
    """A match that reaches beyond the window is served by the reference as long as its caller has not drained the bytes (decode_buffer.rs:
    79-111): with 8 KiB reads they are gone, with one large read they are still there. A run decoded ahead cannot know — it is dropped, and
    the block-by-block schedule decides like the oracle for both readers."""
    import zgpu
    blocks = [raw_block(K, i) for i in range(12)] + [seq_block(6 * K)] + [raw_block(K, 40 + i) for i in range(6)] + [lit_block(100, last=True)]
    z = frame(*blocks)
    seen = set()
    for pat in ([8192] * 400, [20 * K] * 3, [K] * 30, [3 * K, 9 * K, 9 * K, 9 * K]):
        want, o = oracle_reads(z, pat)
        seen.add(want[-1][0])
        for kw in MODES:
            for callback in (False, True):
                got, s = zgpu_reads(ctx, z, pat, callback, **kw)
                same(want, got, (pat[0], kw, callback))
                if kw.get("read_ahead") != 1:
                    assert s.stats()["dropped"] >= 1 and s.stats()["mode"] == 2
                s.close()
    assert seen == {"err", 0}, seen


def test_long_frame_goes_through_the_worker_thread(ctx):
    """192 MiB of text, 8 KiB reads (std::io::copy's buffer): the worker thread decodes runs ahead into the ring; the device window and the
    host ring stay bounded; plaintext, checksum and counters as the generator's / the frame's"""
    import zgdata
    import zgpu
    n = 192 << 20
    data = zgdata.text_like(n, seed=0x57A)
    z = zgdata.zstd_compress(data)
    want = hashlib.sha256(data).hexdigest()
    for kw, cap in ((dict(read_ahead=64 << 20), 8192), (dict(), 1 << 20), (dict(read_ahead=48 << 20, checksum=False), 40 << 20)):
        for callback in (True, False):
            s = zgpu.CStreamingDecoder(ctx, io.BytesIO(z + b"tail") if callback else None, data=None if callback else z + b"tail", **kw)
            h = hashlib.sha256()
            total = 0
            peak = 0
            while True:
                d = s.read(cap)
                if not d:
                    break
                assert len(d) == cap or total + len(d) == n
                h.update(d)
                total += len(d)
                if total % (32 << 20) < cap:
                    peak = max(peak, s.device_bytes())
            st = s.stats()
            assert total == n and h.hexdigest() == want, (kw, cap, callback)
            assert st["mode"] == 1 and st["runs"] >= 2 and st["dropped"] == 0, st
            assert s.is_finished() and s.get_checksum_from_data() is not None
            if kw.get("checksum", True):
                assert s.get_calculated_checksum() == s.get_checksum_from_data()
            assert st["host_bytes"] <= kw.get("read_ahead", 512 << 20) + (4 << 20)
            assert peak < (3 * kw.get("read_ahead", 512 << 20)) // 2 + (64 << 20), peak      # the window + two runs, not the frame
            if not callback:
                assert s.source_position() == len(z)
            s.close()


def test_io_copy_and_the_frame_decoders_read_ahead(ctx):
    import zgdata
    import zgpu
    data = zgdata.text_like(40 << 20, seed=0x10C)
    z = zgdata.zstd_compress(data)
    for bs in (8192, 1 << 20):
        s = zgpu.CStreamingDecoder(ctx, data=z)
        assert s.copy_to_sink(bs) == len(data)
        assert s.is_finished() and s.get_calculated_checksum() == s.get_checksum_from_data()
        s.close()
    # FrameDecoder::decode_blocks(UptoBytes(n)) with a read-ahead of 8 MiB: what UptoBytes(8 MiB) gives, asked for in 8 KiB pieces
    d = zgpu.FrameDecoder(ctx)
    d.set_read_ahead(8 << 20)
    st, c, _, _ = d.reset(z)
    assert st == 0
    pos, out, calls = c, [], 0
    while not d.is_finished():
        st, used, fin = d.decode_blocks(z[pos:pos + (12 << 20)], zgpu.STRAT_UPTO_BYTES, 8192)
        assert st == 0
        pos += used
        calls += 1
        out.append(d.collect())
    out.append(d.collect())
    assert b"".join(out) == data and calls <= 8, calls
    d.close()


def test_large_windows_and_unknown_sizes(ctx):
    """frames whose window is larger than the read-ahead budget (the ring then holds the window + two short runs), a frame without
    Frame_Content_Size (the stream starts on the caller's thread and hands over to the worker after 32 MiB), both against the generator's
    plaintext and the stored checksum; the repetitive text makes matches reach far back into the 32 MiB window"""
    import zgdata
    import zgpu
    rng = random.Random(64)
    base = zgdata.text_like(6 << 20, seed=0x71)
    data = b"".join(base[rng.randrange(0, 5 << 20):][: 1 << 20] for _ in range(72))          # 72 MiB: pieces of 6 MiB of text, repeated
    want = hashlib.sha256(data).hexdigest()
    for z, kw, cap in ((zgdata.zstd_compress(data, window_log=25), dict(read_ahead=16 << 20), 1 << 20),
                       (zgdata.zstd_compress(data, window_log=25), dict(), 8192),
                       (zgdata.zstd_compress(data, content_size=False), dict(read_ahead=24 << 20), 3 << 20),
                       (zgdata.zstd_compress(data, window_log=25, content_size=False), dict(read_ahead=1), 5 << 20)):
        for callback in (False, True):
            s = zgpu.CStreamingDecoder(ctx, io.BytesIO(z) if callback else None, data=None if callback else z, **kw)
            h = hashlib.sha256()
            total = 0
            while True:
                d = s.read(cap)
                if not d:
                    break
                assert len(d) == cap or total + len(d) == len(data)
                h.update(d)
                total += len(d)
            st = s.stats()
            assert total == len(data) and h.hexdigest() == want, (kw, cap, callback, st)
            assert s.is_finished() and s.get_calculated_checksum() == s.get_checksum_from_data()
            # (the frame without a declared size is 576 blocks: runs of 8, 32, 128 and then 512 blocks finish it on the caller's thread)
            assert st["dropped"] == 0 and st["mode"] == (2 if kw.get("read_ahead") == 1 else 1 if z[4] >> 6 else 0), st
            s.close()


def test_decoder_behind_the_stream(ctx):
    """get_ref() / get_mut() of the streaming decoder (streaming_decoder.rs:66-85): the FrameDecoder behind it hands out what the stream has
    buffered (can_collect / collect / read), and refuses to be fed by anyone else"""
    import ctypes as C
    import zgdata
    import zgpu
    data = zgdata.text_like(12 << 20, seed=0x6E7)
    z = zgdata.zstd_compress(data)
    for kw in (dict(), dict(pipe_after=1, read_ahead=6 << 20), dict(read_ahead=1)):
        s = zgpu.CStreamingDecoder(ctx, data=z, **kw)
        L, d = s.L, s._dec()
        got = s.read(1 << 20)
        assert got == data[:1 << 20]
        n = L.zgpu_decoder_can_collect(d)
        buf = C.create_string_buffer(max(n, 1))
        assert L.zgpu_decoder_collect(d, buf, n) == n
        got += buf.raw[:n]
        small = C.create_string_buffer(100)
        k = L.zgpu_decoder_read(d, small, 100)
        got += small.raw[:k]
        assert got == data[:len(got)]
        used, fin = C.c_size_t(), C.c_int()
        assert L.zgpu_decoder_decode_blocks(d, z, len(z), C.byref(used), zgpu.STRAT_ALL, 0, C.byref(fin)) == 93      # ZGPU_E_BAD_ARG
        while True:
            c = s.read(3 << 20)
            if not c:
                break
            got += c
        assert got == data and s.get_calculated_checksum() == s.get_checksum_from_data()
        s.close()
    ctx.L.zgpu_release_caches()          # the worker engine and the pinned ring the streams above left behind
    s = zgpu.CStreamingDecoder(ctx, data=z, pipe_after=1, read_ahead=6 << 20)
    assert s.read(len(data) + 1) == data
    s.close()


def test_fuzz_artifacts_through_the_stream(ctx):
    """the reference's fuzz artefacts (fuzz_regressions.rs:2-27: must not crash) through every mode of the streaming decoder: what the
    oracle's read loop does with them — bytes, the failing read() and its error, or an error when the header is read — is what comes back"""
    pack = read_pack("fuzz_artifacts.pack")
    rng = random.Random(65)
    n = 0
    for name, data in sorted(pack.items()):
        if not (name.startswith("decode/") or name.startswith("interop/")):
            continue
        pat = [rng.choice([1, 100, 8192, K, 1 << 20]) for _ in range(60)]
        want, o = oracle_reads(data, pat)
        for kw in MODES:
            got, s = zgpu_reads(ctx, data, pat, n % 2 == 0, **kw)
            same(want, got, (name, kw))
            if s:
                s.close()
        n += 1
    assert n >= 42


def test_streams_interleaved_on_one_context(ctx):
    """several streams of one context read in turn (every mode, one of them behind an io::Read callback), with decode_all calls of the same
    context in between: each stream keeps its own device window / worker engine, and the context's engine serves whoever calls"""
    import zgdata
    import zgpu
    rng = random.Random(0x171)
    frames = []
    for i, n in enumerate((5 << 20, 9 << 20, 3 << 20, 7 << 20)):
        data = zgdata.text_like(n + 12345 * i, seed=0x900 + i)
        frames.append((data, zgdata.zstd_compress(data)))
    kws = [dict(), dict(pipe_after=1, read_ahead=4 << 20), dict(read_ahead=1), dict(pipe_after=2 << 20, read_ahead=5 << 20, first_run_blocks=2)]
    streams = []
    for i, ((data, z), kw) in enumerate(zip(frames, kws)):
        s = zgpu.CStreamingDecoder(ctx, io.BytesIO(z), **kw) if i == 1 else zgpu.CStreamingDecoder(ctx, data=z, **kw)
        streams.append([s, data, 0])
    other = zgdata.text_like(1 << 20, seed=0x9FF)
    oz = zgdata.zstd_compress(other)
    live = list(range(len(streams)))
    step = 0
    while live:
        i = rng.choice(live)
        s, data, pos = streams[i]
        cap = rng.choice([1, 4096, 8192, K, 300000, 1 << 20, 2 << 20])
        d = s.read(cap)
        assert d == data[pos:pos + cap], (i, pos, cap, len(d))
        streams[i][2] = pos + len(d)
        if not d:
            assert pos == len(data) and s.is_finished() and s.get_calculated_checksum() == s.get_checksum_from_data()
            live.remove(i)
        step += 1
        if step % 7 == 0:
            assert ctx.decode_all(oz, len(other)) == other
    modes = [s.stats()["mode"] for s, _, _ in streams]
    assert modes == [0, 1, 2, 1], modes
    for s, _, _ in streams:
        s.close()
