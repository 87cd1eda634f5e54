"""GPU parity tests: the HIP path (through the C ABI) against the golden fixtures and the CPU oracle.

Bit-exact is the bar: all work on this path is integer/byte arithmetic. Mirrors the reference's own test strategy
(ruzstd/src/tests/decode_corpus.rs, tests/mod.rs, fuzz_regressions.rs)."""
import hashlib
import os
import sys

import pytest

import oracle
from golden_io import read_manifest, read_pack

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def ctx():
    import zgpu
    c = zgpu.Context(0)
    yield c
    c.close()


def _sha(b):
    return hashlib.sha256(b).hexdigest()


def test_corpus_decode_all(ctx):
    """decode_corpus.rs:92-132 through FrameDecoder::decode_all"""
    pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
    bad = []
    for name in sorted(man):
        out = ctx.decode_all(pack[name], man[name]["size"])
        if len(out) != man[name]["size"] or _sha(out) != man[name]["sha256"]:
            bad.append(name)
    assert not bad, bad


def test_corpus_one_batch(ctx):
    """all 101 frames in ONE submit (many frames in flight at once), each frame checked"""
    pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
    names = sorted(man)
    blob = b"".join(pack[n] for n in names)
    b = ctx.prepare(blob)
    assert b.parse_status == 0 and b.nframes == len(names)
    b.run()
    b.sync()
    assert b.bad_status == 0, (b.bad_frame, b.bad_status)
    bad = []
    for f, n in enumerate(names):
        out = b.frame_bytes(f)
        if _sha(out) != man[n]["sha256"]:
            bad.append(n)
    assert not bad, bad
    assert b.total_out == sum(man[n]["size"] for n in names)
    b.close()


def test_corpus_frame_decoder_surface(ctx):
    """reset + decode_blocks(All) + collect, counters and checksum (decode_corpus.rs:51-110)"""
    import zgpu
    pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
    d = zgpu.FrameDecoder(ctx)
    for name in sorted(man)[::7]:
        z = pack[name]
        st, c, _, _ = d.reset(z)
        assert st == 0
        st, used, fin = d.decode_blocks(z[c:], zgpu.STRAT_ALL)
        assert st == 0 and fin
        assert d.is_finished()
        out = d.collect()
        assert _sha(out) == man[name]["sha256"], name
        assert d.bytes_read_from_source() == len(z), name
        assert d.get_checksum_from_data() == d.get_calculated_checksum(), name
    d.close()


def _oracle_blocks(z):
    d = oracle.FrameDecoder()
    st, c, _, _ = d.init(z)
    assert st == 0
    pos, blocks = c, []
    while not d.is_finished():
        st, used, fin = d.decode_blocks(z[pos:], oracle.STRAT_UPTO_BLOCKS, 1)
        assert st == 0
        pos += used
        rec = {"type": d.last_block_type(), "hist_after": d.offset_hist()}
        if rec["type"] == 2:
            rec["literals"] = d.last_literals()
            rec["sequences"] = d.last_sequences()
            rec["huf"] = d.huf_table()
            rec["fse"] = [d.fse_table(k) for k in range(3)]
        blocks.append(rec)
        if fin:
            break
    return blocks


@pytest.mark.parametrize("name", ["z000000.zst", "z000033.zst", "z000059.zst", "z000088.zst"])
def test_kernel_intermediates_match_oracle(ctx, name):
    """per kernel: Huffman tables + literals (zg_k_tables, zg_k_huf), FSE tables + sequences (zg_k_tables, zg_k_seq),
    offset history at every block start (zg_k_scan) — against the oracle's intermediates"""
    z = read_pack("decodecorpus.pack")[name]
    ob = _oracle_blocks(z)
    b = ctx.prepare(z)
    b.run()
    b.sync()
    assert b.bad_status == 0
    assert b.nblocks == len(ob)
    hist = [1, 4, 8]
    for i, rec in enumerate(ob):
        info = b.block_info(i)
        assert info.btype == rec["type"] and info.status == 0
        assert list(info.hist_init) == hist, (name, i)
        hist = rec["hist_after"]
        if rec["type"] != 2:
            continue
        if info.lit_type >= 2:
            assert b.block_literals(i, info.regen_size) == rec["literals"], (name, i)
            tab, mb = b.huf_slot(info.huf_slot)
            oents, omb = rec["huf"]
            assert mb == omb and [(tab[k] & 255, tab[k] >> 8) for k in range(1 << mb)] == oents
        seqs = b.block_sequences(i, info.nseq)
        lit_pos = out_pos = 0
        h = list(info.hist_init)
        for (of, ml, mdst, lit_start), (oll, oml, _o, oactual) in zip(seqs, rec["sequences"]):
            tag, k = of >> 30, of & 0x3FFFFFFF
            actual = of if tag == 0 else max(h[tag - 1] - k, 0)
            assert (actual, ml, mdst, lit_start) == (oactual, oml, out_pos + oll, lit_pos), (name, i)
            lit_pos += oll
            out_pos += oll + oml
        if info.nseq:
            for k, slot in enumerate((info.ll_slot, info.of_slot, info.ml_slot)):
                oents, olog, orle = rec["fse"][k]
                p, logs = b.fse_slot(slot)
                off = (0, 1024, 512)[k]
                if orle >= 0:
                    assert logs[k] == 0 and ((p[off] >> 20) & 63) == orle
                else:
                    got = [(p[off + j] & 0xFFFF, (p[off + j] >> 16) & 15, (p[off + j] >> 20) & 63) for j in range(1 << olog)]
                    assert logs[k] == olog and got == oents, (name, i, k)
    b.close()


def test_packed_sequence_tables(monkeypatch):
    """zg_k_seq's packed form (16-bit table entries, three workgroups per CU: what a submit of more blocks than one round holds gets,
    round 5) forced onto small inputs: every sequence of corpus frames against the oracle's, and the verdicts of corrupted sequence
    sections (the careful form of the step, a stream that runs out of bits, bits left over) like the unpacked form's"""
    import random
    import zgpu
    monkeypatch.setenv("ZGPU_SEQ_PACKED", "1")
    c = zgpu.Context(0, dev=True)
    pack = read_pack("decodecorpus.pack")
    for name in ("z000000.zst", "z000033.zst", "z000059.zst", "z000088.zst", "z000012.zst"):
        test_kernel_intermediates_match_oracle(c, name)
    monkeypatch.setenv("ZGPU_SEQ_PACKED", "0")
    c0 = zgpu.Context(0, dev=True)
    rng = random.Random(55)
    nerr = 0
    for name in ("z000033.zst", "z000059.zst", "z000047.zst"):
        base = pack[name]
        for it in range(40):
            m = bytearray(base)
            i = rng.randrange(12, len(m)); m[i] ^= 1 << rng.randrange(8)
            m = bytes(m)
            ost, oout = oracle.FrameDecoder().decode_all(m, 1 << 24)
            res = []
            for cx in (c, c0):
                try:
                    res.append((0, cx.decode_all(m, 1 << 24)))
                except zgpu.ZgpuError as e:
                    res.append((e.status, None))
            assert res[0] == res[1] and res[0][0] == ost and (ost or res[0][1] == oout), (name, it, ost, res[0][0], res[1][0])
            nerr += 1 if ost else 0
    assert nerr > 30
    c.close(); c0.close()


def test_synthetic_fixtures(ctx):
    """real libzstd streams (committed fixtures): text L1/L3/L19, iso-like, mixed"""
    pack, man = read_pack("synthetic.pack"), read_manifest("synthetic.json")
    for name in sorted(man):
        out = ctx.decode_all(pack[name], man[name]["size"])
        assert _sha(out) == man[name]["sha256"], name


def test_window_fixtures(ctx):
    import zgpu
    pack, man = read_pack("test_fixtures.pack"), read_manifest("test_fixtures.json")
    for name in ("window_8mib.zst", "window_128mib.zst"):
        assert _sha(ctx.decode_all(pack[name], man[name]["size"])) == man[name]["sha256"]
    with pytest.raises(zgpu.ZgpuError) as e:                        # tests/mod.rs:615-637
        ctx.decode_all(pack["window_256mib.zst"], man["window_256mib.zst"]["size"])
    assert e.value.status == zgpu.E_WINDOW_SIZE_TOO_BIG
    ctx.set_max_window_size(300 << 20)
    try:
        two = pack["window_256mib.zst"] * 2                          # tests/mod.rs:639-665
        out = ctx.decode_all(two, 2 * man["window_256mib.zst"]["size"])
        assert _sha(out[: len(out) // 2]) == man["window_256mib.zst"]["sha256"] and out[: len(out) // 2] == out[len(out) // 2:]
    finally:
        ctx.set_max_window_size(128 << 20)


def test_decode_all_multiframe_skippable_and_errors(ctx):
    """tests/mod.rs:490-574"""
    import zgpu
    pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
    z = pack["z000088.zst"]
    n = man["z000088.zst"]["size"]
    skip = bytes([0x50, 0x2A, 0x4D, 0x18, 3, 0, 0, 0, 1, 2, 3])
    data = skip + z + skip + z + skip
    out = ctx.decode_all(data, 2 * n)
    assert len(out) == 2 * n and _sha(out[:n]) == man["z000088.zst"]["sha256"] and out[:n] == out[n:]
    with pytest.raises(zgpu.ZgpuError) as e:
        ctx.decode_all(data, 2 * n - 1)
    assert e.value.status == zgpu.E_TARGET_TOO_SMALL
    with pytest.raises(zgpu.ZgpuError) as e:
        ctx.decode_all(z[:-5], n)
    assert e.value.status in (zgpu.E_FAILED_READ_BLOCK_HEADER, zgpu.E_FAILED_READ_BLOCK_BODY, zgpu.E_FAILED_READ_CHECKSUM)
    with pytest.raises(zgpu.ZgpuError) as e:
        ctx.decode_all(skip[:-1], 16)
    assert e.value.status == zgpu.E_FAILED_SKIP_FRAME
    assert ctx.decode_all(b"", 16) == b""


def test_fuzz_artifacts_never_crash_and_agree(ctx):
    """fuzz_regressions.rs:2-27: must not crash; when the oracle decodes the input, the bytes must agree"""
    import zgpu
    pack = read_pack("fuzz_artifacts.pack")
    n = 0
    for name, data in pack.items():
        if not (name.startswith("decode/") or name.startswith("interop/")):
            continue
        ost, oout = oracle.FrameDecoder().decode_all(data, 1 << 24)
        try:
            out = ctx.decode_all(data, 1 << 24)
            ok = True
        except zgpu.ZgpuError:
            ok = False
        if ost == 0:
            assert ok and out == oout, name
        n += 1
    assert n >= 42


def test_generated_large_text_and_iso(ctx):
    """larger real-encoder inputs generated on the box (8 MiB text, 4 MiB iso, several levels and windows)"""
    try:
        import zgdata
        zgdata.libzstd()
    except Exception as e:  # pragma: no cover
        pytest.skip("libzstd not available to create inputs: %s" % e)
    for plain, lvl, wl in ((zgdata.text_like(8 << 20, seed=0x77), 3, 0), (zgdata.text_like(3 << 20, seed=0x78), 9, 0),
                           (zgdata.iso_like(4 << 20, seed=0x79), 3, 0), (zgdata.text_like(2 << 20, seed=0x7A), 3, 17),
                           (b"ab" * (1 << 20) + bytes(1 << 20), 3, 0)):
        z = zgdata.zstd_compress(plain, level=lvl, window_log=wl)
        out = ctx.decode_all(z, len(plain))
        assert out == plain, (lvl, wl)
        oout, _ = oracle.decode_frame_all(z)                     # and the oracle agrees on the same input
        assert oout == plain


def test_huffman_stream_shapes(ctx):
    """zg_k_huf decodes a stream 64 chunks at a time and relies on the code re-synchronising: stress it with code-length
    mixes from 1-bit codes (up to 128 symbols per 128-bit chunk) to flat 8-bit alphabets, stream lengths around the
    window size (8192 bits), one- and four-stream sections. Inputs come from the real encoder; outputs are compared with
    the oracle and the plaintext, block literals with the oracle's literals."""
    try:
        import zgdata
        zgdata.libzstd()
    except Exception as e:  # pragma: no cover
        pytest.skip("libzstd not available to create inputs: %s" % e)
    import numpy as np
    rng = np.random.default_rng(0xF00D)
    cases = []
    for n in (300, 1000, 1023, 1024, 1025, 4096, 8191, 8192, 8193, 70000, 131072, 300000):
        for probs in ((0.93, 0.04, 0.02, 0.01), (0.5, 0.25, 0.125, 0.0625, 0.0625), tuple([1 / 40.0] * 40), tuple([1 / 250.0] * 250)):
            syms = rng.permutation(256)[:len(probs)].astype(np.uint8)
            cases.append(bytes(rng.choice(syms, size=n, p=np.array(probs) / sum(probs))))
    checked = 0
    for plain in cases:
        z = zgdata.zstd_compress(plain, level=3)
        out = ctx.decode_all(z, len(plain))
        assert out == plain, (len(plain), checked)
        oout, _ = oracle.decode_frame_all(z)
        assert oout == plain
        checked += 1
    assert checked == len(cases)


def test_mutated_frames_agree_with_oracle(ctx):
    """Malformed input: 480 random mutations (bit flips, byte and pair overwrites, truncations) of four valid frames.
    The engine must decide like the oracle: same bytes when both decode, the same error leaf (errors.rs) when both fail,
    never one without the other — the two "offset too far" leaves included (which one applies depends on the reference's
    total_output_counter, which skips raw and RLE blocks, decode_buffer.rs:62-72: zg_k_exact replays it)."""
    import random
    import zgpu
    packs, syn = read_pack("decodecorpus.pack"), read_pack("synthetic.pack")
    bases = [packs["z000033.zst"], packs["z000059.zst"], syn["mixed_640k_l3.zst"], packs["z000000.zst"]]
    rng = random.Random(12345)
    same_ok = same_err = 0
    diffs = {}
    for bi, base in enumerate(bases):
        for it in range(120):
            m = bytearray(base)
            kind = rng.randrange(4)
            if kind == 0:
                i = rng.randrange(4, len(m)); m[i] ^= 1 << rng.randrange(8)
            elif kind == 1:
                i = rng.randrange(4, len(m)); m[i] = rng.randrange(256)
            elif kind == 2:
                m = m[:rng.randrange(8, len(m))]
            else:
                i = rng.randrange(4, len(m) - 4); m[i:i + 2] = bytes([rng.randrange(256), rng.randrange(256)])
            m = bytes(m)
            ost, oout = oracle.FrameDecoder().decode_all(m, 1 << 24)
            try:
                out, gst = ctx.decode_all(m, 1 << 24), 0
            except zgpu.ZgpuError as e:
                out, gst = None, e.status
            if ost == 0 and gst == 0:
                assert out == oout, (bi, it)
                same_ok += 1
            elif ost == gst:
                same_err += 1
            else:
                diffs[(ost, gst)] = diffs.get((ost, gst), 0) + 1
    assert not diffs, diffs
    assert same_ok > 50 and same_err > 200, (same_ok, same_err)


def test_twice_mutated_frames_first_error_in_stream_order(ctx):
    """Two defects in one input: the reference decodes block by block (decode_blocks, frame_decoder.rs:319-375), so a defect inside
    an early block is reported before a header the walk cannot read further back (a truncated body, a reserved block type, the next
    frame's magic) — although the engine's host walk meets the latter first. 360 inputs with two or three random mutations each
    (found by tools/dev/soak.py), plus the hand-made case: a corrupted block in front of a truncation."""
    import random
    import zgpu
    packs, syn = read_pack("decodecorpus.pack"), read_pack("synthetic.pack")
    bases = [packs["z000033.zst"], packs["z000059.zst"], syn["mixed_640k_l3.zst"], packs["z000000.zst"], packs["z000012.zst"], packs["z000047.zst"]]

    def both(m):
        ost, oout = oracle.FrameDecoder().decode_all(m, 1 << 24)
        try:
            out, gst = ctx.decode_all(m, 1 << 24), 0
        except zgpu.ZgpuError as e:
            out, gst = None, e.status
        return ost, oout, gst, out

    # hand-made: find a byte whose corruption makes a block fail, then cut the frame behind it
    base = syn["mixed_640k_l3.zst"]
    found = 0
    for i in range(40, len(base) // 3, 997):
        m = bytearray(base); m[i] ^= 0x5A
        ost, _, _, _ = both(bytes(m))
        if ost in (0, zgpu.E_FAILED_READ_BLOCK_BODY, zgpu.E_FAILED_READ_BLOCK_HEADER):
            continue
        cut = bytes(m[:len(m) - len(m) // 4])
        o2, _, g2, _ = both(cut)
        assert o2 == ost and g2 == o2, (i, ost, o2, g2)
        found += 1
        if found == 3:
            break
    assert found >= 1
    rng = random.Random(4242)
    diffs = {}
    n_walk_second = 0
    for bi, base in enumerate(bases):
        for it in range(60):
            m = bytearray(base)
            for _ in range(2 + rng.randrange(2)):
                if len(m) < 16:
                    break
                kind = rng.randrange(5)
                if kind == 0:
                    i = rng.randrange(4, len(m)); m[i] ^= 1 << rng.randrange(8)
                elif kind == 1:
                    i = rng.randrange(4, len(m)); m[i] = rng.randrange(256)
                elif kind == 2:
                    m = m[:rng.randrange(8, len(m))]
                elif kind == 3:
                    i = rng.randrange(4, len(m) - 4); m[i:i + 2] = bytes([rng.randrange(256), rng.randrange(256)])
                else:
                    i = rng.randrange(4, min(len(m), 40)); m[i] = rng.randrange(256)
            ost, oout, gst, out = both(bytes(m))
            if ost == 0 and gst == 0:
                assert out == oout, (bi, it)
            elif ost != gst:
                diffs[(ost, gst)] = diffs.get((ost, gst), 0) + 1
            elif ost not in (zgpu.E_FAILED_READ_BLOCK_BODY, zgpu.E_FAILED_READ_BLOCK_HEADER):
                n_walk_second += 1
    assert not diffs, diffs
    assert n_walk_second > 50


def test_inorder_fallback_path(monkeypatch):
    """the in-order kernel (zg_k_lz) that serves frames with a block regenerating more than 128 KiB: forced on here (the engine reads
    its switches once, when it is created: the context comes after the switch)"""
    import zgpu
    monkeypatch.setenv("ZGPU_FORCE_INORDER", "1")
    ctx = zgpu.Context(0, dev=True)
    pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
    for name in sorted(man)[::9]:
        assert _sha(ctx.decode_all(pack[name], man[name]["size"])) == man[name]["sha256"], name
    spack, sman = read_pack("synthetic.pack"), read_manifest("synthetic.json")
    assert _sha(ctx.decode_all(spack["text_1m_l3.zst"], sman["text_1m_l3.zst"]["size"])) == sman["text_1m_l3.zst"]["sha256"]
    b = ctx.prepare(spack["text_1m_l3.zst"])
    b.run(); b.sync()
    assert b.timings()["lz"] > 0.5                             # (ms) it really went through zg_k_lz: a megabyte in order takes milliseconds
    b.close()
    ctx.close()


# ---- streaming / partial decode surface, dictionaries ------------------------------------------

def test_dict_corpus(ctx):
    """dict_test.rs:77-262: 207 frames that need the dictionary (tables, offset history and content all come from it)"""
    import zgpu
    pack, man = read_pack("dict_tests.pack"), read_manifest("dict_tests.json")
    raw = pack["dictionary"]
    d = zgpu.FrameDecoder(ctx)
    name0 = sorted(man)[0]
    st, _, _, _ = d.reset(pack[name0])
    assert st == zgpu.E_DICT_NOT_PROVIDED                       # before add_dict (frame_decoder.rs:212-217)
    assert d.add_dict(raw) == 618557512
    bad = []
    for name in sorted(man):
        z = pack[name]
        st, c, _, _ = d.reset(z)
        assert st == 0, name
        st, used, fin = d.decode_blocks(z[c:], zgpu.STRAT_ALL)
        out = d.collect()
        ok = (not st) and fin and _sha(out) == man[name]["sha256"] and d.bytes_read_from_source() == len(z)
        if not ok or d.get_checksum_from_data() != d.get_calculated_checksum():
            bad.append((name, st))
    assert not bad, bad[:5]
    # and through decode_all (several dictionary frames back to back)
    names = sorted(man)[:5]
    out = ctx.decode_all(b"".join(pack[n] for n in names), sum(man[n]["size"] for n in names))
    pos = 0
    for n in names:
        assert _sha(out[pos:pos + man[n]["size"]]) == man[n]["sha256"]
        pos += man[n]["size"]
    d.close()


@pytest.mark.parametrize("name", ["z000088.zst", "z000033.zst", "z000059.zst"])
def test_block_strategies_match_oracle(ctx, name):
    """decode_blocks with UptoBlocks / UptoBytes (frame_decoder.rs:361-373): after every call the counters and the bytes
    that may be collected (window-retention rule, decode_buffer.rs:182-188) equal the oracle's"""
    import zgpu
    z = read_pack("decodecorpus.pack")[name]
    for strat, n in ((zgpu.STRAT_UPTO_BLOCKS, 1), (zgpu.STRAT_UPTO_BLOCKS, 3), (zgpu.STRAT_UPTO_BYTES, 1), (zgpu.STRAT_UPTO_BYTES, 300000)):
        d, o = zgpu.FrameDecoder(ctx), oracle.FrameDecoder()
        st, c, _, _ = d.reset(z)
        ost, oc, _, _ = o.init(z)
        assert (st, c) == (ost, oc) == (0, c)
        pos, out, oout = c, b"", b""
        for _ in range(10000):
            st, used, fin = d.decode_blocks(z[pos:], strat, n)
            ost, oused, ofin = o.decode_blocks(z[pos:], strat, n)
            assert (st, used, fin) == (ost, oused, ofin), (name, strat, n)
            pos += used
            assert d.blocks_decoded() == o.blocks_decoded() and d.bytes_read_from_source() == o.bytes_read_from_source()
            assert d.can_collect() == o.can_collect()
            out += d.collect()
            oout += o.collect()
            if fin:
                break
        assert out == oout and d.is_finished()
        assert d.get_calculated_checksum() == o.calculated_checksum()
        d.close()


def test_decode_from_to(ctx):
    """tests/mod.rs:129-230: source split at 50 KiB, checksum delivered separately, byte counter exact"""
    import zgpu
    pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
    content = pack["z000088.zst"]
    d = zgpu.FrameDecoder(ctx)
    st, read1, out1 = d.decode_from_to(content[:50 * 1024], 1 << 20)
    assert st == 0
    st, read2, out2 = d.decode_from_to(content[read1:len(content) - 4], 1 << 20)
    assert st == 0 and read1 + read2 == len(content) - 4
    st, read3, out3 = d.decode_from_to(content[read1 + read2:], 1 << 20)
    assert (st, read3, out3) == (0, 4, b"")
    assert read1 + read2 + read3 == len(content)
    assert _sha(out1 + out2) == man["z000088.zst"]["sha256"]
    assert d.get_checksum_from_data() == d.get_calculated_checksum()
    d.close()


def test_incremental_read(ctx):
    """tests/mod.rs:382-404: a 3-byte target, then the rest"""
    import zgpu
    z = read_pack("test_fixtures.pack")["abc.txt.zst"]
    d = zgpu.FrameDecoder(ctx)
    st, c, _, _ = d.reset(z)
    assert st == 0
    st, _, out = d.decode_from_to(z[c:], 3)
    assert st == 0 and out == b"abc" and d.is_finished()
    assert d.read(3) == b"def"
    d.close()


def test_streaming_decoder(ctx):
    """tests/mod.rs:294-380: read_to_end through the io::Read mirror, then reuse of the decoder for another frame"""
    import io
    import zgpu
    pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
    s = zgpu.StreamingDecoder(io.BytesIO(pack["z000088.zst"]), ctx=ctx)
    res = s.read()
    assert _sha(res) == man["z000088.zst"]["sha256"]
    s2 = zgpu.StreamingDecoder(io.BytesIO(pack["z000068.zst"]), decoder=s.into_frame_decoder())
    chunks = []
    while True:                                                     # small reads exercise the window-retention rule
        c = s2.read(4096)
        if not c:
            break
        chunks.append(c)
    assert _sha(b"".join(chunks)) == man["z000068.zst"]["sha256"]
    # tests/mod.rs:700-725: the window limit applies to the streaming wrapper too
    big = read_pack("test_fixtures.pack")["window_256mib.zst"]
    with pytest.raises(zgpu.ZgpuError) as e:
        zgpu.StreamingDecoder(io.BytesIO(big), ctx=ctx)
    assert e.value.status == zgpu.E_WINDOW_SIZE_TOO_BIG


def test_dense_sequences_cut_tiles(ctx):
    """blocks with 3-6 output bytes per sequence (up to 42 K sequences per block): zg_k_flatten's tiles end early when they
    hold more sequences than fit (two per thread), both tile shapes"""
    import numpy as np
    import zgdata
    import zgpu
    rng = np.random.default_rng(7)
    for tok, V, lvl in ((3, 2000, 19), (5, 2000, 3)):
        vocab = rng.integers(0, 256, size=(V, tok), dtype=np.uint8)
        data = vocab[rng.integers(0, V, size=(600000 // tok,))].reshape(-1).tobytes()
        z = zgdata.zstd_compress(data, level=lvl)
        ref, _ = oracle.decode_frame_all(z)
        assert ref == data
        for flat_t in ("512", "1024"):
            os.environ["ZGPU_FLAT_T"] = flat_t
            try:
                c = zgpu.Context(0, dev=True)
                out = c.decode_all(z, len(data))
                c.close()
            finally:
                del os.environ["ZGPU_FLAT_T"]
            assert out == data, (tok, V, lvl, flat_t)


@pytest.mark.parametrize("env", [{"ZGPU_FLAT_T": "1024"}, {"ZGPU_UNIT_BLOCKS": "1"}, {"ZGPU_UNIT_BLOCKS": "3", "ZGPU_FLAT_T": "1024"}])
def test_corpus_other_shapes(env):
    """the whole corpus in one submit with the other flatten tile shape and with tiny units (a sweep step per block:
    every cross-block match goes through the sweep, every block start is a unit start)"""
    import zgdata
    import zgpu
    pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
    names = sorted(man)
    text = zgdata.text_like(3 << 20, seed=0x77)
    blob = b"".join(pack[n] for n in names) + zgdata.zstd_compress(text)
    for k, v in env.items():
        os.environ[k] = v
    try:
        c = zgpu.Context(0, dev=True)
        b = c.prepare(blob)
        b.run()
        b.sync()
        assert b.bad_status == 0, (b.bad_frame, b.bad_status)
        bad = [n for f, n in enumerate(names) if _sha(b.frame_bytes(f)) != man[n]["sha256"]]
        assert not bad, bad
        assert b.frame_bytes(len(names)) == text
        b.close()
        c.close()
    finally:
        for k in env:
            del os.environ[k]


def test_flat_scratch_matches_model(monkeypatch):
    """zg_k_flatten alone (the sweep is not launched): the flatten scratch of every pointer-mode unit — one effective offset per
    output byte, 0 for literal bytes — must equal the numpy model built from the oracle's sequences (tests/lz_model.py), for
    both tile shapes, tiny units included; the bytes of direct units (a frame's first unit, resolved to bytes by the flatten
    itself, zg_flat4.h) must be the oracle's plaintext right after the flatten. With ZGPU_DIRECT=0 the first units go through
    the scratch too."""
    import numpy as np
    import zgdata
    import zgpu
    import lz_model
    pack = read_pack("decodecorpus.pack")
    cases = [pack[n] for n in ("z000000.zst", "z000033.zst", "z000059.zst", "z000068.zst", "z000088.zst")]
    cases.append(zgdata.zstd_compress(zgdata.text_like(5 << 20, seed=0xF1)))
    plains = [oracle.decode_frame_all(z)[0] for z in cases]
    monkeypatch.setenv("ZGPU_DEBUG_NO_SWEEP", "1")
    monkeypatch.setenv("ZGPU_SPARSE_MAX", "0")      # (a frame that zg_k_sparse finishes gets no scratch words at all: keep every frame on the sweep path here)
    for shape, ub, direct in (("1024", None, "1"), ("1024", None, "0"), ("512", "2", "1"), ("512", "1", "0"), ("1024", "1", "1")):
        monkeypatch.setenv("ZGPU_FLAT_T", shape)
        monkeypatch.setenv("ZGPU_DIRECT", direct)
        if ub:
            monkeypatch.setenv("ZGPU_UNIT_BLOCKS", ub)
        else:
            monkeypatch.delenv("ZGPU_UNIT_BLOCKS", raising=False)
        c = zgpu.Context(0, dev=True)
        for ci, z in enumerate(cases):
            b = c.prepare(z)
            b.run()
            b.sync()
            units = b.units()
            e, bounds = lz_model.expected_scratch(z, [u[0] for u in units])
            ndirect = 0
            for ui, (fb, nb, base, size, noseq) in enumerate(units):
                want = e[bounds[ui]:bounds[ui + 1]]
                assert size == len(want), (shape, ci, ui)
                if noseq & 1:                   # only literal bytes: no scratch words are written, no sweep step reads them
                    assert not want.any(), (shape, ci, ui)
                    continue
                if noseq & 2:                   # direct unit: final bytes, no scratch
                    ndirect += 1
                    assert ui == 0 and direct == "1"
                    assert b.read(bounds[ui], size) == plains[ci][bounds[ui]:bounds[ui + 1]], (shape, ci, ui)
                    continue
                got = b.scratch_words(base, size)
                bad = np.flatnonzero(got != want)
                assert len(bad) == 0, (shape, ci, ui, int(bad[0]), got[bad[0]:bad[0] + 4], want[bad[0]:bad[0] + 4])
                # the literal bytes of a pointer-mode unit are in place already
                lit = np.flatnonzero(want == 0)
                have = np.frombuffer(b.read(bounds[ui], size), dtype=np.uint8)
                ref = np.frombuffer(plains[ci][bounds[ui]:bounds[ui + 1]], dtype=np.uint8)
                assert np.array_equal(have[lit], ref[lit]), (shape, ci, ui)
            assert ndirect <= 1 and (direct == "1" or ndirect == 0)
            b.close()
        c.close()


def test_new_surface_to_vec_writer_streaming(ctx):
    """decode_all_to_vec (frame_decoder.rs:591-610), collect_to_writer (:395-407) and the C-ABI StreamingDecoder mirror
    (streaming_decoder.rs:40-156) against the golden plaintexts"""
    import io
    import zgpu
    pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
    names = sorted(man)[3:40:6]
    blob = b"".join(pack[n] for n in names)
    out = ctx.decode_all_to_vec(blob)
    assert _sha(out) == _sha(b"".join(ctx.decode_all(pack[n], man[n]["size"]) for n in names))
    d = zgpu.FrameDecoder(ctx)
    for n in names[:3]:
        z = pack[n]
        st, c, _, _ = d.reset(z)
        st, used, fin = d.decode_blocks(z[c:], zgpu.STRAT_ALL)
        assert st == 0 and fin
        sink = io.BytesIO()
        assert d.collect_to_writer(sink) == man[n]["size"] and _sha(sink.getvalue()) == man[n]["sha256"]
        assert d.can_collect() == 0

        class Short:                     # a writer that takes at most 1000 bytes: the call ends after the first short write
            def __init__(self):
                self.got = b""

            def write(self, b):
                self.got += b[:1000]
                return min(len(b), 1000)
        st, c, _, _ = d.reset(z)
        d.decode_blocks(z[c:], zgpu.STRAT_ALL)
        w = Short()
        took = d.collect_to_writer(w)
        assert took == len(w.got) == min(1000, man[n]["size"])
        assert w.got + d.collect() == ctx.decode_all(z, man[n]["size"])
    d.close()
    for n in ("z000088.zst", "z000068.zst"):
        for step in (1 << 20, 4097):
            s = zgpu.CStreamingDecoder(ctx, io.BytesIO(pack[n] + b"trailing bytes the decoder must not touch"))
            chunks = []
            while True:
                c = s.read(step)
                if not c:
                    break
                chunks.append(c)
            assert _sha(b"".join(chunks)) == man[n]["sha256"], (n, step)
            s.close()


def test_force_dict_at_any_time_matches_oracle(ctx):
    """force_dict (frame_decoder.rs:229-243) after blocks were decoded: tables, offset history and dictionary content are
    replaced for the blocks that follow. Frames that do not ask for a dictionary get one forced before / after their first
    block; status and bytes must equal the oracle's, whatever they are."""
    import zgpu
    pack = read_pack("decodecorpus.pack")
    raw = read_pack("dict_tests.pack")["dictionary"]
    d = zgpu.FrameDecoder(ctx)
    did = d.add_dict(raw)
    for name in ("z000033.zst", "z000059.zst", "z000088.zst", "z000012.zst", "z000047.zst"):
        z = pack[name]
        for first in (0, 1, 2):
            o = oracle.FrameDecoder()
            assert o.add_dict(raw) == did
            st, c, _, _ = d.reset(z)
            ost, oc, _, _ = o.init(z)
            assert (st, c) == (ost, oc)
            pos = c
            if first:
                st, used, fin = d.decode_blocks(z[pos:], zgpu.STRAT_UPTO_BLOCKS, first)
                ost, oused, ofin = o.decode_blocks(z[pos:], oracle.STRAT_UPTO_BLOCKS, first)
                assert (st, used, fin) == (ost, oused, ofin)
                pos += used
                if fin:
                    continue
            assert d.force_dict(did) == o.force_dict(did) == 0
            st, used, fin = d.decode_blocks(z[pos:], zgpu.STRAT_ALL)
            ost, oused, ofin = o.decode_blocks(z[pos:], oracle.STRAT_ALL)
            assert (st, fin) == (ost, ofin), (name, first, st, ost)
            if st == 0:
                assert used == oused and d.collect() == o.collect(), (name, first)
    d.close()


def test_streaming_window_stays_bounded(ctx):
    """StreamingDecoder over a 192 MiB frame with 4 MiB reads, WITHOUT read-ahead (the reference's block-by-block schedule): linear time
    and a device window that does not grow with the frame (bytes the caller has drained are dropped when the window buffer is rebuilt,
    decode_buffer.rs:182-219). With read-ahead the bound is the budget: tests/test_gpu_stream.py."""
    import io
    import time
    import zgdata
    import zgpu
    plain = zgdata.text_like(192 << 20, seed=0x5D)
    z = zgdata.zstd_compress(plain)
    s = zgpu.CStreamingDecoder(ctx, io.BytesIO(z), read_ahead=zgpu.NO_READ_AHEAD)
    h = hashlib.sha256()
    marks, done, dev = [], 0, []
    t0 = time.perf_counter()
    while True:
        c = s.read(4 << 20)
        if not c:
            break
        h.update(c)
        done += len(c)
        dev.append(s.device_bytes())
        if done % (48 << 20) < (4 << 20):
            marks.append(time.perf_counter() - t0)
    assert done == len(plain) and h.digest() == hashlib.sha256(plain).digest()
    # bounded: the device window never holds more than a few windows' worth (window 2 MiB, reads of 4 MiB), whatever the frame's
    # length — and the second half of the frame needs no more than the first
    assert max(dev) <= 48 << 20, max(dev)
    assert max(dev[len(dev) // 2:]) <= max(dev[:len(dev) // 2]), (max(dev[:len(dev) // 2]), max(dev[len(dev) // 2:]))
    # linear: the last quarter does not take much longer than the second (a quadratic copy would take ~2.3x)
    assert len(marks) >= 4 and (marks[3] - marks[2]) < 1.7 * (marks[1] - marks[0]) + 0.05, marks
    s.close()


def test_uneven_four_stream_split(ctx):
    """4-stream Huffman literals split differently from the format's (regen + 3) / 4 rule: the reference compares only the total
    (literals_section_decoder.rs:150-155), so must the engine (zg_k_huf counts every stream to its end, zg_k_huf_uneven places
    them); a wrong total and a stream that does not end on its last bit keep their errors"""
    import zgpu
    from test_lane_logic_cpu import uneven_split_cases
    seen = set()
    for name, z, plain in uneven_split_cases():
        ost, oout = oracle.FrameDecoder().decode_all(z, 1 << 20)
        try:
            out, st = ctx.decode_all(z, 1 << 20), 0
        except zgpu.ZgpuError as e:
            out, st = None, e.status
        assert st == ost, (name, st, ost)
        if st == 0:
            assert out == oout == plain, name
        seen.add(st)
    assert seen == {0, 34, 35}, seen


@pytest.mark.parametrize("of_code", [29, 30, 31])
def test_offsets_of_2_pow_30_and_more(ctx, of_code):
    """offset codes 30 / 31: zg_k_seqpost must not take an offset >= 2^30 for a symbolic history reference"""
    import zgpu
    from test_lane_logic_cpu import big_offset_frame
    z = big_offset_frame(of_code, extra=5)
    st, _ = oracle.FrameDecoder().decode_all(z, 1 << 20)
    assert st in (52, 53)
    with pytest.raises(zgpu.ZgpuError) as e:
        ctx.decode_all(z, 1 << 20)
    assert e.value.status == st


def _run_batch(z, env):
    import zgpu
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        c = zgpu.Context(0, dev=True)
        b = c.prepare(z)
        assert b.parse_status == 0
        b.run(); b.sync()
        out, mode, bad = b.read(0, b.total_out), b.sweep_mode(), b.bad_status
        b.close(); c.close()
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    return out, mode, bad


def test_split_sweep_tails_and_heads():
    """a frame whose window (128 KiB) is much shorter than its units (512 KiB): only the last window of every unit is in
    the chain of sweep steps, the heads are filled beside it; same bytes as the plain chain and as the generator"""
    import zgdata
    data = zgdata.text_like(24 << 20, seed=0x5EE)
    z = zgdata.zstd_compress(data, window_log=17)
    out, mode, bad = _run_batch(z, {})
    assert bad == 0 and mode == 1 and out == data
    out, mode, bad = _run_batch(z, {"ZGPU_SWEEP_SPLIT": "0"})
    assert bad == 0 and mode == 0 and out == data
    # several frames of different lengths in one submit: a step holds the k-th unit of every frame that has one
    parts = [zgdata.text_like(n, seed=0x600 + i) for i, n in enumerate((9 << 20, 3 << 20, 700000, 5 << 20))]
    zz = b"".join(zgdata.zstd_compress(p, window_log=17) for p in parts)
    out, mode, bad = _run_batch(zz, {})
    assert bad == 0 and mode == 1 and out == b"".join(parts)


def test_split_sweep_is_repeated_when_a_match_exceeds_the_window():
    """the split relies on matches not reaching beyond the window; the engine is told a window shorter than the real one,
    zg_k_seqpost reports the longer matches and the sweep is repeated as a plain chain: right bytes either way"""
    import zgdata
    data = zgdata.text_like(24 << 20, seed=0x5EF)
    z = zgdata.zstd_compress(data)                     # window 2 MiB or more, units of 1 MiB
    out, mode, bad = _run_batch(z, {"ZGPU_UNIT_BLOCKS": "8", "ZGPU_SWEEP_W": "131072"})
    assert bad == 0 and mode == 2 and out == data
    out, mode, bad = _run_batch(z, {"ZGPU_UNIT_BLOCKS": "8"})
    assert bad == 0 and mode == 0 and out == data      # units shorter than the window: nothing to split


def test_sparse_frames_copy_their_matches_in_order():
    """a frame with hardly any sequences (literal-heavy data) skips the flatten scratch and the sweep: one wave copies its
    matches in order (zg_k_sparse). Same bytes as with the sweep, as the generator's, and — forced onto a frame full of
    sequences, where matches copy from matches — still the same."""
    import zgdata
    iso = zgdata.iso_like(6 << 20, seed=0x151)
    z = zgdata.zstd_compress(iso)
    for env in ({}, {"ZGPU_SPARSE_MAX": "0"}):
        out, mode, bad = _run_batch(z, env)
        assert bad == 0 and out == iso, env
    text = zgdata.text_like(3 << 20, seed=0x152)
    zt = zgdata.zstd_compress(text)
    out, mode, bad = _run_batch(zt + z, {"ZGPU_SPARSE_MAX": "100000000"})     # both frames through zg_k_sparse
    assert bad == 0 and out == text + iso


def test_trailing_frame_without_a_block(ctx):
    """a valid frame followed by a frame header and a first block header the host walk stops at (truncated, reserved type, too large, body
    cut): the batch then holds a trailing frame of zero blocks. Same verdict and the same bytes in front of it as the oracle."""
    import zgpu
    from test_verdict_order_cpu import _trailing_cases
    for k, m in enumerate(_trailing_cases()):
        ost, oout = oracle.FrameDecoder().decode_all(m, 1 << 24)
        try:
            out, gst = ctx.decode_all(m, 1 << 24), 0
        except zgpu.ZgpuError as e:
            out, gst = None, e.status
        assert ost != 0 and gst == ost, (k, ost, gst)


def test_content_size_that_lies_by_gigabytes_over_tiny_blocks(ctx):
    """a 1.5 MB frame of 500,000 (mostly empty) raw blocks that declares 60 GiB. The reference never looks at the field and
    decodes it; the engine must not turn the declaration into an allocation (it is bounded by what the block headers allow: exact for
    raw and RLE blocks) — same bytes as the oracle, and a following honest submit still works."""
    body = bytearray()
    want = bytearray()
    n = 500000
    for i in range(n):
        data = b"abc" if i % 1000 == 999 else b""
        body += (((len(data) << 3) | (1 if i == n - 1 else 0)).to_bytes(3, "little")) + data      # raw block: last | type 0 << 1 | size << 3
        want += data
    # magic, descriptor 0xC0 (8-byte Frame_Content_Size, window descriptor present), window descriptor 0 (1 KiB), FCS = 60 GiB
    z = bytes.fromhex("28b52ffd") + bytes([0xC0, 0x00]) + (60 << 30).to_bytes(8, "little") + bytes(body)
    ost, oout = oracle.FrameDecoder().decode_all(z, len(want) + 16)
    assert ost == 0 and oout == bytes(want)
    assert ctx.decode_all(z, len(want) + 16) == bytes(want)
    spack, sman = read_pack("synthetic.pack"), read_manifest("synthetic.json")
    assert _sha(ctx.decode_all(spack["text_1m_l3.zst"], sman["text_1m_l3.zst"]["size"])) == sman["text_1m_l3.zst"]["sha256"]


def test_output_sized_in_advance_and_frames_that_lie(ctx, monkeypatch):
    """Every frame declares its content size: the engine sizes the output before the run and enqueues the LZ77 stages behind the scan
    without waiting for the host. Frame_Content_Size is never checked by FrameDecoder::decode_all (frame_decoder.rs:541-577), so a
    frame may hold more (the device notices, the stages are repeated with the sizes the scan found) or less than it declares, and
    frames without the field take the sized-after-the-scan path: same bytes every time, and the same bytes with the shortcut off."""
    import zgdata
    import zgpu
    # (beyond the 2 MiB window of level 3: the frames carry a window descriptor, and the size field does not double as the window)
    plains = [zgdata.text_like(2600000 + 4099 * i, seed=0x4400 + i) for i in range(4)] + [zgdata.iso_like(2300000, seed=0x44)]
    zs = [zgdata.zstd_compress(p) for p in plains]
    want = b"".join(plains)

    def lie(z, delta):                                             # Frame_Header: magic(4) descriptor(1) window(1) FCS(4): frame.rs:6-85
        assert (z[4] >> 6) == 2 and not (z[4] >> 5) & 1 and not z[4] & 3, "expected a 4-byte FCS behind a window descriptor"
        fcs = int.from_bytes(z[6:10], "little") + delta
        return z[:6] + fcs.to_bytes(4, "little") + z[10:]
    nofcs = zgdata.zstd_compress(plains[1], content_size=False)
    cases = {"honest": b"".join(zs), "understated": zs[0] + lie(zs[1], -250000) + b"".join(zs[2:]), "overstated": lie(zs[0], 90000) + b"".join(zs[1:]),
             "all understated": b"".join(lie(z, -1000) for z in zs), "one undeclared": zs[0] + nofcs + b"".join(zs[2:])}
    for name, blob in cases.items():
        ost, oout = oracle.FrameDecoder().decode_all(blob, len(want) + 16)
        assert ost == 0 and oout == want, name
        assert ctx.decode_all(blob, len(want) + 16) == want, name
    monkeypatch.setenv("ZGPU_PRESIZE", "0")
    c2 = zgpu.Context(0, dev=True)
    for name, blob in cases.items():
        assert c2.decode_all(blob, len(want) + 16) == want, name
    c2.close()
