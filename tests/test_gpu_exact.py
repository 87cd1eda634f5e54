"""The reference's DecodeBuffer bookkeeping on the GPU (zg_k_exact, zstd-rs_amd/csrc/zg_exact.h): hand-made frames whose matches
reach beyond their window, across the drain points of FrameDecoder::decode_all, or into a dictionary after
total_output_counter has passed window_size. Every surface must decide like the oracle: same bytes, or the same error leaf."""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "zstd-rs_amd"))
import oracle
from golden_io import read_pack
from test_exact_cpu import CASES, K, build, frame, lit_block, raw_block, seq_block

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import zgpu
    c = zgpu.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def pool():
    import zgpu
    p = zgpu.Pool(devices=[0])
    yield p
    p.close()


@pytest.mark.parametrize("name,blocks", CASES, ids=[c[0] for c in CASES])
def test_decode_all_and_decode_blocks_decide_like_the_oracle(ctx, pool, name, blocks):
    import zgpu
    z = build(name, blocks)
    # FrameDecoder::decode_all (drains every MiB): zgpu_decode_all and the queue's decode_all
    ost, oout = oracle.FrameDecoder().decode_all(z, 1 << 26)
    for surface in ("ctx", "pool"):
        try:
            if surface == "ctx":
                out, st = ctx.decode_all(z, 1 << 26), 0
            else:
                out, st = pool.decode_all(z, 1 << 26), 0
        except zgpu.ZgpuError as e:
            out, st = None, e.status
        assert st == ost, (name, surface, st, ost)
        if st == 0:
            assert out == oout, (name, surface)
    # FrameDecoder::decode_blocks(All) on the first frame: nothing is drained inside the run
    d, o = zgpu.FrameDecoder(ctx), oracle.FrameDecoder()
    st, c, _, _ = d.reset(z)
    ost, oc, _, _ = o.init(z)
    assert (st, c) == (ost, oc) == (0, 6)
    st, used, fin = d.decode_blocks(z[c:], zgpu.STRAT_ALL)
    ost, oused, ofin = o.decode_blocks(z[c:], oracle.STRAT_ALL)
    assert (st, fin) == (ost, ofin), (name, st, ost)
    if st == 0:
        assert used == oused and d.collect() == o.collect(), name
    d.close()


def test_streamed_runs_keep_the_counter_and_the_reach(ctx):
    """the same frame block run by block run with reads in between (what StreamingDecoder does): what the caller has drained is
    out of reach, and the counter carried from run to run picks the error leaf"""
    import zgpu
    for blocks in ([raw_block(K, 1)] * 12 + [seq_block(11 * K, last=True)],
                   [lit_block(4000)] * 70 + [seq_block(69 * 4000, last=True)],
                   [raw_block(K, 1)] * 12 + [seq_block(K - 100, last=True)]):
        z = frame(*blocks)
        for per_run in (1, 5):
            d, o = zgpu.FrameDecoder(ctx), oracle.FrameDecoder()
            st, c, _, _ = d.reset(z)
            ost, oc, _, _ = o.init(z)
            assert (st, c) == (ost, oc)
            pos, got, want = c, b"", b""
            while True:
                st, used, fin = d.decode_blocks(z[pos:], zgpu.STRAT_UPTO_BLOCKS, per_run)
                ost, oused, ofin = o.decode_blocks(z[pos:], oracle.STRAT_UPTO_BLOCKS, per_run)
                assert (st, fin) == (ost, ofin), (len(blocks), per_run, st, ost)
                if st:
                    break
                assert used == oused
                pos += used
                got += d.read(1 << 22)
                want += o.read(1 << 22)
                assert got == want
                if fin:
                    break
            d.close()


def test_dictionary_reach_ends_with_the_counter(ctx):
    import zgpu
    raw = read_pack("dict_tests.pack")["dictionary"]
    cases = {
        "into_dict_early": [lit_block(1000), seq_block(1000 + 50, last=True)],
        "dict_only_matches_are_not_counted": [lit_block(10)] + [seq_block(5000, lits=b"")] * 3 + [seq_block(5000, last=True)],
        "into_dict_counter_beyond_window": [lit_block(4000)] * 33 + [seq_block(33 * 4000 + 50, last=True)],
        "into_dict_counter_kept_small_by_raw_blocks": [raw_block(K, 1), raw_block(K, 2), lit_block(10), seq_block(2 * K + 10 + 50, last=True)],
        "beyond_the_dictionary": [lit_block(1000), seq_block(1000 + (1 << 22), last=True)],
    }
    d = zgpu.FrameDecoder(ctx)
    did = d.add_dict(raw)
    for name, blocks in cases.items():
        z = frame(*blocks)
        o = oracle.FrameDecoder()
        assert o.add_dict(raw) == did
        st, c, _, _ = d.reset(z)
        ost, oc, _, _ = o.init(z)
        assert (st, c) == (ost, oc)
        assert d.force_dict(did) == o.force_dict(did) == 0
        st, used, fin = d.decode_blocks(z[c:], zgpu.STRAT_ALL)
        ost, oused, ofin = o.decode_blocks(z[c:], oracle.STRAT_ALL)
        assert (st, fin) == (ost, ofin), (name, st, ost)
        if st == 0:
            assert d.collect() == o.collect(), name
    d.close()


def test_many_sequences_per_block_with_dictionary(ctx):
    """blocks of hundreds of one-byte-code sequences that reach into the dictionary, into the frame, or nowhere: status and bytes as the
    oracle's, whichever sequence of the block is the first the reference objects to"""
    import random
    import zgpu
    from test_exact_cpu import multi_seq_block
    raw = read_pack("dict_tests.pack")["dictionary"]
    d = zgpu.FrameDecoder(ctx)
    did = d.add_dict(raw)
    rng = random.Random(78)
    seen = set()
    for case in range(18):
        n = rng.choice([5, 255, 257, 600, 1100])
        pre = rng.choice([0, 100, 3000])
        code_lo = 1 << (17 if case % 3 == 0 else 15 if case % 3 == 1 else 16)
        offs = []
        for i in range(n):
            kind = rng.randrange(4) if case % 3 != 1 else 4
            at = pre + 3 * i
            if kind == 0:
                o = code_lo - 3 + rng.randrange(0, 200)
            elif kind == 1:
                o = min(code_lo - 3 + at + rng.randrange(0, 40), 2 * code_lo - 4)
            elif kind == 4:
                o = code_lo - 3 + rng.randrange(0, 8000)          # well inside the dictionary's content
            else:
                o = code_lo - 3 + rng.randrange(0, 60000)
            offs.append(o)
        blocks = ([lit_block(pre)] if pre else []) + [multi_seq_block(offs, last=True)]
        if case % 3 == 0:
            blocks = [lit_block(4000)] * 33 + blocks
        z = frame(*blocks)
        o = oracle.FrameDecoder()
        assert o.add_dict(raw) == did
        st, c, _, _ = d.reset(z)
        ost, oc, _, _ = o.init(z)
        assert (st, c) == (ost, oc)
        assert d.force_dict(did) == o.force_dict(did) == 0
        st, used, fin = d.decode_blocks(z[c:], zgpu.STRAT_ALL)
        ost, oused, ofin = o.decode_blocks(z[c:], oracle.STRAT_ALL)
        assert (st, fin) == (ost, ofin), (case, n, pre, st, ost)
        if st == 0:
            assert d.collect() == o.collect(), case
        seen.add(st)
    assert seen == {0, 52, 53}, seen
    d.close()


def test_dictionary_spliced_behind_drained_bytes(ctx):
    """a dictionary, three raw blocks (not counted: total_output_counter stays below the window), a collect() that drains down to the
    window, then a match that starts in front of what is left: the reference copies the dictionary's tail and goes on with the OLDEST byte
    it still holds (decode_buffer.rs:159-163) — not with the drained byte that once stood in front of it. Bytes and verdicts as the oracle's,
    through the FrameDecoder mirror; the same frame without the drain takes the drained byte (both are checked)."""
    import zgpu
    raw = read_pack("dict_tests.pack")["dictionary"]
    d = zgpu.FrameDecoder(ctx)
    did = d.add_dict(raw)
    seen = set()
    for off_extra in (1, 2, 3, 4, 50, 40000, 200000):
        for drain in (True, False):
            for tail_blocks in ([seq_block(K + off_extra, last=True)],
                                [seq_block(K + off_extra), raw_block(1000, 9), seq_block(K + 7 + 1000 + off_extra, last=True)]):
                z = frame(*([raw_block(K, 1), raw_block(K, 2), raw_block(K, 3)] + tail_blocks))
                o = oracle.FrameDecoder()
                assert o.add_dict(raw) == did
                st, c, _, _ = d.reset(z)
                ost, oc, _, _ = o.init(z)
                assert (st, c) == (ost, oc) and d.force_dict(did) == o.force_dict(did) == 0
                got, want = b"", b""
                used = oused = 0
                if drain:
                    st, used, fin = d.decode_blocks(z[c:], zgpu.STRAT_UPTO_BLOCKS, 3)
                    ost, oused, ofin = o.decode_blocks(z[c:], oracle.STRAT_UPTO_BLOCKS, 3)
                    assert (st, used, fin) == (ost, oused, ofin)
                    got, want = d.collect(), o.collect()
                    assert got == want and len(got) == 2 * K
                st, u2, fin = d.decode_blocks(z[c + used:], zgpu.STRAT_ALL)
                ost, ou2, ofin = o.decode_blocks(z[c + oused:], oracle.STRAT_ALL)
                assert (st, fin) == (ost, ofin), (off_extra, drain, st, ost)
                seen.add(ost)
                assert d.collect() == o.collect(), (off_extra, drain)
    assert 0 in seen and 53 in seen
    d.close()


def test_dictionary_splice_behind_a_drain_inside_decode_all(ctx):
    """decode_all on a frame that names a dictionary is ONE submit (decode_all_per_frame -> decode_blocks(All) with the drain
    rule of decode_all). With fewer than 1 MiB in front nothing is drained inside it and the splice yields the oracle's bytes; with a
    drain inside the submit the device still holds the drained bytes in place: zg_k_exact refuses that submit (never wrong bytes), and
    since late in round 6 decode_all then decodes the frame again on the reference's own schedule — rounds of UptoBytes(1 MiB) + read(),
    the drains between submits — and returns the reference's bytes."""
    import zgpu
    raw = read_pack("dict_tests.pack")["dictionary"]
    did = ctx.add_dict(raw)
    hdr = bytes([0x28, 0xB5, 0x2F, 0xFD, 0x03, (17 - 10) << 3]) + did.to_bytes(4, "little")
    for nraw, held in ((3, 3 * K), (9, 2 * K)):
        z = hdr + b"".join([raw_block(K, i) for i in range(nraw)] + [seq_block(held + 4 + 10, last=True)])
        o = oracle.FrameDecoder()
        assert o.add_dict(raw) == did
        ost, oout = o.decode_all(z, 1 << 26)
        assert ost == 0
        try:
            out, st = ctx.decode_all(z, 1 << 26), 0
        except zgpu.ZgpuError as e:
            out, st = None, e.status
        assert st == 0 and out == oout, (nraw, st)
        assert ctx.decode_all_to_vec(z) == oout
    # two such frames and an ordinary one between them in one call; a target that is too small by one byte
    from golden_io import read_pack as rp
    plain = rp("decodecorpus.pack")["z000033.zst"]
    z2 = z + plain + z
    o = oracle.FrameDecoder()
    o.add_dict(raw)
    ost, oout = o.decode_all(z2, 1 << 26)
    assert ost == 0 and ctx.decode_all(z2, 1 << 26) == oout
    with pytest.raises(zgpu.ZgpuError) as e:
        ctx.decode_all(z2, len(oout) - 1)
    o = oracle.FrameDecoder()
    o.add_dict(raw)
    assert e.value.status == o.decode_all(z2, len(oout) - 1)[0] == 12
