"""The engine's lane routines (zstd-rs_amd/csrc/zg_dev.h) and host parser, run on the CPU through the test-only
harness, against the golden fixtures and — block by block — against the oracle's intermediates."""
import hashlib

import pytest

import emu
import oracle
from golden_io import read_manifest, read_pack


def test_corpus_bit_exact():
    pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
    for name in sorted(man):
        e = emu.EmuBatch(pack[name])
        assert e.parse_status == 0, name
        assert e.nframes == 1
        out, st = e.frame_bytes(0)
        assert st == 0, (name, st)
        assert len(out) == man[name]["size"], name
        assert hashlib.sha256(out).hexdigest() == man[name]["sha256"], name


def test_corpus_bit_exact_reference_sequence_routine():
    """the plain (one read per field) sequence routine, kept as the readable statement of the windowed one"""
    pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
    for name in sorted(man)[::5]:
        out, st = emu.EmuBatch(pack[name], fast_seq=False).frame_bytes(0)
        assert st == 0 and hashlib.sha256(out).hexdigest() == man[name]["sha256"], name


def test_window_fixtures():
    pack, man = read_pack("test_fixtures.pack"), read_manifest("test_fixtures.json")
    for name in ("window_8mib.zst", "window_128mib.zst"):
        out, st = emu.EmuBatch(pack[name]).frame_bytes(0)
        assert st == 0 and hashlib.sha256(out).hexdigest() == man[name]["sha256"]
    assert emu.EmuBatch(pack["window_256mib.zst"]).parse_status == 6        # WindowSizeTooBig at the default limit
    out, st = emu.EmuBatch(pack["window_256mib.zst"], max_window=300 << 20).frame_bytes(0)
    assert st == 0 and hashlib.sha256(out).hexdigest() == man["window_256mib.zst"]["sha256"]


def _oracle_blocks(z):
    """decode block by block with the oracle, collecting its intermediates"""
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


@pytest.mark.parametrize("name", ["z000000.zst", "z000013.zst", "z000033.zst", "z000059.zst", "z000088.zst", "z000099.zst"])
def test_intermediates_match_oracle(name):
    z = read_pack("decodecorpus.pack")[name]
    ob = _oracle_blocks(z)
    e = emu.EmuBatch(z)
    assert e.nblocks == len(ob)
    hist = [1, 4, 8]
    for b, rec in enumerate(ob):
        info = e.block(b)
        assert info["btype"] == rec["type"] and info["status"] == 0
        assert e.block_hist(b) == hist, (name, b)
        hist = rec["hist_after"]
        if rec["type"] != 2:
            continue
        if info["lit_type"] >= 2:   # Huffman literals: bytes and the table they were decoded with
            assert e.block_literals(b, info["regen_size"]) == rec["literals"], (name, b)
            tab, mb = e.huf_slot(info["huf_slot"])
            oents, omb = rec["huf"]
            assert mb == omb
            assert [(tab[i] & 255, tab[i] >> 8) for i in range(1 << mb)] == oents
        seqs = e.block_sequences(b, info["nseq"])
        oseq = rec["sequences"]
        assert len(oseq) == info["nseq"]
        lit_pos = out_pos = 0
        h = e.block_hist(b)
        for (of, ml, mdst, lit_start), (oll, oml, _oof, oactual) in zip(seqs, oseq):
            tag, k = of >> 30, of & 0x3FFFFFFF
            actual = of if tag == 0 else max(h[tag - 1] - k, 0)
            assert (actual, ml, mdst, lit_start) == (oactual, oml, out_pos + oll, lit_pos), (name, b)
            lit_pos += oll
            out_pos += oll + oml
        if info["nseq"]:            # the three FSE tables this block decoded with
            for k, slot in enumerate((info["ll_slot"], info["of_slot"], info["ml_slot"])):
                oents, olog, orle = rec["fse"][k]
                p, logs = e.fse_slot(slot)
                off = (0, 1024, 512)[k]
                if orle >= 0:
                    assert logs[k] == 0 and ((p[off] >> 20) & 63) == orle
                else:
                    assert logs[k] == olog
                    got = [(p[off + i] & 0xFFFF, (p[off + i] >> 16) & 15, (p[off + i] >> 20) & 63) for i in range(1 << olog)]
                    assert got == oents, (name, b, k)


def test_fuzz_artifacts_do_not_crash_and_agree_on_failure():
    pack = read_pack("fuzz_artifacts.pack")
    for name, data in pack.items():
        if not (name.startswith("decode/") or name.startswith("interop/")):
            continue
        e = emu.EmuBatch(data)
        # oracle verdict through decode_all
        st, out = oracle.FrameDecoder().decode_all(data, 1 << 24)
        ok_engine = e.parse_status == 0 and all(e.frame(f)[2] == 0 for f in range(e.nframes))
        if st == 0:
            assert ok_engine, name
            got = b"".join(e.frame_bytes(f)[0] for f in range(e.nframes))
            assert got == out, name


def test_multiframe_and_skippable():
    pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
    z1, z2 = pack["z000088.zst"], pack["z000033.zst"]
    skip = bytes([0x50, 0x2A, 0x4D, 0x18, 3, 0, 0, 0, 1, 2, 3])
    e = emu.EmuBatch(skip + z1 + skip + z2 + skip)
    assert e.parse_status == 0 and e.nframes == 2
    assert hashlib.sha256(e.frame_bytes(0)[0]).hexdigest() == man["z000088.zst"]["sha256"]
    assert hashlib.sha256(e.frame_bytes(1)[0]).hexdigest() == man["z000033.zst"]["sha256"]
    assert e.frame(1)[0] == man["z000088.zst"]["size"]        # frames are packed back to back
    assert emu.EmuBatch(skip[:-1]).parse_status == 13          # FailedToSkipFrame
    assert emu.EmuBatch(z1[:-5]).parse_status in (9, 10, 11)


def big_offset_frame(of_code, extra=0):
    """one frame, one compressed block: 4 raw literals, one sequence with offset code `of_code` in RLE mode (the predefined
    OF table stops at code 28), LL and ML predefined. Used by the CPU and GPU tests of offsets >= 2^30."""
    # the reversed bitstream, in read order from its top: final-bit marker, LL state (6 bits), OF state (RLE: 0 bits),
    # ML state (6 bits), then the extra bits OF, ML, LL (state 0 of the predefined LL / ML tables carries none)
    acc, accn = 1, 1
    for v, w in ((0, 6), (0, 6), (extra, of_code)):
        acc = (acc << w) | (v & ((1 << w) - 1))
        accn += w
    stream = acc.to_bytes((accn + 7) // 8, "little")
    lits = b"abcd"
    body = bytes([len(lits) << 3]) + lits + bytes([1, 0x10, of_code]) + stream     # raw literals; nseq 1, modes LL predefined / OF RLE / ML predefined, RLE symbol
    bh = (len(body) << 3) | (2 << 1) | 1                  # last block, compressed
    return bytes([0x28, 0xB5, 0x2F, 0xFD, 0x00, 0x50]) + bh.to_bytes(3, "little") + body   # window descriptor 0x50: 1 MiB


@pytest.mark.parametrize("of_code", [29, 30, 31])
def test_offsets_of_2_pow_30_and_more(of_code):
    """offset codes 30 and 31 (offset >= 2^30) must not be taken for the symbolic history references the engine tags with the
    top two bits: they travel as ZG_OFF_HUGE, the serial model rejects them, and zg_k_exact's source picks the reference's leaf"""
    z = big_offset_frame(of_code, extra=5)
    st, _ = oracle.FrameDecoder().decode_all(z, 1 << 20)
    e = emu.EmuBatch(z)
    est = e.parse_status or e.frame(0)[2]
    assert st in (52, 53), st
    assert est in (52, 53), est
    assert e.exact(drain_rule=1)[0][0] == st
    # with 1 GiB held undrained the reference could serve such an offset; the engine has lost its value: ZGPU_E_UNSUPPORTED
    if of_code >= 30:
        assert e.exact(drain_rule=0, prior_out=3 << 30, prior_reach=3 << 30, prior_counted=3 << 30)[0][0] == 80


def uneven_split_cases():
    """(name, frame, plaintext or None): the four literal streams of a libzstd block written back with other splits"""
    import os
    import random
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import zgdata
    import uneven
    rng = random.Random(5)
    plain = bytes(rng.choice(b"aaaaaaabbbbccdeefghijklmnop qrstu") for _ in range(20000))
    z = zgdata.zstd_compress(plain)
    out = [("format_split", uneven.resplit(z, (0, 0, 0))[0], plain)]
    for i, deltas in enumerate(((5, -3, 7), (-100, 200, -50), (300, 300, 300), (-4000, -4000, -4000), (1, 0, 0))):
        out.append(("uneven_%d" % i, uneven.resplit(z, deltas)[0], plain))
    # and what must still fail: a wrong total, a stream that does not end on its last bit (with the format's split and another one)
    out.append(("wrong_total_plus", uneven.resplit(z, (2, 2, 2), regen_bias=1)[0], None))
    out.append(("wrong_total_minus", uneven.resplit(z, (-9, 2, 2), regen_bias=-1)[0], None))
    out.append(("spare_bit_format_split", uneven.resplit(z, (0, 0, 0), spare_bits=(0, 1, 0, 0))[0], None))
    out.append(("spare_bits_uneven", uneven.resplit(z, (40, -7, 3), spare_bits=(0, 0, 3, 0))[0], None))
    out.append(("spare_bits_first_overfull", uneven.resplit(z, (500, -7, 3), spare_bits=(2, 0, 0, 0))[0], None))
    return out


def test_uneven_four_stream_split_is_accepted_like_the_reference():
    """4-stream Huffman literals whose streams do not hold (regen + 3) / 4 symbols each: the reference only checks the total
    (literals_section_decoder.rs:150-155); the engine must decode them to the same bytes"""
    seen = set()
    for name, z, plain in uneven_split_cases():
        st, out = oracle.FrameDecoder().decode_all(z, 1 << 20)
        assert (st == 0 and out == plain) if plain is not None else st != 0, (name, st)
        e = emu.EmuBatch(z)
        assert e.parse_status == 0 and e.frame(0)[2] == st, (name, e.frame(0), st)
        if st == 0:
            assert e.frame_bytes(0)[0] == plain, name
        seen.add(st)
    assert seen == {0, 34, 35}, seen       # ok, BitstreamReadMismatch, DecodedLiteralCountMismatch
