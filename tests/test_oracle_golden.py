"""Pins the CPU oracle against the reference's own golden fixtures.

Follows ruzstd/src/tests/decode_corpus.rs:2-189, dict_test.rs:77-262, tests/mod.rs:576-741 and
fuzz_regressions.rs:2-27: output bytes, consumed-byte counter and checksum must match.
"""
import hashlib
import os

import pytest

import oracle
from golden_io import read_manifest, read_pack

REF = "/root/reference/ruzstd"


def _check(name, z, meta, dict_raw=None, max_window=None):
    out, d = oracle.decode_frame_all(z, dict_raw=dict_raw, max_window=max_window)
    assert len(out) == meta["size"], name
    assert hashlib.sha256(out).hexdigest() == meta["sha256"], name
    assert d.bytes_read_from_source() == len(z), name          # decode_corpus.rs:102-110
    ck = d.checksum_from_data()
    if ck is not None:
        assert d.calculated_checksum() == ck, name              # decode_corpus.rs:61-82
    assert d.is_finished()
    return d


def test_decode_corpus():
    pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
    assert len(man) == 101
    for name in sorted(man):
        _check(name, pack[name], man[name])


def test_decode_from_to_as_the_reference_tests_it():
    """tests/mod.rs:129-230 (source cut at 50 KiB, then everything but the checksum, then the checksum alone: (4, 0); the byte counter adds
    up to the file's length) and :382-404 (a three-byte target, the rest through read())"""
    pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
    content = pack["z000088.zst"]
    d = oracle.FrameDecoder()
    st, read1, out1 = d.decode_from_to(content[:50 * 1024], 1 << 20)
    assert st == 0
    st, read2, out2 = d.decode_from_to(content[read1:len(content) - 4], 1 << 20)
    assert st == 0 and read1 + read2 == len(content) - 4
    st, read3, out3 = d.decode_from_to(content[read1 + read2:], 1 << 20)
    assert (st, read3, out3) == (0, 4, b"")
    assert read1 + read2 + read3 == len(content)
    assert hashlib.sha256(out1 + out2).hexdigest() == man["z000088.zst"]["sha256"]
    assert d.checksum_from_data() == d.calculated_checksum()
    z = read_pack("test_fixtures.pack")["abc.txt.zst"]
    d = oracle.FrameDecoder()
    st, c, _, _ = d.init(z)
    assert st == 0
    st, _, out = d.decode_from_to(z[c:], 3)
    assert st == 0 and out == b"abc" and d.is_finished()
    assert d.read(3) == b"def"
    # every corpus file in pieces of every size: the bytes, the counter
    for name in sorted(man)[:40]:
        z = pack[name]
        for piece in (1, 7, 1000, 70000):
            d = oracle.FrameDecoder()
            st, pos, _, _ = d.init(z)
            assert st == 0
            out, stall = b"", 0
            while not d.is_finished() and stall < 3:
                st, r, o = d.decode_from_to(z[pos:pos + piece] if stall == 0 else z[pos:], 1 << 22)
                assert st == 0, (name, piece)
                stall = stall + 1 if r == 0 else 0          # (a piece smaller than the next block: hand over everything next time)
                pos += r
                out += o
            out += d.read(1 << 22)
            assert hashlib.sha256(out).hexdigest() == man[name]["sha256"] and pos == len(z), (name, piece)


def test_dict_corpus():
    pack, man = read_pack("dict_tests.pack"), read_manifest("dict_tests.json")
    assert len(man) == 207
    dict_raw = pack["dictionary"]
    for name in sorted(man):
        d = _check(name, pack[name], man[name], dict_raw=dict_raw)
        assert d.L.zor_dict_id(d.h) == 618557512                # dict_test.rs id


def test_dict_missing_is_error():
    pack, man = read_pack("dict_tests.pack"), read_manifest("dict_tests.json")
    name = sorted(man)[0]
    with pytest.raises(oracle.OracleError) as e:
        oracle.decode_frame_all(pack[name])
    assert e.value.status == oracle.ZOR_DICT_NOT_PROVIDED


def test_window_fixtures():
    pack, man = read_pack("test_fixtures.pack"), read_manifest("test_fixtures.json")
    _check("window_8mib.zst", pack["window_8mib.zst"], man["window_8mib.zst"])
    _check("window_128mib.zst", pack["window_128mib.zst"], man["window_128mib.zst"])
    # 256 MiB window: rejected at the default limit, accepted when raised (tests/mod.rs:597-637)
    with pytest.raises(oracle.OracleError) as e:
        oracle.decode_frame_all(pack["window_256mib.zst"])
    assert e.value.status == oracle.ZOR_WINDOW_SIZE_TOO_BIG
    _check("window_256mib.zst", pack["window_256mib.zst"], man["window_256mib.zst"], max_window=300 * 1024 * 1024)


def test_decode_all_multiframe_and_skippable():
    """tests/mod.rs:490-574: skippable frames between frames, TargetTooSmall, truncated inputs."""
    pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
    z = pack["z000088.zst"]
    plain_len = man["z000088.zst"]["size"]
    skip = bytes([0x50, 0x2A, 0x4D, 0x18, 3, 0, 0, 0, 1, 2, 3])
    data = skip + z + skip + z + skip
    d = oracle.FrameDecoder()
    st, out = d.decode_all(data, plain_len * 2)
    assert st == 0 and len(out) == plain_len * 2
    assert hashlib.sha256(out[:plain_len]).hexdigest() == man["z000088.zst"]["sha256"]
    assert out[:plain_len] == out[plain_len:]
    st, _ = d.decode_all(data, plain_len * 2 - 1)
    assert st == oracle.ZOR_TARGET_TOO_SMALL
    st, _ = d.decode_all(z[:-5], plain_len)                      # truncated frame
    assert st in (oracle.ZOR_FAILED_READ_BLOCK_BODY, 9, 11)
    st, _ = d.decode_all(skip[:-1], 10)                          # truncated skippable frame
    assert st == oracle.ZOR_FAILED_SKIP_FRAME


def test_fuzz_artifacts_do_not_crash():
    pack = read_pack("fuzz_artifacts.pack")
    dict_raw = read_pack("dict_tests.pack")["dictionary"]
    n = 0
    for name, data in pack.items():
        if not (name.startswith("decode/") or name.startswith("decode_dict/") or name.startswith("interop/")):
            continue
        d = oracle.FrameDecoder()
        if name.startswith("decode_dict/"):
            d.add_dict(dict_raw)
        st, c, _, _ = d.init(data)
        if st == 0:
            d.decode_blocks(data[c:], oracle.STRAT_ALL)
            d.collect()
        n += 1
    assert n >= 44


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted (GPU box)")
def test_packs_match_reference_tree_and_plaintexts():
    """In the build container: the packs are exactly the reference's fixtures and the oracle's bytes equal
    the reference's plaintext files byte for byte (not only by hash)."""
    pack = read_pack("decodecorpus.pack")
    names = sorted(f for f in os.listdir(os.path.join(REF, "decodecorpus_files")) if f.endswith(".zst"))
    assert names == sorted(pack)
    for name in names:
        with open(os.path.join(REF, "decodecorpus_files", name), "rb") as f:
            assert f.read() == pack[name]
        with open(os.path.join(REF, "decodecorpus_files", name[:-4]), "rb") as f:
            plain = f.read()
        out, _ = oracle.decode_frame_all(pack[name])
        assert out == plain, name
