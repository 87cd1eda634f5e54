"""Known-answer tests the reference holds inside its hot-path files, replayed on the oracle
(SURVEY.md §8c item 5)."""
import ctypes as C

import oracle

LL_DEFAULT = [4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1,
              -1, -1, -1, -1]


def _build(acc_log, probs, max_symbol):
    L = oracle.lib()
    arr = (C.c_int32 * len(probs))(*probs)
    out = (oracle.FseEntry * (1 << acc_log))()
    st = L.zor_fse_build_from_probs(acc_log, arr, len(probs), max_symbol, out)
    assert st == 0
    return [(e.symbol, e.num_bits, e.base_line) for e in out]


def test_ll_default_table():
    """decoding/sequence_section_decoder.rs:444-487"""
    t = _build(6, LL_DEFAULT, 35)
    assert len(t) == 64
    assert t[0] == (0, 4, 0)
    assert t[19] == (27, 6, 0)
    assert t[39] == (25, 4, 16)
    assert t[60] == (35, 6, 0)
    assert t[59] == (24, 5, 32)


def test_fse_distribution_shape():
    """fse/mod.rs:21-29 distribution; every symbol's states must tile [0, size) exactly once"""
    probs = [0, 0, -1, 3, 2, 2, (1 << 6) - 8]
    t = _build(6, probs, 255)
    for sym, p in enumerate(probs):
        ents = [e for e in t if e[0] == sym]
        assert len(ents) == (1 if p == -1 else p)
        covered = sorted((bl, bl + (1 << nb)) for _, nb, bl in ents)
        if ents:
            assert covered[0][0] == 0 and covered[-1][1] == 64
            for a, b in zip(covered, covered[1:]):
                assert a[1] == b[0]


def test_revbits_it_works():
    """bit_io/bit_reader_reverse.rs:168-183"""
    L = oracle.lib()
    data = bytes([0b10101010, 0b01010101])
    widths = bytes([1, 1, 1, 4, 4, 4, 4, 4])
    vals = (C.c_uint64 * len(widths))()
    rem = L.zor_revbits_read(data, len(data), widths, len(widths), vals)
    assert list(vals) == [0, 1, 0, 0b1010, 0b1101, 0b0101, 0, 0]
    assert rem == -7


ENC = bytes([0xC1, 0x41, 0x08, 0x00, 0x00, 0xEC, 0xC8, 0x96, 0x42, 0x79, 0xD4, 0xBC, 0xF7, 0x2C, 0xD5, 0x48])
NUM = 0x48D52CF7BCD4794296C8EC00000841C1


def _pattern():
    widths, bits_read, x = [], 0, 0
    while True:
        x += 3
        nb = x % 16
        if bits_read > 128 - nb:
            nb = 128 - bits_read
        widths.append(nb)
        bits_read += nb
        if bits_read >= 128:
            return widths


def test_bitreader_reversed_128():
    """tests/bit_reader.rs:1-38"""
    L = oracle.lib()
    w = _pattern()
    vals = (C.c_uint64 * len(w))()
    rem = L.zor_revbits_read(ENC, 16, bytes(w), len(w), vals)
    acc, rd = 0, 0
    for nb, v in zip(w, vals):
        rd += nb
        acc |= v << (128 - rd)
    assert acc == NUM and rem == 0


def test_bitreader_normal_128():
    """tests/bit_reader.rs:40-79"""
    L = oracle.lib()
    w = _pattern()
    vals = (C.c_uint64 * len(w))()
    assert L.zor_fwdbits_read(ENC, 16, bytes(w), len(w), vals) == 0
    acc, rd = 0, 0
    for nb, v in zip(w, vals):
        acc |= v << rd
        rd += nb
    assert acc == NUM
    assert L.zor_fwdbits_read(ENC, 16, bytes(w + [1]), len(w) + 1, (C.c_uint64 * (len(w) + 1))()) == -1


def test_offset_history_underflow():
    """decoding/sequence_execution.rs:124-133"""
    L = oracle.lib()
    h = (C.c_uint32 * 3)(0, 4, 8)
    assert L.zor_do_offset_history(3, 0, C.byref(h)) == 0


def test_offset_history_table():
    """SURVEY A.6 / sequence_execution.rs:59-118"""
    L = oracle.lib()

    def run(of, ll, h):
        a = (C.c_uint32 * 3)(*h)
        r = L.zor_do_offset_history(of, ll, C.byref(a))
        return r, list(a)

    assert run(1, 5, [10, 20, 30]) == (10, [10, 20, 30])
    assert run(2, 5, [10, 20, 30]) == (20, [20, 10, 30])
    assert run(3, 5, [10, 20, 30]) == (30, [30, 10, 20])
    assert run(7, 5, [10, 20, 30]) == (4, [4, 10, 20])
    assert run(1, 0, [10, 20, 30]) == (20, [20, 10, 30])
    assert run(2, 0, [10, 20, 30]) == (30, [30, 10, 20])
    assert run(3, 0, [10, 20, 30]) == (9, [9, 10, 20])
    assert run(9, 0, [10, 20, 30]) == (6, [6, 10, 20])


def test_xxh64_known_answers():
    L = oracle.lib()
    assert L.zor_xxh64(b"", 0, 0) == 0xEF46DB3751D8E999
    assert L.zor_xxh64(b"a", 1, 0) == 0xD24EC4F1A98C6E5B
    assert L.zor_xxh64(b"abc", 3, 0) == 0x44BC2CF5AD770999
    s = b"Nobody inspects the spammish repetition"
    assert L.zor_xxh64(s, len(s), 0) == 0xFBCEA83C8A378BF1


def test_dict_parsing():
    """tests/dict_test.rs:2-75 shape: id / offset history / content split of the golden dictionary."""
    from golden_io import read_pack
    raw = read_pack("dict_tests.pack")["dictionary"]
    d = oracle.FrameDecoder()
    assert d.add_dict(raw) == 618557512
    assert raw[:4] == bytes([0x37, 0xA4, 0x30, 0xEC])
