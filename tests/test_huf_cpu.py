"""The SOURCE of zg_k_huf (zstd-rs_amd/csrc/zg_huf.h: one wave per Huffman stream, self-synchronising 128-bit chunks, the bit
window of a lane held in registers) on the CPU: tests/emu runs it through the SIMT emulator (fibers for threads, real wave
collectives) on the tables and headers the CPU harness produced. Checked without a GPU: the literals of every Huffman-coded
block equal the serial model's and the oracle's (literals_section_decoder.rs:40-158), on the synthetic frames, the reference's
corpus and streams of every shape (1 and 4 streams, code lengths from 1 to 11 bits, streams shorter than one lane's chunk,
chunks that overflow the symbol rows); with ZG_FLAG_LIT_DIRECT the blocks without sequences land in the output instead of the
arena; corrupted streams end with the serial model's verdict. The same source is compiled for gfx950 into libzgpu.so."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import emu
import oracle
from golden_io import read_pack

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

LIT_STATUSES = (30, 31, 32, 33, 34, 35)


def _lib():
    L = emu.lib()
    L.zgemu_huf.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    for f in ("zgemu_lit_bytes",):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = C.c_uint64
    for f in ("zgemu_block_lit_base", "zgemu_block_out_base"):
        getattr(L, f).argtypes = [C.c_void_p, C.c_uint32]
        getattr(L, f).restype = C.c_uint64
    return L


def run_huf(z, direct=False):
    """returns (EmuBatch, arena bytes, output bytes or None, lit_status[], lit_counts[])"""
    L = _lib()
    e = emu.EmuBatch(z)
    assert e.parse_status == 0
    nb = e.nblocks
    total = 0
    for f in range(e.nframes):
        b, s, _, _ = e.frame(f)
        total = max(total, b + s)
    lit = np.zeros(L.zgemu_lit_bytes(e.h) + 1, dtype=np.uint8)
    dst = np.zeros(total + 1, dtype=np.uint8)
    st = np.zeros(nb + 1, dtype=np.uint32)
    cnt = np.zeros(4 * nb + 4, dtype=np.uint32)
    assert L.zgemu_huf(e.h, 1 if direct else 0, lit.ctypes.data, dst.ctypes.data if direct else None, st.ctypes.data, cnt.ctypes.data) == 0
    return e, lit, (dst if direct else None), st[:nb], cnt[:4 * nb]


def check_against_model(z, direct=False):
    """every Huffman-coded block: literals == the serial model's (which the lane-logic tests pin on the oracle's); returns their number"""
    L = _lib()
    e, lit, dst, st, cnt = run_huf(z, direct)
    out = None
    if direct:
        p = L.zgemu_output(e.h)
    seen = 0
    for b in range(e.nblocks):
        k = e.block(b)
        if k["btype"] != 2 or k["lit_type"] < 2:
            continue
        if k["status"] in LIT_STATUSES or k["status"] == 36:
            continue
        n = k["regen_size"]
        want = e.block_literals(b, n)
        if direct and k["nseq"] == 0:
            if not k["active"]:
                continue
            ob = L.zgemu_block_out_base(e.h, b)
            got = dst[ob:ob + n].tobytes()
        else:
            lb = L.zgemu_block_lit_base(e.h, b)
            got = lit[lb:lb + n].tobytes()
        assert got == want, (b, k)
        assert st[b] == 0, (b, hex(int(st[b])))
        if k["nstreams"] == 4:
            seg = (n + 3) // 4
            assert list(cnt[4 * b:4 * b + 4]) == [min(seg, n), min(seg, max(n - seg, 0)), min(seg, max(n - 2 * seg, 0)), max(n - 3 * seg, 0)] or n < 4, (b, list(cnt[4 * b:4 * b + 4]), n)
        seen += 1
    return seen


@pytest.mark.parametrize("name", ["text_1m_l3.zst", "mixed_640k_l3.zst", "text_768k_l19.zst", "iso_512k_l3.zst", "text_1m_l1.zst"])
def test_synthetic_frames(name):
    z = read_pack("synthetic.pack")[name]
    assert check_against_model(z) > 0
    assert check_against_model(z, direct=True) > 0


def test_literals_equal_the_oracles():
    """the same literals, straight from the oracle (the last block's literal buffer after every single-block run)"""
    z = read_pack("synthetic.pack")["iso_512k_l3.zst"]
    L = _lib()
    e, lit, _, st, _ = run_huf(z)
    d = oracle.FrameDecoder()
    s, c, _, _ = d.init(z)
    assert s == 0
    pos, b, checked = c, 0, 0
    while True:
        s, used, fin = d.decode_blocks(z[pos:], oracle.STRAT_UPTO_BLOCKS, 1)
        assert s == 0
        pos += used
        k = e.block(b)
        if k["btype"] == 2 and k["lit_type"] >= 2:
            n = C.c_size_t()
            p = d.L.zor_last_literals(d.h, C.byref(n))
            want = C.string_at(p, n.value)
            lb = L.zgemu_block_lit_base(e.h, b)
            assert lit[lb:lb + n.value].tobytes() == want, b
            checked += 1
        b += 1
        if fin:
            break
    assert checked >= 3


def test_reference_corpus():
    pack = read_pack("decodecorpus.pack")
    names = sorted(n for n in pack if n.endswith(".zst"))
    seen = 0
    for n in names[:40]:
        seen += check_against_model(pack[n])
    assert seen > 100
    blob = b"".join(pack[n] for n in names[40:70])
    assert check_against_model(blob, direct=True) > 50


def _skewed(n, seed, spread):
    """bytes with a few very frequent values: short codes (down to one bit), many symbols per chunk"""
    rng = np.random.default_rng(seed)
    p = np.array([0.5 ** (i / spread + 1) for i in range(24)])
    p /= p.sum()
    return rng.choice(24, size=n, p=p).astype(np.uint8).tobytes()


def test_stream_shapes():
    import zgdata
    rng = np.random.default_rng(7)
    cases = []
    for n in (1, 2, 3, 5, 17, 63, 64, 65, 255, 256, 257, 1000, 1023, 1024, 1025, 4097, 70000, 131072):
        cases.append(_skewed(n, n, 3.0))                                  # mixed code lengths
    for n in (300, 5000, 131072):
        cases.append(_skewed(n, n + 1, 0.6))                              # dominated by a one- or two-bit code: chunks overflow the rows (dense mode)
        cases.append(bytes(rng.integers(0, 200, size=n, dtype=np.uint8)))  # nearly flat: long codes
        cases.append(zgdata.iso_like(n, seed=n))
    total = 0
    for data in cases:
        z = zgdata.zstd_compress(data)
        total += check_against_model(z)
        check_against_model(z, direct=True)
    assert total >= 12                                                     # (tiny inputs are stored raw: not every case has a Huffman block)


def test_corrupted_streams_end_with_the_models_verdict():
    """bytes flipped inside the literals section: whatever the serial model reports for the block (extra padding, stream does not
    end on its last bit, wrong symbol count — first failing stream first, literals_section_decoder.rs:98-155) is what the kernel's
    source reports"""
    z0 = read_pack("synthetic.pack")["iso_512k_l3.zst"]
    rng = np.random.default_rng(11)
    hits = 0
    for trial in range(24):
        z = bytearray(z0)
        at = int(rng.integers(200, len(z) - 200))
        z[at] ^= int(rng.integers(1, 256))
        z = bytes(z)
        e = emu.EmuBatch(z)
        if e.parse_status:
            continue
        _, lit, _, st, cnt = run_huf(z)
        for b in range(e.nblocks):
            k = e.block(b)
            if k["btype"] != 2 or k["lit_type"] < 2:
                continue
            if k["status"] in (31, 32, 33, 34, 35):
                # (a count mismatch whose four streams add up to the section's size is what zg_k_huf_uneven repairs: the model decodes it)
                assert int(st[b]) & 0xFF == k["status"], (trial, b, hex(int(st[b])), k["status"])
                hits += 1
            elif k["status"] == 0 and int(st[b]) & 0xFF == 35 and k["nstreams"] == 4:
                assert int(cnt[4 * b:4 * b + 4].sum()) == k["regen_size"], (trial, b)
            elif k["status"] == 0:
                assert st[b] == 0, (trial, b, hex(int(st[b])))
    assert hits >= 3


def test_two_failing_streams_the_earlier_one_decides():
    """a four-stream block with TWO defects: a stream that does not end on its last bit, and the all-zero last byte (ExtraPadding) of
    a LATER stream. The reference decodes the streams in order and checks a stream's padding when its turn comes
    (literals_section_decoder.rs:94-122): the earlier stream's BitstreamReadMismatch is the verdict. (Found while chasing a soak
    disagreement: the kernel ranked every ExtraPadding in front of all streams.)"""
    base = read_pack("synthetic.pack")["text_1m_l3.zst"]
    st0, c, _, _ = oracle.FrameDecoder().init(base)
    body = c + 3
    assert base[body] & 3 == 2 and (base[body] >> 2) & 3 == 3            # compressed literals, four streams, 5-byte header
    comp = (base[body + 2] >> 6) + (base[body + 3] << 2) + (base[body + 4] << 10)
    payload = body + 5
    hb = base[payload]
    desc = 1 + hb if hb < 128 else 1 + ((hb - 127) + 1) // 2
    jt = payload + desc
    j = [int.from_bytes(base[jt + 2 * i:jt + 2 * i + 2], "little") for i in range(3)]
    s0 = jt + 6
    starts = [s0, s0 + j[0], s0 + j[0] + j[1], s0 + j[0] + j[1] + j[2]]
    end3 = payload + comp
    cases = 0
    for k in (0, 1, 2):
        for off, val in ((0, 0xFF), (1, 0xFF), (1, 0x0F), (2, 0xFF)):
            m = bytearray(base)
            m[starts[k] + off] = val
            if oracle.FrameDecoder().decode_all(bytes(m), 1 << 25)[0] != 34:
                continue
            m[end3 - 1] = 0                                              # stream 3: ExtraPadding
            m = bytes(m)
            want = oracle.FrameDecoder().decode_all(m, 1 << 25)[0]
            assert want == 34
            _, _, _, st, _ = run_huf(m)
            assert int(st[0]) & 0xFF == want, (k, off, hex(int(st[0])))
            cases += 1
    assert cases >= 6
    # and the other way round: the padding defect in the EARLIER stream wins
    m = bytearray(base)
    m[starts[1] - 1] = 0                                                  # last byte of stream 0: ExtraPadding
    m[starts[2] + 1] = 0xFF                                               # stream 2 does not end on its last bit
    m = bytes(m)
    assert oracle.FrameDecoder().decode_all(m, 1 << 25)[0] == 33
    _, _, _, st, _ = run_huf(m)
    assert int(st[0]) & 0xFF == 33
