"""The SOURCE of the pointer-mode flatten (zstd-rs_amd/csrc/zg_flat1.h, the body zg_k_flatten runs on every unit that may copy from
in front of itself) on the CPU: tests/emu runs it through the SIMT emulator on the intermediates of the CPU harness, next to the
direct body (zg_flat4.h) for the frames' first units, and a plain model of zg_k_sweep resolves the effective offsets. Checked
here, without a GPU: the scratch words of every pointer-mode unit equal the numpy model of tests/lz_model.py (effective offsets
from the oracle's sequences), and the plaintext after the sweep equals the oracle's — tile shapes up to the GPU's 1024 x 16 KiB,
units from two blocks upwards, frames packed back to back at odd offsets, raw / RLE blocks inside units, the reference's corpus."""
import ctypes as C
import hashlib
import os
import sys

import numpy as np
import pytest

import emu
import lz_model
from golden_io import read_manifest, read_pack
from test_flat4_cpu import oracle_plain

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def _lib():
    L = emu.lib()
    L.zgemu_decode3.restype = C.c_void_p
    L.zgemu_decode3.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.c_int, C.c_uint32, C.c_uint32]
    L.zgemu_flatten.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


def run_flatten(z, unit_blocks, shape):
    """returns (status, plaintext of all frames after the sweep model, scratch words by output position, units [(frame, first_block, nblocks, mode)])"""
    L = _lib()
    h = L.zgemu_decode3(z, len(z), 1 << 31, 1, unit_blocks, 0)
    try:
        assert L.zgemu_parse_status(h) == 0
        total = 0
        b, s, st, bb = C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_uint32()
        for f in range(L.zgemu_num_frames(h)):
            L.zgemu_frame(h, f, C.byref(b), C.byref(s), C.byref(st), C.byref(bb))
            total = max(total, b.value + s.value)
        nu = L.zgemu_num_units(h)
        dst = np.zeros(total + 1, dtype=np.uint8)
        og = np.zeros(total + 1, dtype=np.uint32)
        modes = np.zeros(nu + 1, dtype=np.uint32)
        st = L.zgemu_flatten(h, shape, dst.ctypes.data, og.ctypes.data, modes.ctypes.data)
        u4 = (C.c_uint32 * 4)()
        f7 = (C.c_uint32 * 7)()
        L.zgemu_frame_plan.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32 * 7)]
        run_flatten.sparse = []
        for f in range(L.zgemu_num_frames(h)):
            L.zgemu_frame_plan(h, f, C.byref(f7))
            run_flatten.sparse.append(int(f7[6]))
        units = []
        for u in range(nu):
            L.zgemu_unit(h, u, C.byref(u4))
            units.append(tuple(u4))
        return st, dst[:total].tobytes(), og[:total], units
    finally:
        L.zgemu_free(h)


def check_scratch(z, unit_blocks, shape):
    """one frame: plaintext == oracle, and the scratch of every pointer-mode unit == the numpy model"""
    st, got, og, units = run_flatten(z, unit_blocks, shape)
    assert st == 0
    assert got == oracle_plain(z)
    want, bounds = lz_model.expected_scratch(z, [u[1] for u in units])
    npointer = 0
    for i, (_, _, _, mode) in enumerate(units):
        if mode != 0:
            continue                         # direct units and units without sequences leave no scratch words
        a, b = bounds[i], bounds[i + 1]
        assert np.array_equal(og[a:b], want[a:b]), (i, int(np.flatnonzero(og[a:b] != want[a:b])[0]))
        npointer += 1
    return npointer


@pytest.mark.parametrize("name", ["text_1m_l3.zst", "mixed_640k_l3.zst", "text_768k_l19.zst", "text_1m_l1.zst"])
def test_scratch_and_plaintext_small_shape(name):
    z = read_pack("synthetic.pack")[name]
    assert check_scratch(z, 2, 0) >= 2
    assert hashlib.sha256(oracle_plain(z)).hexdigest() == read_manifest("synthetic.json")[name]["sha256"]


@pytest.mark.parametrize("shape,unit_blocks", [(1, 3), (2, 2), (2, 4)])
def test_scratch_and_plaintext_gpu_shapes(shape, unit_blocks):
    assert check_scratch(read_pack("synthetic.pack")["text_1m_l3.zst"], unit_blocks, shape) >= 1


def test_no_direct_units_every_unit_through_the_scratch():
    """a frame that continues an earlier submit has no direct unit; here: the iso-like frame (few sequences: sparse) and a frame of raw / RLE blocks"""
    z = read_pack("synthetic.pack")["iso_512k_l3.zst"]
    st, got, og, units = run_flatten(z, 2, 0)
    assert st == 0 and got == oracle_plain(z)


@pytest.mark.parametrize("shape", [0, 2])
def test_sparse_frames_literal_runs_placed_without_tiles(shape):
    """frames whose few matches zg_k_sparse copies in order: zg_flat1_unit checks the offsets and copies the literal runs, the whole
    workgroup per run (a sequence or two per block) or a wave per run (one block of many sequences in a frame that is sparse on average)"""
    import zgdata
    rng = np.random.default_rng(77)
    noise = rng.integers(0, 256, 30 * 131072, dtype=np.uint8).tobytes()
    few = bytearray(zgdata.iso_like(700001, seed=5))
    many = noise + zgdata.text_like(1500, seed=9)                # the last block: tens of short matches
    rle_lit = noise[:131072] + b"ab" * 5 + b"\x07" * 70000 + noise[:40]   # a block with RLE or near-RLE literals and a match
    z = zgdata.zstd_compress(bytes(few)) + zgdata.zstd_compress(many) + zgdata.zstd_compress(rle_lit)
    st, got, og, units = run_flatten(z, 4, shape)
    assert st == 0
    assert run_flatten.sparse[:2] == [1, 1], run_flatten.sparse
    assert got == bytes(few) + many + rle_lit


def test_sparse_frames_offsets_are_checked_without_tiles():
    """hand-made frames of raw blocks and one-sequence blocks (sparse by construction): an offset that reaches in front of the frame is
    found by the run-by-run path of zg_flat1_unit with the oracle's verdict, one that does not decodes to the oracle's bytes"""
    import oracle
    import test_exact_cpu as X
    good = X.frame(X.raw_block(3000), X.seq_block(100), X.raw_block(500, seed=3), X.seq_block(3000, last=True))
    bad = X.frame(X.raw_block(3000), X.seq_block(100), X.raw_block(500, seed=3), X.seq_block(70000, last=True))
    zero = X.frame(X.raw_block(3000), X.seq_block(100), X.seq_block(3604 + 7, last=True))          # exactly one byte too far
    for shape in (0, 2):
        st, got, og, units = run_flatten(good, 4, shape)
        assert run_flatten.sparse == [1] and st == 0 and got == oracle.FrameDecoder().decode_all(good, 1 << 26)[1]
        for z in (bad, zero):
            st, got, og, units = run_flatten(z, 4, shape)
            assert run_flatten.sparse == [1] and st != 0 and st == X.oracle_all(z), (st, X.oracle_all(z))


def test_frames_back_to_back_at_odd_offsets():
    import zgdata
    parts = [zgdata.text_like(300001 + 1237 * i + (i % 4), seed=177 + i) for i in range(4)]
    parts.insert(2, b"")
    parts.insert(3, bytes(1000))
    z = b"".join(zgdata.zstd_compress(q) for q in parts)
    want = b"".join(parts)
    for shape in (0, 2):
        st, got, og, units = run_flatten(z, 1, shape)          # one block per unit: every frame has pointer-mode units behind its first
        assert st == 0
        assert got == want
        assert sum(1 for u in units if u[3] == 0) >= 4


def test_reference_corpus():
    """the reference's decodecorpus files (tests/decode_corpus.rs) with one block per unit: raw / RLE / compressed blocks of every kind"""
    pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
    names = sorted(n for n in pack if n.endswith(".zst"))[:40]
    for n in names:
        st, got, og, units = run_flatten(pack[n], 1, 0)
        assert st == 0, n
        assert hashlib.sha256(got).hexdigest() == man[n]["sha256"], n
