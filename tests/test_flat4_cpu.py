"""The flatten kernel's SOURCE (zstd-rs_amd/csrc/zg_flat4.h, the body of zg_k_flat4) on the CPU: tests/emu runs it through a small SIMT
emulator (fibers for threads, real barriers and wave collectives) on the intermediates of the CPU harness, then applies a
serial statement of the sweep. Checked here, without a GPU:
  * the plaintext equals the oracle's, in pointer mode (every unit through scratch + sweep) and with direct first units;
  * the scratch words of pointer-mode units equal the numpy model built from the oracle's sequences (tests/lz_model.py);
  * all tile shapes (the GPU's 1024 x 16 KiB among them), unit sizes that make tiles start at every alignment, frames packed
    back to back at odd offsets, raw / RLE blocks inside units, the reference's corpus.
The same source is compiled for gfx950 into libzgpu.so; the GPU tests repeat the comparison there."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import emu
import lz_model
import oracle
from golden_io import read_manifest, read_pack


def _lib():
    L = emu.lib()
    L.zgemu_decode3.restype = C.c_void_p
    L.zgemu_decode3.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.c_int, C.c_uint32, C.c_uint32]
    L.zgemu_flat4.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


def run_flat4(z, unit_blocks, shape, force_pointer, want_og=False):
    """returns (status, plaintext of all frames, scratch words or None, unit modes)"""
    L = _lib()
    h = L.zgemu_decode3(z, len(z), 1 << 31, 1, unit_blocks, 0)
    try:
        assert L.zgemu_parse_status(h) == 0
        nf = L.zgemu_num_frames(h)
        total = 0
        b, s, st, bb = C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_uint32()
        for f in range(nf):
            L.zgemu_frame(h, f, C.byref(b), C.byref(s), C.byref(st), C.byref(bb))
            total = max(total, b.value + s.value)
        nu = L.zgemu_num_units(h)
        dst = np.zeros(total + 1, dtype=np.uint8)
        og = np.zeros(total + 1, dtype=np.uint32) if want_og else None
        modes = np.zeros(nu + 1, dtype=np.uint32)
        st = L.zgemu_flat4(h, shape, 1 if force_pointer else 0, dst.ctypes.data, og.ctypes.data if want_og else None, modes.ctypes.data)
        return st, dst[:total].tobytes(), (og[:total] if want_og else None), modes[:nu]
    finally:
        L.zgemu_free(h)


def oracle_plain(z):
    out = b""
    pos = 0
    while pos < len(z):
        d = oracle.FrameDecoder()
        st, c, _, _ = d.init(z[pos:])
        assert st == 0
        st, c2, fin = d.decode_blocks(z[pos + c:], oracle.STRAT_ALL)
        assert st == 0 and fin
        out += d.collect()
        pos += c + c2 + (4 if d.checksum_from_data() is not None else 0)
    return out


SYN = None


def syn():
    global SYN
    if SYN is None:
        SYN = read_pack("synthetic.pack")
    return SYN


@pytest.mark.parametrize("name", ["text_1m_l3.zst", "mixed_640k_l3.zst", "text_768k_l19.zst", "iso_512k_l3.zst", "text_1m_l1.zst"])
@pytest.mark.parametrize("force_pointer", [True, False])
def test_plaintext_small_shape(name, force_pointer):
    z = syn()[name]
    want = oracle_plain(z)
    st, got, _, modes = run_flat4(z, 2, 0, force_pointer)
    assert st == 0
    assert got == want
    assert hashlib.sha256(got).hexdigest() == read_manifest("synthetic.json")[name]["sha256"]
    if not force_pointer and name.startswith("text"):
        assert (modes == 2).sum() >= 1          # the frame's first unit went the direct way


@pytest.mark.parametrize("shape", [1, 2])
def test_plaintext_gpu_shapes(shape):
    z = syn()["text_1m_l3.zst"]
    want = oracle_plain(z)
    for fp in (True, False):
        st, got, _, _ = run_flat4(z, 3, shape, fp)
        assert st == 0 and got == want


@pytest.mark.parametrize("unit_blocks", [1, 2, 3])
def test_scratch_matches_model(unit_blocks):
    """every scratch word of the pointer-mode units against the numpy model of the effective offsets"""
    z = syn()["text_1m_l3.zst"]
    st, got, og, modes = run_flat4(z, unit_blocks, 0, True, want_og=True)
    assert st == 0
    p = emu.Plan(z, unit_blocks=unit_blocks)
    firsts = [fb for (_, fb, _, _) in p.units]
    want, bounds = lz_model.expected_scratch(z, firsts)
    assert len(want) == len(og)
    bad = np.flatnonzero(want != og)
    assert bad.size == 0, (bad[:10], want[bad[:10]], og[bad[:10]])


def test_frames_back_to_back_at_odd_offsets():
    """frames packed one after the other: units start at every alignment of the scratch and of the output"""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import zgdata
    parts = [zgdata.text_like(40000 + 1237 * i + (i % 4), seed=77 + i) for i in range(9)]
    parts.insert(3, b"")                      # an empty frame in between
    parts.insert(5, bytes(1000))              # one RLE-ish frame
    z = b"".join(zgdata.zstd_compress(q) for q in parts)
    want = b"".join(parts)
    for fp in (True, False):
        st, got, _, _ = run_flat4(z, 1, 0, fp)
        assert st == 0
        assert got == want


def test_reference_corpus_pointer_and_direct():
    """the reference's decodecorpus files (tests/decode_corpus.rs): raw / RLE / compressed blocks of every kind in one unit"""
    pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
    names = sorted(n for n in pack if n.endswith(".zst"))[:40]
    for n in names:
        z = pack[n]
        for fp in (True, False):
            st, got, _, _ = run_flat4(z, 2, 0, fp)
            assert st == 0, n
            assert hashlib.sha256(got).hexdigest() == man[n]["sha256"], (n, fp)
