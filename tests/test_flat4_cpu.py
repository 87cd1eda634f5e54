"""The SOURCE of the direct-unit flatten (zstd-rs_amd/csrc/zg_flat4.h, the body zg_k_flatten runs on a frame's first unit) on the
CPU: tests/emu runs it through a small SIMT emulator (fibers for threads, real barriers and wave collectives) on the intermediates
of the CPU harness. Checked here, without a GPU: the plaintext the direct units produce equals the oracle's — all tile shapes (the
GPU's 1024 x 16 KiB among them), unit sizes from one block to whole frames, frames packed back to back at odd offsets (tiles start
at every alignment), raw / RLE blocks inside units, the reference's corpus. The same source is compiled for gfx950 into
libzgpu.so; the GPU tests repeat the comparison there (and cover the pointer-mode units, zg_flat1_unit + zg_k_sweep)."""
import ctypes as C
import hashlib
import os
import sys

import numpy as np
import pytest

import emu
import oracle
from golden_io import read_manifest, read_pack


def _lib():
    L = emu.lib()
    L.zgemu_decode3.restype = C.c_void_p
    L.zgemu_decode3.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.c_int, C.c_uint32, C.c_uint32]
    L.zgemu_flat4.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    return L


def run_flat4(z, unit_blocks, shape):
    """returns (status, plaintext of all frames, unit modes)"""
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
        modes = np.zeros(nu + 1, dtype=np.uint32)
        st = L.zgemu_flat4(h, shape, dst.ctypes.data, modes.ctypes.data)
        return st, dst[:total].tobytes(), modes[:nu]
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
@pytest.mark.parametrize("unit_blocks", [2, 256])
def test_plaintext_small_shape(name, unit_blocks):
    """unit_blocks 256: the whole frame is its own first unit"""
    z = syn()[name]
    st, got, modes = run_flat4(z, unit_blocks, 0)
    assert st == 0
    assert got == oracle_plain(z)
    assert hashlib.sha256(got).hexdigest() == read_manifest("synthetic.json")[name]["sha256"]
    if name.startswith("text"):
        assert (modes == 2).sum() == 1          # the frame's first unit went the direct way, and only that one
        if unit_blocks == 256:
            assert len(modes) == 1


@pytest.mark.parametrize("shape", [1, 2, 3, 4, 5, 6])
def test_plaintext_gpu_shapes(shape):
    """shape 3 = 1024 threads x 8 KiB tiles, one sequence per thread: what zg_k_flatten4 runs (two workgroups per CU); 2: round 4's shape"""
    for name in ("text_1m_l3.zst", "text_768k_l19.zst", "mixed_640k_l3.zst"):
        z = syn()[name]
        st, got, _ = run_flat4(z, 256, shape)
        assert st == 0 and got == oracle_plain(z), name


def test_frames_back_to_back_at_odd_offsets():
    """frames packed one after the other: direct units start at every alignment of the output"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import zgdata
    parts = [zgdata.text_like(40000 + 1237 * i + (i % 4), seed=77 + i) for i in range(9)]
    parts.insert(3, b"")                      # an empty frame in between
    parts.insert(5, bytes(1000))              # one RLE-ish frame
    parts.append(zgdata.text_like(300001, seed=99))      # several blocks, odd size
    z = b"".join(zgdata.zstd_compress(q) for q in parts)
    want = b"".join(parts)
    for shape in (0, 2, 3):
        st, got, modes = run_flat4(z, 256, shape)
        assert st == 0
        assert got == want
        assert (modes == 2).sum() >= 10


def test_reference_corpus():
    """the reference's decodecorpus files (tests/decode_corpus.rs): raw / RLE / compressed blocks of every kind in one direct unit"""
    pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
    names = sorted(n for n in pack if n.endswith(".zst"))[:50]
    for n in names:
        st, got, _ = run_flat4(pack[n], 256, 0)
        assert st == 0, n
        assert hashlib.sha256(got).hexdigest() == man[n]["sha256"], n
    # and all of them in one submit, back to back
    blob = b"".join(pack[n] for n in names)
    st, got, modes = run_flat4(blob, 0, 3)
    assert st == 0 and hashlib.sha256(got).digest() == hashlib.sha256(b"".join(oracle_plain(pack[n]) for n in names)).digest()
    st, got, modes = run_flat4(blob, 0, 2)
    assert st == 0 and len(got) == sum(man[n]["size"] for n in names)
    assert hashlib.sha256(got).digest() == hashlib.sha256(b"".join(oracle_plain(pack[n]) for n in names)).digest()
