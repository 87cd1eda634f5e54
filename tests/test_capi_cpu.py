"""CPU-side checks of the product: the C-ABI library loads and exports every symbol include/zgpu.h declares, it
refuses to run without a GPU (no CPU fallback), and nothing under oracle/ is linked into it."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "zstd-rs_amd", "libzgpu.so")
HDR = os.path.join(ROOT, "include", "zgpu.h")


def _built():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    return LIB


def test_header_symbols_exported():
    _built()
    hdr = open(HDR).read()
    declared = set(re.findall(r"\b(zgpu_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"zgpu_status"}
    import zgpu
    L = zgpu.load_library()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert set(zgpu.EXPORTS) <= declared


def test_no_oracle_in_product():
    _built()
    syms = subprocess.check_output(["nm", "-D", LIB]).decode()
    assert "zor_" not in syms
    for f in os.listdir(os.path.join(ROOT, "zstd-rs_amd", "csrc")):
        if not f.endswith((".h", ".cpp", ".hip", "Makefile")):
            continue
        src = open(os.path.join(ROOT, "zstd-rs_amd", "csrc", f), errors="ignore").read()
        assert "zstd_oracle" not in src and "zor_" not in src, f


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import zgpu
    with pytest.raises(zgpu.ZgpuError) as e:
        zgpu.Context()
    assert e.value.status == zgpu.E_HIP


def test_product_library_has_no_measurement_switches():
    """an environment variable must not be able to make the drop-in decoder slower, let alone wrong. The product library
    (libzgpu.so) reads no ZGPU_* variable and calls getenv nowhere; the switches — timing modes that return wrong bytes, the ramped sweep
    chain, unit sizes — exist only in libzgpu_dev.so (-DZG_DEV_SWITCHES), which the tests that force a path and tools/dev load."""
    _built()
    dev = os.path.join(ROOT, "zstd-rs_amd", "libzgpu_dev.so")
    assert os.path.exists(dev)
    names = ("ZGPU_FLAT_MODE", "ZGPU_SWEEP_MODE", "ZGPU_DEBUG_NO_SWEEP", "ZGPU_DEBUG_NO_EXACT", "ZGPU_RAMP", "ZGPU_UNIT_BLOCKS", "ZGPU_SEQ_PACKED",
             "ZGPU_FORCE_INORDER", "ZGPU_POOL_JOBS", "ZGPU_DA_SPLIT")
    rel = open(LIB, "rb").read()
    devb = open(dev, "rb").read()
    for n in names:
        assert n.encode() not in rel, n
        assert n.encode() in devb, n
    assert b"ZGPU_" not in rel.replace(b"ZGPU_E_", b"")          # no other switch either
    und = subprocess.check_output(["nm", "-D", "--undefined-only", LIB]).decode()
    assert "getenv" not in und
    assert "getenv" in subprocess.check_output(["nm", "-D", "--undefined-only", dev]).decode()
    # the timing variants of the sweep kernel are not even compiled into the product
    assert b"zg_k_sweepILi5" not in rel and b"zg_k_sweepILi5" in devb
