"""The C ABI used from plain C (no Python in the data path): tests/capi_client.c is compiled with gcc against include/zgpu.h,
linked with libzgpu.so, and run on golden frames. It walks the binding INTEGRATION.md shows for the Rust side."""
import os
import subprocess
import sys

import pytest

from golden_io import read_manifest, read_pack

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "zstd-rs_amd")


def _build(tmp):
    exe = os.path.join(tmp, "capi_client")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "capi_client.c"), "-L" + PKG, "-lzgpu",
                           "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_c_client_on_golden_frames(tmp_path):
    import oracle
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import zgdata
    exe = _build(str(tmp_path))
    pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
    cases = [(n, pack[n]) for n in ("z000033.zst", "z000059.zst", "z000088.zst")]
    text = zgdata.text_like(3 << 20, seed=0xC)
    cases.append(("text3m.zst", zgdata.zstd_compress(text)))
    for name, z in cases:
        plain, _ = oracle.decode_frame_all(z)
        zp, pp = tmp_path / name, tmp_path / (name + ".plain")
        zp.write_bytes(z)
        pp.write_bytes(plain)
        r = subprocess.run([exe, str(zp), str(pp)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (name, r.stdout, r.stderr)
        assert "capi_client ok" in r.stdout
