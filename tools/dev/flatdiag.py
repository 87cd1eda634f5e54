"""diagnostic: zg_k_flatten's scratch words against tests/lz_model.py, per frame of a multi-frame submit (ZGPU_DEBUG_NO_SWEEP), printing where they differ"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import zgdata, zgpu, lz_model
os.environ["ZGPU_DEBUG_NO_SWEEP"] = "1"

def run(name, plains, ub, direct="1"):
    os.environ["ZGPU_UNIT_BLOCKS"] = str(ub); os.environ["ZGPU_DIRECT"] = direct
    zs = [zgdata.zstd_compress(p) for p in plains]
    c = zgpu.Context(0, dev=True)
    b = c.prepare(b"".join(zs)); b.run(); b.sync()
    units = b.units()
    # frames: blocks per frame
    fb = 0; nbad = 0; ui = 0; base_out = 0
    for fi, (p, z) in enumerate(zip(plains, zs)):
        info = b.frame_info(fi)
        nb = info.nblocks
        mine = [u for u in units if fb <= u[0] < fb + nb]
        e, bounds = lz_model.expected_scratch(z, [u[0] - fb for u in mine])
        for k, (ufb, unb, sbase, size, noseq) in enumerate(mine):
            want = e[bounds[k]:bounds[k + 1]]
            if size != len(want):
                print(name, "frame", fi, "unit", k, "SIZE", size, len(want)); nbad += 1; continue
            if noseq & 1:
                continue
            if noseq & 2:
                got = b.read(base_out + bounds[k], size)
                if got != p[bounds[k]:bounds[k + 1]]:
                    g = np.frombuffer(got, dtype=np.uint8); w = np.frombuffer(p[bounds[k]:bounds[k + 1]], dtype=np.uint8)
                    bad = np.flatnonzero(g != w); nbad += 1
                    print(name, "frame", fi, "unit", k, "DIRECT bytes differ:", len(bad), "first", bad[:8], "tilepos", bad[:8] % 16384)
                continue
            got = b.scratch_words(sbase, size)
            bad = np.flatnonzero(got != want)
            if len(bad):
                nbad += 1
                print(name, "frame", fi, "unit", k, "scratch differs:", len(bad), "of", size, "first", bad[:6], "got", got[bad[:6]], "want", want[bad[:6]], "pos%16384", bad[:6] % 16384)
        fb += nb; base_out += len(p)
    print(name, "ub", ub, "direct", direct, "units", len(units), "BAD units" if nbad else "all units fine", nbad, flush=True)
    b.close(); c.close()

run("3x2MiB", [zgdata.text_like(2 << 20, seed=0xE9 + i) for i in range(3)], 4)
run("3x2MiB", [zgdata.text_like(2 << 20, seed=0xE9 + i) for i in range(3)], 4, "0")
run("one5MiB", [zgdata.text_like(5 << 20, seed=5)], 15)
run("16x8MiB", [zgdata.text_like(8 << 20, seed=0xE9 + i) for i in range(16)], 4)
run("one64MiB", [zgdata.text_like(64 << 20, seed=5)], 15)
