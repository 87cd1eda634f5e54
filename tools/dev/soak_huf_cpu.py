#!/usr/bin/env python3
"""zg_k_huf's source (zg_huf.h, CPU emulator) against the serial literal model of the harness on randomly mutated frames: the literal
verdict of every Huffman block (extra padding, bitstream mismatch, count mismatch: which, and of which stream first).
usage: soak_huf_cpu.py [mutations per frame] [seed]"""
import ctypes as C, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import emu
from golden_io import read_pack
per = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L = emu.lib()
L.zgemu_huf.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4
L.zgemu_lit_bytes.restype = C.c_uint64; L.zgemu_lit_bytes.argtypes = [C.c_void_p]
packs, syn = read_pack("decodecorpus.pack"), read_pack("synthetic.pack")
bases = [packs[n] for n in sorted(packs) if n.endswith(".zst")] + [syn[n] for n in sorted(syn) if n.endswith(".zst") and len(syn[n]) < (1 << 20)]
rng = random.Random(seed)
checked = bad = 0
t0 = time.time()
for bi, base in enumerate(bases):
    for it in range(per):
        m = bytearray(base)
        for _ in range(1 + rng.randrange(3)):
            i = rng.randrange(6, len(m))
            k = rng.randrange(3)
            if k == 0: m[i] ^= 1 << rng.randrange(8)
            elif k == 1: m[i] = rng.randrange(256)
            else: m[i] = 0
        m = bytes(m)
        e = emu.EmuBatch(m)
        if e.nblocks == 0:
            continue
        nb = e.nblocks
        lit = np.zeros(L.zgemu_lit_bytes(e.h) + 64, dtype=np.uint8); st = np.zeros(nb + 1, dtype=np.uint32); cnt = np.zeros(4 * nb + 4, dtype=np.uint32)
        L.zgemu_huf(e.h, 0, lit.ctypes.data, None, st.ctypes.data, cnt.ctypes.data)
        for b in range(nb):
            k = e.block(b)
            if k["btype"] != 2 or k["lit_type"] < 2:
                continue
            model, kern = k["status"], int(st[b]) & 0xFF
            if model in (31, 32, 33, 34, 35):
                checked += 1
                if kern != model and not (model == 35 and kern == 0):
                    bad += 1
                    if bad <= 20: print("DISAGREE", bi, it, b, "model", model, "kernel", hex(int(st[b])), "streams", k["nstreams"])
            elif model == 0 and kern not in (0, 35):
                bad += 1
                if bad <= 20: print("DISAGREE", bi, it, b, "model 0 kernel", hex(int(st[b])))
print("huf soak seed %d: %d literal verdicts checked, DISAGREE %d (%.0f s)" % (seed, checked, bad, time.time() - t0))
