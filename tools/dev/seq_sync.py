#!/usr/bin/env python3
"""Can a block's sequence chain be cut? Huffman streams self-synchronise, which is what zg_k_huf lives on; this asks
the same of zstd's sequence bitstream: three interleaved FSE state chains (LL, ML, OF) + the codes' extra bits, read backwards
(sequence_section_decoder.rs:154-221). A SPECULATIVE decoder is started in the middle of a block's stream — in the best case for it: exactly
on a true sequence boundary (which a real one could not know), with states it has to guess — and runs until its (bit position, three states)
coincide with the true chain's at a sequence boundary; from there on the two are identical. Counted: sequences until that happens.
CPU only (tests/emu harness for the tables and the bitstream). usage: seq_sync.py [bytes of text] [starts per block]"""
import ctypes as C, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import emu, zgdata
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
starts = int(sys.argv[2]) if len(sys.argv) > 2 else 12
LLB = [0] * 16 + [1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]
MLB = [0] * 32 + [1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]
z = zgdata.zstd_compress(zgdata.text_like(n, seed=0xE9))
e = emu.EmuBatch(z)
L = e.L
L.zgemu_block_seq_bits.restype = C.POINTER(C.c_uint8)
L.zgemu_block_seq_bits.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
rng = random.Random(1)
results = []
for b in range(e.nblocks):
    info = e.block(b)
    if info["btype"] != 2 or info["nseq"] < 2000:
        continue
    ln = C.c_uint32()
    p = L.zgemu_block_seq_bits(e.h, b, C.byref(ln))
    bits = int.from_bytes(bytes(p[i] for i in range(ln.value)), "little")
    tabs = []
    for k, (slot, off) in enumerate(((info["ll_slot"], 0), (info["of_slot"], 1024), (info["ml_slot"], 512))):   # LL, OF, ML
        ent, logs = e.fse_slot(slot)
        lg = logs[k]
        tabs.append(([(ent[off + i] & 0xFFFF, (ent[off + i] >> 16) & 15, (ent[off + i] >> 20) & 63) for i in range(1 << lg)], lg))
    (tll, lll), (tof, lof), (tml, lml) = tabs
    top = bits.bit_length() - 1                      # the marker bit
    def rd(pos, nb):                                 # nb bits that END at position pos (exclusive), i.e. bits [pos - nb, pos)
        return (bits >> (pos - nb)) & ((1 << nb) - 1) if nb else 0
    def step(pos, sl, so, sm, last):
        bl_l, nb_l, sy_l = tll[sl]; bl_o, nb_o, sy_o = tof[so]; bl_m, nb_m, sy_m = tml[sm]
        pos -= sy_o + (MLB[sy_m] if sy_m < 53 else 0) + (LLB[sy_l] if sy_l < 36 else 0)        # extra bits: OF, ML, LL
        if not last:                                                                          # state updates: LL, ML, OF (:204-206)
            sl = bl_l + rd(pos, nb_l); pos -= nb_l
            sm = bl_m + rd(pos, nb_m); pos -= nb_m
            so = bl_o + rd(pos, nb_o); pos -= nb_o
        return pos, sl, so, sm
    pos = top
    sl = rd(pos, lll); pos -= lll
    so = rd(pos, lof); pos -= lof
    sm = rd(pos, lml); pos -= lml
    nseq = info["nseq"]
    traj = {}
    order = []
    for i in range(nseq):
        traj[pos] = (i, sl, so, sm)
        order.append((pos, sl, so, sm))
        pos, sl, so, sm = step(pos, sl, so, sm, i == nseq - 1)
    assert pos == 0, ("the model of the chain must end on bit 0", pos)
    for s in range(starts):
        i0 = rng.randrange(nseq // 8, nseq // 2)
        p0, tl, to, tm = order[i0]
        for guess in ("zeros", "random", "two of three right"):
            if guess == "zeros": gl = go = gm = 0
            elif guess == "random": gl, go, gm = rng.randrange(1 << lll), rng.randrange(1 << lof), rng.randrange(1 << lml)
            else: gl, go, gm = tl, to, rng.randrange(1 << lml)
            if (gl, go, gm) == (tl, to, tm):
                continue
            pos, sl, so, sm, k = p0, gl, go, gm, 0
            met = None
            while pos > 64 and k < nseq:
                t = traj.get(pos)
                if t is not None and t[1:] == (sl, so, sm):
                    met = k; break
                pos, sl, so, sm = step(pos, sl, so, sm, False)
                k += 1
            results.append((guess, nseq - i0, met))
for guess in ("zeros", "random", "two of three right"):
    r = [x for x in results if x[0] == guess]
    met = sorted(x[2] for x in r if x[2] is not None)
    print("%-20s %4d starts on a true boundary: fell onto the true chain %4d times%s; never within the rest of the block (%d sequences on average): %d"
          % (guess, len(r), len(met), (" (median after %d sequences, max %d)" % (met[len(met) // 2], met[-1])) if met else "", sum(x[1] for x in r) // max(len(r), 1), len(r) - len(met)))
