#!/usr/bin/env python3
"""The differential soak of soak.py without a GPU: the engine's host parser, verdict rules and zg_k_exact (CPU emulator harness,
tests/emu) against the oracle on randomly mutated frames. The harness models the entropy kernels serially, so this finds
disagreements in what is shared with the product: the host walk, the order in which verdicts outrank each other, zg_exact.h.
usage: soak_cpu.py [mutations per frame] [seed] [concat]   (concat: every input is two corpus frames back to back)"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import emu, oracle
from golden_io import read_pack
per = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
concat = len(sys.argv) > 3 and sys.argv[3] == "concat"
packs, syn = read_pack("decodecorpus.pack"), read_pack("synthetic.pack")
bases = [packs[n] for n in sorted(packs) if n.endswith(".zst")] + [syn[n] for n in sorted(syn) if n.endswith(".zst") and len(syn[n]) < (1 << 20)]
rng = random.Random(seed)


def block_starts(z):
    """offsets of the block headers of a (single, valid) frame"""
    st, c, _, _ = oracle.FrameDecoder().init(z)
    out, p = [], c
    while st == 0 and p + 3 <= len(z):
        h = int.from_bytes(z[p:p + 3], "little")
        out.append(p)
        size = 1 if ((h >> 1) & 3) == 1 else h >> 3
        p += 3 + size
        if h & 1:
            break
    return out


starts = [block_starts(b) for b in bases]


verdict = emu.decode_all_verdict


same_ok = same_err = 0
diffs = []
t0 = time.time()
for bi, base in enumerate(bases):
    for it in range(per):
        m = bytearray(base + (bases[rng.randrange(len(bases))] if concat else b""))
        for _ in range(1 + rng.randrange(3)):
            if len(m) < 16:
                break
            kind = rng.randrange(7)
            if kind == 0:
                i = rng.randrange(4, len(m)); m[i] ^= 1 << rng.randrange(8)
            elif kind == 1:
                i = rng.randrange(4, len(m)); m[i] = rng.randrange(256)
            elif kind == 2:
                m = m[:rng.randrange(8, len(m))]
            elif kind == 3:
                i = rng.randrange(4, len(m) - 4); m[i:i + 2] = bytes([rng.randrange(256), rng.randrange(256)])
            elif kind == 4:
                i = rng.randrange(4, min(len(m), 40)); m[i] = rng.randrange(256)
            else:                                     # the first bytes of a block: its header, its literals / sequences section headers, table descriptions
                i = starts[bi][rng.randrange(len(starts[bi]))] + rng.randrange(0, 24)
                if i < len(m):
                    m[i] = rng.randrange(256) if kind == 5 else m[i] ^ (1 << rng.randrange(8))
        m = bytes(m)
        ost, _ = oracle.FrameDecoder().decode_all(m, 1 << 25)
        gst = verdict(m)
        if ost == gst:
            if ost: same_err += 1
            else: same_ok += 1
        else:
            diffs.append((bi, it, ost, gst))
            if len(diffs) <= 8:
                os.makedirs("/tmp/soak", exist_ok=True)
                open("/tmp/soak/cpu_%d_%d_%d.zst" % (seed, bi, it), "wb").write(m)
print("cpu soak seed %d: %d frames x %d mutations, both decode %d, same error %d, DISAGREE %d  (%.0f s)" % (seed, len(bases), per, same_ok, same_err, len(diffs), time.time() - t0))
for d in diffs[:40]: print("  ", d)
