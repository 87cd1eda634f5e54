#!/usr/bin/env python3
"""Differential soak on the GPU: random mutations (bit flips, byte / pair overwrites, truncations, block-header edits) of every frame
of the golden corpora through zgpu_decode_all against the oracle — the same verdict (bytes when both decode, the same error leaf when
both fail), under the engine's path switches. usage: soak.py [mutations per frame] [seed] [VAR=V,VAR=V ...]
Prints every disagreement; exit code 1 if there is one."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
per = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for kv in (sys.argv[3].split(",") if len(sys.argv) > 3 and sys.argv[3] else []):
    k, v = kv.split("="); os.environ[k] = v
import oracle, zgpu
from golden_io import read_pack
packs, syn = read_pack("decodecorpus.pack"), read_pack("synthetic.pack")
bases = [packs[n] for n in sorted(packs) if n.endswith(".zst")] + [syn[n] for n in sorted(syn) if n.endswith(".zst") and len(syn[n]) < (1 << 20)]
rng = random.Random(seed)
ctx = zgpu.Context(0, dev=True)
same_ok = same_err = 0
diffs = []
t0 = time.time()
for bi, base in enumerate(bases):
    for it in range(per):
        m = bytearray(base)
        for _ in range(1 + rng.randrange(2)):
            kind = rng.randrange(5)
            if len(m) < 16: break
            if kind == 0:
                i = rng.randrange(4, len(m)); m[i] ^= 1 << rng.randrange(8)
            elif kind == 1:
                i = rng.randrange(4, len(m)); m[i] = rng.randrange(256)
            elif kind == 2:
                m = m[:rng.randrange(8, len(m))]
            elif kind == 3 and len(m) > 12:
                i = rng.randrange(4, len(m) - 4); m[i:i + 2] = bytes([rng.randrange(256), rng.randrange(256)])
            else:                                     # early bytes: frame header, first block header, section headers
                i = rng.randrange(4, min(len(m), 40)); m[i] = rng.randrange(256)
        m = bytes(m)
        ost, oout = oracle.FrameDecoder().decode_all(m, 1 << 25)
        try:
            out, gst = ctx.decode_all(m, 1 << 25), 0
        except zgpu.ZgpuError as e:
            out, gst = None, e.status
        if ost == 0 and gst == 0:
            if out == oout: same_ok += 1
            else: diffs.append((bi, it, "bytes", len(out), len(oout)))
        elif ost == gst:
            same_err += 1
        else:
            diffs.append((bi, it, ost, gst))
            if len(diffs) <= 5:
                open(os.path.join(ROOT, "gpurun_out", "soak_diff_%d_%d_%d.zst" % (seed, bi, it)), "wb").write(m)
print("soak seed %d env %s: %d frames x %d mutations, both decode %d, same error %d, DISAGREE %d  (%.0f s)" % (seed, sys.argv[3] if len(sys.argv) > 3 else "", len(bases), per, same_ok, same_err, len(diffs), time.time() - t0))
for d in diffs[:40]: print("  ", d)
sys.exit(1 if diffs else 0)
