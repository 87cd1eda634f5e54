#!/usr/bin/env python3
"""zg_exact.h (CPU emulator) against the oracle on random HAND-MADE frames: raw / RLE / literal-only blocks that move the DecodeBuffer's
length and total_output_counter apart, and one- or several-sequence blocks whose offsets sit around the interesting borders (what the
frame has produced, the window, what decode_all's drains left, one byte either side). Verdict of decode_all (drain rule: every MiB).
usage: soak_exact_cpu.py [frames] [seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import emu, oracle
import test_exact_cpu as X
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
W = 1 << X.WINDOW_LOG
same_ok = same_err = 0
diffs = []
t0 = time.time()
for it in range(n_frames):
    blocks, produced = [], 0
    nb = rng.randrange(2, 24)
    for k in range(nb):
        last = k == nb - 1
        kind = rng.randrange(6)
        if kind == 0:
            n = rng.choice([1, 100, 4096, 65536, 131072, rng.randrange(1, 131073)])
            blocks.append(X.raw_block(n, seed=rng.randrange(256), last=last)); produced += n
        elif kind == 1:
            n = rng.choice([1, 300, 131072, rng.randrange(1, 131073)])
            blocks.append(X.rle_block(n, byte=rng.randrange(256), last=last)); produced += n
        elif kind == 2:
            n = rng.randrange(1, 4000)
            blocks.append(X.lit_block(n, last=last)); produced += n
        else:
            # offsets around the borders: what exists, the window, a MiB round's leftovers
            borders = [produced + 4, W, W + 4, produced + 4 - W if produced + 4 > W else 1, (produced + 4) % (1 << 20) + W, 1, 2, 3, 7]
            def pick():
                if rng.random() < 0.65:                       # mostly executable offsets, so that frames live long enough to cross decode_all's drain points
                    top = max(1, min(produced + 4, W))
                    return rng.choice([top, max(1, top - 1), max(1, top - rng.randrange(0, 64)), rng.randrange(1, top + 1)])
                b = rng.choice(borders) + rng.choice([-2, -1, 0, 0, 1, 2, 5])
                return max(1, min(b, (1 << 28)))
            if kind == 3:
                blocks.append(X.seq_block(pick(), last=last)); produced += 7
            else:
                base = pick()
                code = (base + 3).bit_length() - 1
                lo, hi = (1 << code) - 3, (1 << (code + 1)) - 4
                offs = [min(max(base + rng.randrange(-3, 4), max(lo, 1)), hi) for _ in range(rng.randrange(2, 6))]
                blocks.append(X.multi_seq_block(offs, last=last)); produced += 4 + 3 * len(offs)
    z = X.frame(*blocks)
    ost, _ = oracle.FrameDecoder().decode_all(z, 1 << 26)
    gst = emu.decode_all_verdict(z)
    if ost == gst:
        if ost: same_err += 1
        else: same_ok += 1
    else:
        diffs.append((it, ost, gst))
        if len(diffs) <= 5:
            os.makedirs("/tmp/soak", exist_ok=True)
            open("/tmp/soak/exact_%d_%d.zst" % (seed, it), "wb").write(z)
print("exact soak seed %d: %d frames, both decode %d, same error %d, DISAGREE %d (%.0f s)" % (seed, n_frames, same_ok, same_err, len(diffs), time.time() - t0))
for d in diffs[:30]: print("  ", d)
