#!/usr/bin/env python3
"""zg_exact.h (CPU emulator) against the oracle on random HAND-MADE frames: raw / RLE / literal-only blocks that move the DecodeBuffer's
length and total_output_counter apart, and one- or several-sequence blocks whose offsets sit around the interesting borders (what the
frame has produced, the window, what decode_all's drains left, one byte either side). Verdict of decode_all (drain rule: every MiB).
usage: soak_exact_cpu.py [frames] [seed] [dict]   (dict: frames with a dictionary, decoded in runs with drains between them)"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import emu, oracle
import test_exact_cpu as X
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
W = 1 << X.WINDOW_LOG


def dict_mode():
    """(round 5) frames WITH a dictionary, decoded run by run through FrameDecoder::decode_blocks(UptoBlocks) with a collect() after some
    of the runs (which drains the buffer down to the window): offsets around what is left, what was drained and the dictionary's length —
    among them the splice of the dictionary's tail behind drained bytes (decode_buffer.rs:159-163). zg_exact.h gets one submit per run with
    the carried state (prior_out / prior_reach / prior_counted), as the engine's FrameDecoder mirror hands it over."""
    from golden_io import read_pack
    raw = read_pack("dict_tests.pack")["dictionary"]
    did = 618557512
    probe = lambda n: X.oracle_blocks(X.frame(X.seq_block(n, lits=b"", last=True)), raw, did)
    lo, hi = 1, len(raw)
    while lo < hi:
        mid = (lo + hi + 1) // 2
        lo, hi = (mid, hi) if probe(mid) == 0 else (lo, mid - 1)
    dict_len = lo
    same = collections.Counter()
    diffs = []
    for it in range(n_frames):
        runs, produced = [], 0
        for r in range(rng.randrange(1, 5)):
            blocks = []
            for k in range(rng.randrange(1, 5)):
                kind = rng.randrange(5)
                if kind == 0:
                    n = rng.choice([1, 4096, 65536, 131072]); blocks.append(("raw", n)); produced += n
                elif kind == 1:
                    n = rng.randrange(1, 4000); blocks.append(("lit", n)); produced += n
                    if rng.random() < 0.15:                   # a window's worth of COUNTED bytes: the dictionary is closed behind them (OffsetTooBig)
                        blocks += [("lit", 3999)] * 34; produced += 34 * 3999
                else:
                    held = min(produced, W) if runs else produced        # (after a collect the buffer holds at most the window)
                    base = rng.choice([held, held + 1, held + 2, held + 3, held + 50, held + dict_len, held + dict_len + 1, produced, max(1, held - 5), W, rng.randrange(1, max(2, held + dict_len + 10))])
                    blocks.append(("seq", max(1, min(base, 1 << 27)))); produced += 7
            runs.append((blocks, rng.random() < 0.6))
        flat = [b for blocks, _ in runs for b in blocks]
        def enc(b, last):
            return X.raw_block(b[1], seed=b[1] & 255, last=last) if b[0] == "raw" else X.lit_block(b[1], last=last) if b[0] == "lit" else X.seq_block(b[1], last=last)
        z = X.frame(*[enc(b, i == len(flat) - 1) for i, b in enumerate(flat)])
        o = oracle.FrameDecoder()
        assert o.add_dict(raw) == did
        st, c, _, _ = o.init(z)
        assert st == 0 and o.force_dict(did) == 0
        pos = c
        prior_out = prior_reach = prior_counted = 0
        verdict_o = verdict_e = 0
        nblk = 0
        for ri, (blocks, drain) in enumerate(runs):
            last_run = ri == len(runs) - 1
            ost, used, fin = o.decode_blocks(z[pos:], oracle.STRAT_ALL if last_run else oracle.STRAT_UPTO_BLOCKS, len(blocks))
            sub = X.frame(*[enc(b, last_run and i == len(blocks) - 1) for i, b in enumerate(blocks)])
            e = emu.EmuBatch(sub)
            est, _, counted = e.exact(0, dict_len=dict_len, prior_out=prior_out, prior_reach=prior_reach, prior_counted=prior_counted)[0]
            # (the harness's serial model knows nothing of the bytes in front of a run: its own frame status is not consulted here; the
            #  hand-made blocks can only fail in execution, which is zg_exact.h's verdict)
            if ost or est:
                verdict_o, verdict_e = ost, est
                break
            size = e.frame(0)[1]
            prior_out += size; prior_reach += size; prior_counted += counted
            pos += used
            if drain and not last_run:
                got = o.collect()
                prior_reach -= len(got)
        same[(verdict_o, verdict_e)] += 1
        if verdict_o != verdict_e:
            diffs.append((it, verdict_o, verdict_e))
    print("exact soak (dictionary, runs, drains) seed %d: %d frames, verdict pairs (oracle, zg_exact.h) %s, DISAGREE %d" % (seed, n_frames, dict(same), len(diffs)))
    for d in diffs[:20]: print("  ", d)
    sys.exit(1 if diffs else 0)


if len(sys.argv) > 3 and sys.argv[3] == "dict":
    import collections
    dict_mode()
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
