#!/usr/bin/env python3
"""What the sequences of a workload look like (CPU, through the oracle): share of repeat-offset codes, match / literal lengths,
how far offsets reach. Compares the synthetic stand-in (text_like) with the real text found in this image (tools/realtext.py):
seq_stats.py [bytes of each sample, default 64 MiB]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import oracle, zgdata, realtext


def stats(name, plain):
    z = zgdata.zstd_compress(plain, level=3)
    o = oracle.FrameDecoder()
    st, c, _, _ = o.init(z)
    assert st == 0
    pos, nseq, rep, ml_sum, ll_sum, blocks = c, 0, 0, 0, 0, 0
    offs = []
    while True:
        st, used, fin = o.decode_blocks(z[pos:], oracle.STRAT_UPTO_BLOCKS, 1)
        assert st == 0
        pos += used
        blocks += 1
        sq = o.last_sequences()            # [(ll, ml, of)] of the block just decoded (of: the raw offset value, 1..3 = repeat codes)
        if sq:
            a = np.array(sq, dtype=np.int64)
            nseq += len(a); rep += int((a[:, 2] <= 3).sum()); ml_sum += int(a[:, 1].sum()); ll_sum += int(a[:, 0].sum())
            offs.append(a[a[:, 2] > 3, 2] - 3)
        o.read(1 << 24)
        if fin:
            break
    offs = np.concatenate(offs) if offs else np.zeros(0, dtype=np.int64)
    D = len(plain)
    print("%-10s D %10d  C %9d ratio %.3f  blocks %5d  sequences %8d (%.1f per KiB)  repeat-offset codes %.1f %%  mean ml %.1f  mean ll %.1f  "
          "literal bytes %.1f %%" % (name, D, len(z), D / len(z), blocks, nseq, nseq / (D / 1024), 100.0 * rep / max(nseq, 1), ml_sum / max(nseq, 1),
                                     ll_sum / max(nseq, 1), 100.0 * (D - ml_sum) / D))
    if len(offs):
        q = [float(np.mean(offs > t)) * 100 for t in (1 << 10, 16 << 10, 128 << 10, 1 << 20)]
        print("%-10s new offsets: median %d, > 1 KiB %.1f %%, > 16 KiB (a flatten tile) %.1f %%, > 128 KiB %.1f %%, > 1 MiB %.1f %%" % ("", int(np.median(offs)), *q))


n = int(sys.argv[1]) if len(sys.argv) > 1 else 64 << 20
stats("text_like", zgdata.text_like(n, seed=0xE9))
real, info = realtext.load()
stats("realtext", real[:n])
