"""What the sweep has to do, from zg_k_flatten's scratch of a generated text frame: share of literal bytes, of match bytes whose
final source is a literal byte inside their unit, and of match bytes that copy from in front of their unit; lengths of runs of
equal offset. usage: og_classes.py [size] [unit blocks]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import zgdata, zgpu
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256 << 20
if len(sys.argv) > 2:
    os.environ["ZGPU_UNIT_BLOCKS"] = sys.argv[2]
os.environ["ZGPU_DEBUG_NO_SWEEP"] = "1"
z = zgdata.zstd_compress(zgdata.text_like(size))
ctx = zgpu.Context(0, dev=True)
b = ctx.prepare(z)
b.run(); b.sync()
tot = lit = inside = before = runs = full64 = n64 = 0
for fb, nb, base, usz, noseq in b.units()[:48]:
    if noseq:
        continue
    e = b.scratch_words(base, usz).astype(np.int64)
    x = np.arange(usz, dtype=np.int64)
    tot += usz; lit += int((e == 0).sum()); inside += int(((e > 0) & (e <= x)).sum()); before += int((e > x).sum())
    runs += int((np.diff(e) != 0).sum()) + 1
    k = usz // 64 * 64
    blk = (e[:k] > x[:k]).reshape(-1, 64)
    n64 += blk.shape[0]; full64 += int((~blk.any(axis=1)).sum())
print("bytes %d: literal %.1f%%, source is a literal of the unit %.1f%%, source in front of the unit %.1f%%; runs of equal offset: %.1f bytes on average; "
      "64-byte lines with nothing to fetch from in front of the unit: %.1f%%" % (tot, 100 * lit / tot, 100 * inside / tot, 100 * before / tot, tot / runs, 100 * full64 / n64))
