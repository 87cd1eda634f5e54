"""zg_k_flatten's scratch against the numpy model (tests/lz_model.py), without the sweep: flatcheck.py [golden frame names ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import zgpu, lz_model
from golden_io import read_manifest, read_pack
os.environ["ZGPU_DEBUG_NO_SWEEP"] = "1"
pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
names = sys.argv[1:] or sorted(man)
ctx = zgpu.Context(0, dev=True)
nbad = 0
for name in names:
    z = pack[name]
    b = ctx.prepare(z)
    b.run(); b.sync()
    units = b.units()
    e, bounds = lz_model.expected_scratch(z, [u[0] for u in units])
    for ui, (fb, nb, base, size, noseq) in enumerate(units):
        want = e[bounds[ui]:bounds[ui + 1]]
        if noseq:
            assert not want.any()
            continue
        if size != len(want):
            print(name, "unit", ui, "size", size, "want", len(want)); nbad += 1; continue
        got = b.scratch_words(base, size)
        bad = np.flatnonzero(got != want)
        if len(bad):
            nbad += 1
            i = int(bad[0])
            print(name, "unit", ui, "blocks", fb, nb, "size", size, "mismatches", len(bad), "first at", i, "got", got[i:i + 8], "want", want[i:i + 8], flush=True)
    b.close()
print("frames", len(names), "bad units", nbad)
