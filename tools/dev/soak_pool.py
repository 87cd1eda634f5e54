"""(GPU) differential soak of the pool's staged form (zgpu_pool_stage / run / frame / read: what bench.py --gpus N drives): 1 to 60 inputs per
stage call — corpus frames, some mutated or truncated, some made of two frames and a skippable one — item by item against the oracle's
decode_all on that item: the same verdict, the same bytes; run twice (a pass over resident jobs is repeatable).  usage: soak_pool.py [stages] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for d in ("zstd-rs_amd", "tools", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import oracle, zgpu
from golden_io import read_pack

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
pool = zgpu.Pool()
pack = read_pack("decodecorpus.pack")
names = sorted(k for k in pack if k.endswith(".zst") and len(pack[k]) < 200000)
bad = nitems = nerr = 0
leaves = {}
for it in range(n):
    items, walk_bad = [], []
    for _ in range(rng.choice([1, 2, 7, 25, 60])):
        f = bytearray(pack[rng.choice(names)])
        r = rng.random()
        if r < 0.2:
            i = rng.randrange(6, len(f))
            f[i] ^= 1 << rng.randrange(8)
        elif r < 0.25:
            f = f[:rng.randrange(5, len(f))]
        elif r < 0.35:
            k = rng.randrange(0, 40)
            f += bytes([0x50 + rng.randrange(16), 0x2A, 0x4D, 0x18]) + k.to_bytes(4, "little") + bytes(rng.randrange(256) for _ in range(k)) + pack[rng.choice(names)]
        f = bytes(f)
        # (an entry the header walk cannot read to its end fails the whole stage call, include/zgpu.h: such entries are checked one by one)
        if oracle.FrameDecoder().decode_all(f, 1 << 25)[0] in (2, 3, 4, 5, 6, 7, 9, 10, 11, 13, 20, 21):
            try:
                pool.stage([f])
                walk_bad.append(("staged although the oracle says", oracle.FrameDecoder().decode_all(f, 1 << 25)[0]))
            except zgpu.ZgpuError as e:
                pass
            continue
        items.append(f)
    res = list(walk_bad)
    try:
        pool.stage(items)
        for rep in range(2):
            pool.run()
            for i, z in enumerate(items):
                ost, oout = oracle.FrameDecoder().decode_all(z, 1 << 25)
                gpu, size, st = pool.frame(i)
                if rep == 0:
                    nitems += 1
                    if ost:
                        nerr += 1
                        leaves[ost] = leaves.get(ost, 0) + 1
                if st != ost:
                    res.append((rep, i, "status", st, ost))
                elif ost == 0 and (size != len(oout) or pool.read(i, size) != oout):
                    res.append((rep, i, "bytes", size, len(oout)))
    except zgpu.ZgpuError as e:
        # a stage call fails as a whole when the HOST cannot read an entry (headers down to the blocks' section headers): then one of the
        # entries must fail the same way alone
        alone = []
        for z in items:
            try:
                pool.stage([z])
            except zgpu.ZgpuError as e2:
                alone.append(e2.status)
        if not alone:                                   # (a truncated entry runs into its neighbour: the error's leaf may differ)
            res.append(("call failed", e.status, str(e)[:80], len(items), alone[:5]))
    if res:
        bad += 1
        if bad <= 6:
            print("DISAGREE", it, len(items), res[:4])
print("stage calls", n, "items", nitems, "rejected", nerr, "disagreements", bad, "leaves", dict(sorted(leaves.items())))
sys.exit(1 if bad else 0)
