"""(GPU) differential soak of decode_all on CONCATENATIONS: two to four corpus frames (each mutated or truncated now and then), skippable frames
and junk between and behind them, through zgpu_decode_all, zgpu_decode_all_alloc and zgpu_pool_decode_all against the oracle's decode_all:
the same verdict, the same bytes when it is Ok, also with a target that is too small by a little.   usage: soak_concat.py [inputs] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for d in ("zstd-rs_amd", "tools", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import oracle, zgpu
from golden_io import read_pack

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = zgpu.Context(0)
pool = zgpu.Pool()
pack = read_pack("decodecorpus.pack")
names = sorted(k for k in pack if k.endswith(".zst") and len(pack[k]) < 300000)
dpack = read_pack("dict_tests.pack")
dnames = sorted(k for k in dpack if k.endswith(".zst"))
ctx.add_dict(dpack["dictionary"])             # (frames that name it go one by one through the FrameDecoder mirror inside zgpu_decode_all; the pool has no dictionaries)
bad = nerr = 0
leaves = {}
for it in range(n):
    m = bytearray()
    with_dict = it % 4 == 3
    for _ in range(rng.randrange(1, 5)):
        r = rng.random()
        if r < 0.2:
            k = rng.randrange(0, 60)
            m += bytes([0x50 + rng.randrange(16), 0x2A, 0x4D, 0x18]) + k.to_bytes(4, "little") + bytes(rng.randrange(256) for _ in range(k))
        f = bytearray(dpack[rng.choice(dnames)] if with_dict and rng.random() < 0.6 else pack[rng.choice(names)])
        if rng.random() < 0.25:
            i = rng.randrange(4, len(f))
            f[i] ^= 1 << rng.randrange(8)
        if rng.random() < 0.05:
            f = f[:rng.randrange(5, len(f))]
        m += f
    if rng.random() < 0.1:
        m += bytes(rng.randrange(256) for _ in range(rng.randrange(1, 12)))
    m = bytes(m)
    def ora(cap):
        o = oracle.FrameDecoder()
        o.add_dict(dpack["dictionary"])
        return o.decode_all(m, cap)
    ost, oout = ora(1 << 25)
    cap = 1 << 25
    if ost == 0 and rng.random() < 0.15:
        cap = max(0, len(oout) - rng.choice([1, 2, 100, 70000]))
        ost, oout = ora(cap)
    res = []
    for what, fn in (("decode_all", lambda: ctx.decode_all(m, cap)),) + ((("pool", lambda: pool.decode_all(m, cap)),) if not with_dict else ()) + \
                    ((("alloc", lambda: ctx.decode_all_to_vec(m)),) if cap == 1 << 25 else ()):
        try:
            out, gst = fn(), 0
        except zgpu.ZgpuError as e:
            out, gst = None, e.status
        if gst != ost or (ost == 0 and out != oout):
            res.append((what, ost, gst, None if out is None else len(out), len(oout)))
    if ost:
        nerr += 1
        leaves[ost] = leaves.get(ost, 0) + 1
    if res:
        bad += 1
        if bad <= 6:
            print("DISAGREE", it, len(m), res)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            open(os.path.join(ROOT, "gpurun_out", "concat_diff_%d.zst" % it), "wb").write(m)
print("inputs", n, "rejected", nerr, "disagreements", bad, "leaves", dict(sorted(leaves.items())))
sys.exit(1 if bad else 0)
