"""(GPU) differential soak of the FrameDecoder surface: for mutated corpus / dictionary frames, a random sequence of init / decode_blocks
(All, UptoBlocks(n), UptoBytes(n)) / can_collect / collect / read(n) / is_finished / blocks_decoded / bytes_read_from_source calls goes
to zgpu's FrameDecoder and to the oracle's, slices of the source of random length (whole blocks or not) included: every call returns the
same thing on both sides, up to and including the first error.   usage: soak_api.py [inputs] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for d in ("zstd-rs_amd", "tools", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import oracle, zgpu
from golden_io import read_manifest, read_pack

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = zgpu.Context(0)
pack = read_pack("decodecorpus.pack")
dpack = read_pack("dict_tests.pack")
raw = dpack["dictionary"]
ctx.add_dict(raw)
names = sorted(k for k in pack if k.endswith(".zst"))
dnames = sorted(k for k in dpack if k.endswith(".zst"))
bad = nerr = ncalls = 0
leaves = {}
for it in range(n):
    use_dict = it % 5 == 4
    name = rng.choice(dnames if use_dict else names)
    m = bytearray((dpack if use_dict else pack)[name])
    for _ in range(rng.choice([0, 0, 1, 1, 2])):
        i = rng.randrange(4, len(m))
        m[i] ^= 1 << rng.randrange(8)
    if rng.random() < 0.1:
        m = m[:rng.randrange(6, len(m))]
    m = bytes(m)
    o = oracle.FrameDecoder()
    g = zgpu.FrameDecoder(ctx)
    if use_dict:
        o.add_dict(raw)
    a, b = o.init(m), g.init(m)
    trace = [("init", a, b)]
    ok = a == b
    if ok and a[0] == 0:
        pos = a[1]
        for step in range(rng.randrange(4, 60)):
            op = rng.randrange(10)
            if op < 5 and o.is_finished() and rng.random() < 0.9:
                op = 5 + rng.randrange(5)                      # (decode_blocks on a finished frame: an error; seldom)
            if op < 5:
                strat = rng.choice([oracle.STRAT_ALL, oracle.STRAT_UPTO_BLOCKS, oracle.STRAT_UPTO_BLOCKS, oracle.STRAT_UPTO_BYTES, oracle.STRAT_UPTO_BYTES])
                k = rng.choice([0, 1, 2, 3, 7]) if strat == oracle.STRAT_UPTO_BLOCKS else rng.choice([0, 1, 1000, 70000, 200000, 1 << 20])
                # what the caller hands over: everything, or a slice that may end inside a block
                end = len(m) if rng.random() < 0.93 else min(len(m), pos + rng.choice([0, 1, 2, 3, 5, 100, 5000, 140000]))
                a = o.decode_blocks(m[pos:end], strat, k)
                b = g.decode_blocks(m[pos:end], strat, k)
                trace.append(("decode_blocks", strat, k, end - pos, a, b))
                if a[0] != b[0] or (a[0] == 0 and a != b):       # (an Err carries no byte count and no "finished" in the reference)
                    ok = False
                    break
                if a[0] == 0:
                    pos += a[1]
                if a[0]:
                    nerr += 1
                    leaves[a[0]] = leaves.get(a[0], 0) + 1
                    break
            elif op == 5:
                a, b = o.collect(), g.collect()
                trace.append(("collect", len(a), len(b)))
                if a != b:
                    ok = False
                    break
            elif op == 6:
                c = rng.choice([0, 1, 100, 8192, 200000])
                a, b = o.read(c), g.read(c)
                trace.append(("read", c, len(a), len(b)))
                if a != b:
                    ok = False
                    break
            else:
                a = (o.can_collect(), o.is_finished(), o.blocks_decoded(), o.bytes_read_from_source(), o.checksum_from_data())
                b = (g.can_collect(), g.is_finished(), g.blocks_decoded(), g.bytes_read_from_source(), g.get_checksum_from_data())
                trace.append(("state", a, b))
                if a != b:
                    ok = False
                    break
            ncalls += 1
        if ok and o.is_finished() and g.is_finished():
            a, b = o.collect(), g.collect()
            if a != b or o.calculated_checksum() != g.get_calculated_checksum():
                ok = False
                trace.append(("tail", len(a), len(b), o.calculated_checksum(), g.get_calculated_checksum()))
    g.close()
    if not ok:
        bad += 1
        if bad <= 5:
            print("DISAGREE", it, name, len(m), trace[-3:])
print("inputs", n, "calls", ncalls, "ended in an error", nerr, "disagreements", bad, "leaves", dict(sorted(leaves.items())))
sys.exit(1 if bad else 0)
