"""(GPU) differential soak of the io::Read surface: mutated corpus / dictionary frames through zgpu_streaming_* in every mode, with random
read sizes, against the oracle's FrameDecoder driven by the reference's read loop (tests/test_gpu_stream.py oracle_reads): the same bytes,
the same return value of every read(), an error in the same call with the same leaf.   usage: soak_stream.py [inputs] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for d in ("zstd-rs_amd", "tools", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import zgpu
from golden_io import read_manifest, read_pack
from test_gpu_stream import MODES, oracle_reads, patterns, zgpu_reads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = zgpu.Context(0)
pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
dpack, dman = read_pack("dict_tests.pack"), read_manifest("dict_tests.json")
raw = dpack["dictionary"]
ctx.add_dict(raw)
names = sorted(man)
dnames = sorted(k for k in dman if k.endswith(".zst"))
bad = nerr = 0
leaves = {}
for it in range(n):
    use_dict = it % 5 == 4
    name = rng.choice(dnames if use_dict else names)
    m = bytearray((dpack if use_dict else pack)[name])
    for _ in range(rng.choice([0, 1, 1, 2, 3])):
        i = rng.randrange(4, len(m))
        m[i] ^= 1 << rng.randrange(8)
    if rng.random() < 0.15:
        m = m[:rng.randrange(5, len(m))]                     # a truncated source
    z = bytes(m)
    size = (dman if use_dict else man)[name]["size"]
    pat = list(patterns(rng, size))[it % 3]
    want, o = oracle_reads(z, pat, raw if use_dict else None)
    if want[-1][0] in ("err", "init"):
        nerr += 1
        leaves[want[-1][1]] = leaves.get(want[-1][1], 0) + 1
    for kw in (MODES[it % 4], MODES[(it + 1) % 4]):
        got, s = zgpu_reads(ctx, z, pat, it % 2 == 1, **kw)
        ok = len(got) == len(want) and all(g[0] == w[0] and g[1] == w[1] for g, w in zip(got, want))
        if not ok:
            bad += 1
            k = next((i for i, (g, w) in enumerate(zip(got, want)) if g[0] != w[0] or g[1] != w[1]), min(len(got), len(want)))
            print("DISAGREE", it, name, kw, "dict" if use_dict else "", "read", k, "cap", pat[k] if k < len(pat) else None,
                  "got", got[k][:1] if k < len(got) else None, (got[k][1] if k < len(got) and isinstance(got[k][1], int) else ""),
                  "want", want[k][:1] if k < len(want) else None, (want[k][1] if k < len(want) and isinstance(want[k][1], int) else ""), flush=True)
        if s and ok and kw.get("read_ahead") == 1 and o is not None:
            # block by block, on the reference's schedule: the counters are the reference's too, also after an Err
            a = (o.blocks_decoded(), o.bytes_read_from_source(), o.is_finished())
            b = (s.blocks_decoded(), s.bytes_read_from_source(), s.is_finished())
            if a != b:
                bad += 1
                print("DISAGREE", it, name, "counters", a, b, want[-1][:2] if want and want[-1][0] == "err" else "", flush=True)
        if s:
            s.close()
print("inputs", n, "with errors", nerr, "disagreements", bad, "leaves", dict(sorted(leaves.items())))
