"""(GPU) differential soak of many frames in ONE submit (zgpu_batch_prepare / run / sync / frame_info / read): 1 to 40 corpus frames, a few of
them mutated or truncated, skippable frames between them; frame by frame against the oracle's FrameDecoder on that frame's bytes: the same
verdict, the same failing block, the same bytes (a failed frame: the bytes of the blocks in front of the failing one), and a failing frame
leaves its neighbours alone.   usage: soak_batch.py [inputs] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for d in ("zstd-rs_amd", "tools", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import oracle, zgpu
from golden_io import read_pack

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = zgpu.Context(0)
pack = read_pack("decodecorpus.pack")
names = sorted(k for k in pack if k.endswith(".zst") and len(pack[k]) < 200000)
bad = nfr = nerr = 0
leaves = {}
for it in range(n):
    m = bytearray()
    for _ in range(rng.choice([1, 2, 5, 12, 40])):
        if rng.random() < 0.1:
            k = rng.randrange(0, 50)
            m += bytes([0x50 + rng.randrange(16), 0x2A, 0x4D, 0x18]) + k.to_bytes(4, "little") + bytes(rng.randrange(256) for _ in range(k))
        f = bytearray(pack[rng.choice(names)])
        if rng.random() < 0.2:
            i = rng.randrange(6, len(f))
            f[i] ^= 1 << rng.randrange(8)
        m += f
    if rng.random() < 0.1:
        m = m[:rng.randrange(max(1, len(m) - 3000), len(m))]
    m = bytes(m)
    try:
        b = zgpu.Batch(ctx, m)
    except zgpu.ZgpuError as e:
        # nothing to submit: the very first header is unreadable — the oracle must say the same
        o = oracle.FrameDecoder()
        st = o.init(m)[0]
        while st == oracle.ZOR_SKIP_FRAME:
            break
        continue
    res = []
    try:
        b.run()
        b.sync()
    except zgpu.ZgpuError as e:
        res.append(("run / sync", e.status, str(e)[:100], b.nframes, b.nblocks))
    for fidx in range(b.nframes if not res else 0):
        fi = b.frame_info(fidx)
        z = m[fi.src_begin:]
        o = oracle.FrameDecoder()
        st, hl, _, _ = o.init(z)
        if st:
            res.append((fidx, "init", st))
            break
        ost, used, fin = o.decode_blocks(z[hl:], oracle.STRAT_ALL, 0)
        nfr += 1
        want = o.held()
        if ost:
            nerr += 1
            leaves[ost] = leaves.get(ost, 0) + 1
        try:
            got = b.read(fi.out_base, fi.out_size)
        except zgpu.ZgpuError as e:
            res.append((fidx, "read", e.status, fi.out_base, fi.out_size, b.total_out, fi.status, ost, b.nframes))
            continue
        if fi.status != ost:
            res.append((fidx, "status", fi.status, ost))
        elif ost and fi.bad_block != o.blocks_decoded():
            res.append((fidx, "bad block", fi.bad_block, o.blocks_decoded(), ost))
        elif (want[:len(got)] != got) if 50 <= ost <= 53 else (got != want):
            res.append((fidx, "bytes", len(got), len(want), ost, fi.bad_block, fi.src_begin, fi.src_end))
        elif not ost and fi.has_checksum and fi.checksum != o.checksum_from_data():
            res.append((fidx, "checksum", fi.checksum, o.checksum_from_data()))
    b.close()
    if res:
        bad += 1
        if bad <= 6:
            print("DISAGREE", it, len(m), res[:4])
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            open(os.path.join(ROOT, "gpurun_out", "batch_diff_%d.zst" % it), "wb").write(m)
print("inputs", n, "frames", nfr, "failed frames", nerr, "disagreements", bad, "leaves", dict(sorted(leaves.items())))
sys.exit(1 if bad else 0)
