"""(GPU) differential soak on LONG frames (hundreds of blocks to thousands: several units per frame, the split sweep from ~570 MB on, sizes
declared in the header so that the submit is sized in advance): one to two bit flips or a truncation somewhere in a text frame of 24 MiB /
200 MB / 640 MB, through one submit of zgpu_batch_* (verdict, failing block, the bytes of the good blocks), zgpu_decode_all and one read() of
the streaming decoder, against the oracle.   usage: soak_big.py [inputs per size] [seed] [sizes in MiB, comma separated]"""
import hashlib, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for d in ("zstd-rs_amd", "tools", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import oracle, zgdata, zgpu

per = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
sizes = [int(x) << 20 for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["24", "200", "640"])]
ctx = zgpu.Context(0)
bad = nerr = 0
leaves = {}
sha = lambda b: hashlib.sha256(b).hexdigest()
for n in sizes:
    data = zgdata.text_like(n, seed=0xB16 + (n >> 20))
    base = zgdata.zstd_compress(data)
    del data
    ends, p = [], oracle.FrameDecoder().init(base)[1]
    while p + 3 <= len(base):                                   # where the blocks end: a sequence bitstream is read from its last byte backwards
        h = int.from_bytes(base[p:p + 3], "little")
        p += 3 + (1 if ((h >> 1) & 3) == 1 else h >> 3)
        ends.append(p)
        if h & 1:
            break
    for it in range(per):
        m = bytearray(base)
        kind = rng.random()
        if kind < 0.15:
            m = m[:rng.randrange(len(m) // 2, len(m))]
        else:
            for _ in range(rng.choice([1, 1, 2])):
                i = rng.randrange(32, len(m)) if kind < 0.5 else rng.choice(ends) - 1 - rng.randrange(0, 20)   # anywhere / the first sequences' extra bits
                m[i] ^= 1 << rng.randrange(8)
        m = bytes(m)
        t0 = time.time()
        o = oracle.FrameDecoder()
        st, hl, _, _ = o.init(m)
        assert st == 0
        ost, _, _ = o.decode_blocks(m[hl:])
        good = o.blocks_decoded()
        held = o.held(1 << 31)
        t_or = time.time() - t0
        if ost:
            nerr += 1
            leaves[ost] = leaves.get(ost, 0) + 1
        res = []
        try:
            b = zgpu.Batch(ctx, m)
        except zgpu.ZgpuError as e:
            b = None
            res.append(("prepare", e.status))
        if b is not None:
            walk = b.parse_status
            b.run()
            b.sync()
            fi = b.frame_info(0)
            got = b.read(fi.out_base, fi.out_size)
            gst = fi.status or walk
            if gst != ost or (ost and fi.bad_block != good):
                res.append(("batch verdict", ost, gst, good, fi.bad_block, b.sweep_mode()))
            elif (len(got) > len(held) or sha(held[:len(got)]) != sha(got)) if 50 <= ost <= 53 else (len(got) != len(held) or sha(got) != sha(held)):
                res.append(("batch bytes", ost, len(got), len(held), b.sweep_mode()))
            mode = b.sweep_mode()
            b.close()
        try:
            out, gst = ctx.decode_all(m, n + 1024), 0
        except zgpu.ZgpuError as e:
            out, gst = None, e.status
        if gst != ost or (ost == 0 and sha(out) != sha(held)):
            res.append(("decode_all", ost, gst))
        del out
        s = zgpu.CStreamingDecoder(ctx, data=m, checksum=False)
        try:
            out, sst = s.read(n + 1024), 0
        except zgpu.ZgpuError as e:
            out, sst = None, e.status
        if sst != ost or (ost == 0 and sha(out) != sha(held)):
            res.append(("stream", ost, sst))
        s.close()
        del out, held
        print("size %d MiB input %d: oracle %d at block %d (%.1f s) sweep mode %s %s" % (n >> 20, it, ost, good, t_or, mode if b is not None else "-", "DISAGREE " + str(res) if res else "ok"), flush=True)
        if res:
            bad += 1
print("inputs", per * len(sizes), "rejected", nerr, "disagreements", bad, "leaves", dict(sorted(leaves.items())))
sys.exit(1 if bad else 0)
