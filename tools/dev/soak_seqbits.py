"""(GPU) differential soak aimed at the sequence EXECUTION errors (offset in front of everything, offset beyond the window, no literals left,
offset 0) and at what stands beside them: one to three bit flips inside the sequence bitstream of one compressed block of a corpus / synthetic
frame (the bitstream is read backwards: its last bytes hold the first sequences' extra bits). Every input goes through zgpu_decode_all,
zgpu_pool_decode_all, FrameDecoder::decode_blocks(All), one read() of the streaming decoder and one submit of zgpu_batch_*: the oracle's verdict
everywhere, its bytes where it decodes, the bytes of the good blocks where it does not.   usage: soak_seqbits.py [inputs] [seed]"""
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
packs, syn = read_pack("decodecorpus.pack"), read_pack("synthetic.pack")
bases = [packs[k] for k in sorted(packs) if k.endswith(".zst")] + [syn[k] for k in sorted(syn) if k.endswith(".zst") and len(syn[k]) < (1 << 20)]


def compressed_blocks(z):
    st, c, _, _ = oracle.FrameDecoder().init(z)
    out, p = [], c
    while st == 0 and p + 3 <= len(z):
        h = int.from_bytes(z[p:p + 3], "little")
        ty, size = (h >> 1) & 3, h >> 3
        clen = 1 if ty == 1 else size
        if ty == 2 and clen > 16:
            out.append((p + 3, clen))
        p += 3 + clen
        if h & 1:
            break
    return out


blocks = [compressed_blocks(b) for b in bases]
usable = [i for i, b in enumerate(blocks) if b]
bad = nerr = 0
leaves = {}
for it in range(n):
    bi = rng.choice(usable)
    m = bytearray(bases[bi])
    off, ln = rng.choice(blocks[bi])
    for _ in range(rng.choice([1, 1, 2, 3])):
        # the end of the block (start of the backward stream) most of the time, anywhere in its second half else
        i = off + ln - 1 - rng.randrange(0, min(ln, 24)) if rng.random() < 0.6 else off + rng.randrange(ln // 2, ln)
        m[i] ^= 1 << rng.randrange(8)
    m = bytes(m)
    o = oracle.FrameDecoder()
    st, hl, _, _ = o.init(m)
    assert st == 0
    ost, _, _ = o.decode_blocks(m[hl:])
    held = o.held()
    good = o.blocks_decoded()
    ast, aout = oracle.FrameDecoder().decode_all(m, 1 << 25)
    res = []
    if ost:
        nerr += 1
        leaves[ost] = leaves.get(ost, 0) + 1
    for what, fn in (("decode_all", lambda: ctx.decode_all(m, 1 << 25)), ("pool", lambda: pool.decode_all(m, 1 << 25))):
        try:
            out, gst = fn(), 0
        except zgpu.ZgpuError as e:
            out, gst = None, e.status
        if gst != ast or (ast == 0 and out != aout):
            res.append((what, ast, gst))
    g = zgpu.FrameDecoder(ctx)
    g.init(m)
    gst = g.decode_blocks(m[hl:])[0]
    if gst != ost or g.blocks_decoded() != good or g.can_collect() != o.can_collect():
        res.append(("decode_blocks", ost, gst, good, g.blocks_decoded(), g.can_collect(), o.can_collect()))
    g.close()
    # the thin boundary, everything in one run: after an error all that the reference's buffer holds is readable, the failing block's
    # partial output included
    if ost:
        from test_gpu_thin_boundary import parse_frame_header
        tblocks, p = [], hl
        while p + 3 <= len(m):
            h = int.from_bytes(m[p:p + 3], "little")
            ty, size = (h >> 1) & 3, h >> 3
            if ty == 3 or size > (128 << 10) or len(m) - p - 3 < (1 if ty == 1 else size):
                break
            tblocks.append((p + 3, 1 if ty == 1 else size, ty, h & 1, size if ty != 2 else 0))
            p += 3 + (1 if ty == 1 else size)
            if h & 1:
                break
        if len(tblocks) > good:
            tf = zgpu.BlockFrame(ctx, o.window_size(), o.content_size(), 0)
            tf.submit(m, tblocks)
            tb, ts = tf.sync()
            x = tf.read(1 << 25, True)
            if (tb, ts) != (good, ost) or x != held:
                res.append(("thin", (tb, ts), (good, ost), len(x), len(held)))
            tf.close()
    s = zgpu.CStreamingDecoder(ctx, data=m)
    try:
        sout, sst = s.read(1 << 25), 0
    except zgpu.ZgpuError as e:
        sout, sst = None, e.status
    if sst != ost or (ost == 0 and sout != held):
        res.append(("stream", ost, sst))
    s.close()
    b = zgpu.Batch(ctx, m)
    b.run()
    b.sync()
    fi = b.frame_info(0)
    got = b.read(fi.out_base, fi.out_size)
    if fi.status != ost or (ost and fi.bad_block != good) or ((held[:len(got)] != got or len(got) > len(held)) if 50 <= ost <= 53 else got != held):
        res.append(("batch", ost, fi.status, good, fi.bad_block, len(got), len(held)))
    b.close()
    if res:
        bad += 1
        if bad <= 6:
            print("DISAGREE", it, bi, len(m), res)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            open(os.path.join(ROOT, "gpurun_out", "seqbits_diff_%d.zst" % it), "wb").write(m)
print("inputs", n, "rejected", nerr, "disagreements", bad, "leaves", dict(sorted(leaves.items())))
sys.exit(1 if bad else 0)
