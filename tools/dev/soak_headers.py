"""(GPU) differential soak aimed at the HEADERS: structured edits of a corpus frame — frame header descriptor bits (checksum flag, single
segment, content-size and dictionary-id field sizes), the window descriptor, the declared content size, block headers (type, last-block bit,
size up or down), a frame cut right behind a header — through zgpu_decode_all, zgpu_pool_decode_all, FrameDecoder::decode_blocks(All) with the
accessors behind it, and the streaming decoder under a random read pattern in two modes.   usage: soak_headers.py [inputs] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for d in ("zstd-rs_amd", "tools", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import oracle, zgpu
from golden_io import read_pack
from test_gpu_stream import oracle_reads, zgpu_reads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = zgpu.Context(0)
pool = zgpu.Pool()
pack = read_pack("decodecorpus.pack")
names = sorted(k for k in pack if k.endswith(".zst"))


def headers(z):
    st, c, _, _ = oracle.FrameDecoder().init(z)
    out, p = [], c
    while st == 0 and p + 3 <= len(z):
        h = int.from_bytes(z[p:p + 3], "little")
        out.append(p)
        p += 3 + (1 if ((h >> 1) & 3) == 1 else h >> 3)
        if h & 1:
            break
    return c, out


bad = nerr = 0
leaves = {}
for it in range(n):
    m = bytearray(pack[rng.choice(names)])
    hl, hs = headers(bytes(m))
    for _ in range(rng.choice([1, 1, 2])):
        k = rng.randrange(9)
        if k == 0:
            m[4] ^= 1 << rng.choice([2, 5, 6, 7, 0, 1, 3, 4])            # descriptor: checksum / single segment / FCS size / dict id size / reserved / unused
        elif k == 1 and hl > 5:
            m[5] = rng.randrange(256)                                    # window descriptor (or the first byte of what follows the descriptor)
        elif k == 2 and hl > 6:
            i = rng.randrange(5, hl)
            m[i] = rng.randrange(256)                                    # dictionary id / content size bytes
        elif k == 3 and hs:
            m[rng.choice(hs)] ^= 1                                       # last-block bit
        elif k == 4 and hs:
            m[rng.choice(hs)] ^= rng.choice([2, 4, 6])                   # block type
        elif k == 5 and hs:
            p = rng.choice(hs)
            h = int.from_bytes(m[p:p + 3], "little")
            size = max(0, min((1 << 21) - 1, (h >> 3) + rng.choice([-1, 1, -3, 7, 1 << 17, -(1 << 10)])))
            m[p:p + 3] = ((h & 7) | (size << 3)).to_bytes(3, "little")   # block size
        elif k == 6 and hs:
            p = rng.choice(hs)
            m = m[:p + rng.choice([0, 1, 2, 3, 4])]                      # cut at / inside / right behind a block header
            break
        elif k == 7:
            m = m[:rng.randrange(0, min(len(m), hl + 2))]                # cut inside the frame header
            break
        else:
            m += bytes(rng.randrange(256) for _ in range(rng.randrange(1, 9)))   # junk behind the frame
        if len(m) < 5:
            break
    m = bytes(m)
    res = []
    ast, aout = oracle.FrameDecoder().decode_all(m, 1 << 25)
    if ast:
        nerr += 1
        leaves[ast] = leaves.get(ast, 0) + 1
    for what, fn in (("decode_all", lambda: ctx.decode_all(m, 1 << 25)), ("pool", lambda: pool.decode_all(m, 1 << 25))):
        try:
            out, gst = fn(), 0
        except zgpu.ZgpuError as e:
            out, gst = None, e.status
        if gst != ast or (ast == 0 and out != aout):
            res.append((what, ast, gst))
    o, g = oracle.FrameDecoder(), zgpu.FrameDecoder(ctx)
    a, b = o.init(m), g.init(m)
    if a != b:
        res.append(("init", a, b))
    elif a[0] == 0:
        a, b = o.decode_blocks(m[a[1]:]), g.decode_blocks(m[a[1]:])
        sa = (a[0], o.can_collect(), o.is_finished(), o.blocks_decoded(), o.bytes_read_from_source(), o.checksum_from_data(), o.content_size(), o.window_size() if hasattr(o, "window_size") else 0)
        sb = (b[0], g.can_collect(), g.is_finished(), g.blocks_decoded(), g.bytes_read_from_source(), g.get_checksum_from_data(), g.content_size(), sa[7])
        if (a != b if a[0] == 0 else False) or sa != sb:
            res.append(("decode_blocks", a, b, sa, sb))
    g.close()
    total = len(aout) if ast == 0 else 300000
    reads = []
    done = 0
    while done < total + 1000 and len(reads) < 400:
        c = rng.choice([1, 4096, 8192, 65536, 131072, 131073, 1 << 20])
        reads.append(c)
        done += c
    want, _ = oracle_reads(m, reads)
    for kw in (dict(), dict(read_ahead=1), dict(pipe_after=1, read_ahead=4 << 20)):
        got, s = zgpu_reads(ctx, m, reads, rng.random() < 0.5, **kw)
        if len(got) != len(want) or any(x[0] != y[0] or x[1] != y[1] for x, y in zip(got, want)):
            res.append(("stream", kw, len(got), len(want), got[-1][:1] if got else None, want[-1][:1] if want else None,
                        got[-1][1] if got and got[-1][0] in ("err", "init") else None, want[-1][1] if want and want[-1][0] in ("err", "init") else None))
        if s is not None:
            s.close()
    if res:
        bad += 1
        if bad <= 8:
            print("DISAGREE", it, len(m), res[:3])
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            open(os.path.join(ROOT, "gpurun_out", "hdr_diff_%d.zst" % it), "wb").write(m)
print("inputs", n, "rejected", nerr, "disagreements", bad, "leaves", dict(sorted(leaves.items())))
sys.exit(1 if bad else 0)
