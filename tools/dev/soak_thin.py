"""(GPU) differential soak of the thin boundary (zgpu_frame_begin / zgpu_blocks_submit / zgpu_sync / zgpu_read): the host side — this script —
reads the frame header and the block headers of a mutated corpus frame the way ruzstd's FrameDecoder does (it stops at a header it cannot
read or a body that is not all there), hands the blocks over in runs of random length and drains a random amount between them; the oracle's
FrameDecoder gets the same blocks with UptoBlocks(run length). After every run: the same verdict, the same failing block, the same number of
blocks decoded, the same bytes out of read().   usage: soak_thin.py [inputs] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for d in ("zstd-rs_amd", "tools", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import oracle, zgpu
from golden_io import read_pack

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = zgpu.Context(0)
pack = read_pack("decodecorpus.pack")
names = sorted(k for k in pack if k.endswith(".zst"))
bad = nerr = nruns = 0
leaves = {}


def walk(z, p):
    """the host's read_block_header loop (block_decoder.rs:201-247): whole blocks only; stops like the reference's reader would fail"""
    out = []
    while len(z) - p >= 3:
        h = z[p] | (z[p + 1] << 8) | (z[p + 2] << 16)
        last, ty, size = h & 1, (h >> 1) & 3, h >> 3
        if ty == 3 or size > (128 << 10):
            break
        clen = 1 if ty == 1 else size
        if len(z) - p - 3 < clen:
            break
        out.append((p + 3, clen, ty, last, size if ty != 2 else 0))
        p += 3 + clen
        if last:
            break
    return out


for it in range(n):
    m = bytearray(pack[rng.choice(names)])
    for _ in range(rng.choice([0, 1, 1, 2])):
        i = rng.randrange(6, len(m))
        m[i] ^= 1 << rng.randrange(8)
    m = bytes(m)
    o = oracle.FrameDecoder()
    st, hl, _, _ = o.init(m)
    if st:
        continue
    blocks = walk(m, hl)
    if not blocks:
        continue
    f = zgpu.BlockFrame(ctx, o.window_size(), o.content_size(), 0)
    i, pos, ok, trace = 0, hl, True, []
    while i < len(blocks) and ok:
        k = rng.choice([1, 1, 2, 3, 8, 1000])
        run = blocks[i:i + k]
        f.submit(m, run)
        gb, gs = f.sync()
        # the oracle reads the same blocks: hand it exactly their bytes (a checksum behind the last block is the host's business here)
        end = run[-1][0] + run[-1][1]
        a = o.decode_blocks(m[pos:end] + (b"\0\0\0\0" if run[-1][3] else b""), oracle.STRAT_UPTO_BLOCKS, len(run))
        nruns += 1
        trace.append((i, len(run), (gb, gs), a))
        if a[0] == 0:
            if gb is not None or gs != 0:
                ok = False
                break
            pos += a[1] - (4 if run[-1][3] and o.checksum_from_data() is not None else 0)
            i += len(run)
            fin = bool(run[-1][3])
            if f.blocks_decoded() != o.blocks_decoded():
                ok = False
                break
            want = f.available(fin)
            cap = rng.choice([want, want, want // 2, 0, 1 << 22])
            x, y = f.read(cap, fin), o.read(cap)
            if x != y:
                trace.append(("read", cap, len(x), len(y)))
                ok = False
        else:
            nerr += 1
            leaves[a[0]] = leaves.get(a[0], 0) + 1
            if gs != a[0] or gb != o.blocks_decoded() or f.blocks_decoded() != o.blocks_decoded():
                trace.append(("verdict", gb, gs, a[0], o.blocks_decoded(), f.blocks_decoded()))
                ok = False
                break
            # everything the reference's buffer holds now — the good blocks of this run and, after a sequence execution error, what the
            # failing block wrote before it failed (zg_k_partial) — is there: read it all (finished = 1 hands out the window too)
            x, y = f.read(1 << 24, True), o.held()
            if x != y:
                trace.append(("held after the error", len(x), len(y)))
                ok = False
            break
    f.close()
    if not ok:
        bad += 1
        if bad <= 6:
            print("DISAGREE", it, len(m), trace[-3:])
print("inputs", n, "runs", nruns, "ended in an error", nerr, "disagreements", bad, "leaves", dict(sorted(leaves.items())))
sys.exit(1 if bad else 0)
