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

o = g = None
rahead = 0
oracle_default_window = 128 << 20          # DEFAULT_MAX_WINDOW_SIZE, frame_decoder.rs:25
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
    # something behind the frame now and then: a skippable frame, a second (mutated or not) corpus frame — a caller that calls init again
    # where the first frame ended meets them
    if rng.random() < 0.3:
        if rng.random() < 0.5:
            k = rng.randrange(0, 40)
            m += bytes([0x50 + rng.randrange(16), 0x2A, 0x4D, 0x18]) + k.to_bytes(4, "little") + bytes(rng.randrange(256) for _ in range(k))
        m += pack[rng.choice(names)] if not use_dict else b""
    # the decoders live on from input to input (reset() keeps what new() allocated, frame_decoder.rs:200-221), in whatever state the input
    # in front left them; a new pair every 40 inputs
    if it % 40 == 0 or o is None:
        o = oracle.FrameDecoder()
        o.add_dict(raw)
        if g is not None:
            g.close()
        g = zgpu.FrameDecoder(ctx)
        if rng.random() < 0.3:
            w = rng.choice([1 << 10, 1 << 17, 1 << 20, 1 << 23])
            o.set_max_window_size(w)
            g.set_max_window_size(w)
        else:
            g.set_max_window_size(oracle_default_window)
        # the decoder's read-ahead: UptoBytes(n) behaves like UptoBytes(max(n, read_ahead)) of the reference (zgpu_decoder_set_read_ahead)
        rahead = rng.choice([0, 0, 1, 200000, 1 << 20])
        g.set_read_ahead(rahead)
    a, b = o.init(m), g.init(m)
    trace = [("init", a, b)]
    ok = a == b
    if ok and a[0]:
        # a header that is rejected leaves the decoder as the frame in front left it (FrameDecoderState::reset fails before it changes anything)
        sa = (o.can_collect(), o.is_finished(), o.blocks_decoded(), o.bytes_read_from_source(), o.checksum_from_data())
        sb = (g.can_collect(), g.is_finished(), g.blocks_decoded(), g.bytes_read_from_source(), g.get_checksum_from_data())
        trace.append(("state after a failed init", sa, sb))
        ok = sa == sb and o.collect() == g.collect()
    if ok and a[0] == 0:
        pos = a[1]
        for step in range(rng.randrange(4, 60)):
            if o.is_finished() and o.can_collect() == 0 and pos < len(m) and rng.random() < 0.5:
                a, b = o.init(m[pos:]), g.init(m[pos:])                  # the next frame, where this one ended
                trace.append(("init", pos, a, b))
                if a != b:
                    ok = False
                    break
                if a[0] == oracle.ZOR_SKIP_FRAME:                        # (an Err in the reference too: the caller skips the frame's bytes itself)
                    pos = min(len(m), pos + a[1] + a[3])
                    continue
                if a[0]:
                    nerr += 1
                    leaves[a[0]] = leaves.get(a[0], 0) + 1
                    break
                pos += a[1]
                continue
            op = rng.randrange(10)
            if op < 5 and o.is_finished() and rng.random() < 0.9:
                op = 5 + rng.randrange(5)                      # (decode_blocks on a finished frame: an error; seldom)
            if op < 5:
                strat = rng.choice([oracle.STRAT_ALL, oracle.STRAT_UPTO_BLOCKS, oracle.STRAT_UPTO_BLOCKS, oracle.STRAT_UPTO_BYTES, oracle.STRAT_UPTO_BYTES])
                k = rng.choice([0, 1, 2, 3, 7]) if strat == oracle.STRAT_UPTO_BLOCKS else rng.choice([0, 1, 1000, 70000, 200000, 1 << 20])
                # what the caller hands over: everything, or a slice that may end inside a block
                end = len(m) if rng.random() < 0.93 else min(len(m), pos + rng.choice([0, 1, 2, 3, 5, 100, 5000, 140000]))
                a = o.decode_blocks(m[pos:end], strat, max(k, rahead) if strat == oracle.STRAT_UPTO_BYTES else k)
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
                    # the decoder after an Err: its counters, what it still hands out (read() looks at frame_finished, collect() at is_finished())
                    sa = (o.can_collect(), o.is_finished(), o.blocks_decoded(), o.bytes_read_from_source(), o.checksum_from_data())
                    sb = (g.can_collect(), g.is_finished(), g.blocks_decoded(), g.bytes_read_from_source(), g.get_checksum_from_data())
                    ra, rb = o.read(1 << 22), g.read(1 << 22)
                    trace.append(("after the error", sa, sb, len(ra), len(rb)))
                    # (sequence execution failed: what the failing block wrote before it failed is in the buffer on both sides, zg_k_partial)
                    if sa != sb or ra != rb:
                        ok = False
                    break
            elif op == 5 and rng.random() < 0.5:
                # decode_from_to (frame_decoder.rs:439-529): whole blocks of a slice that may end anywhere, a target of any size
                end = len(m) if rng.random() < 0.5 else min(len(m), pos + rng.choice([0, 1, 2, 3, 4, 5, 100, 5000, 140000, 400000]))
                c = rng.choice([0, 3, 1000, 200000, 1 << 22])
                a, b = o.decode_from_to(m[pos:end], c), g.decode_from_to(m[pos:end], c)
                trace.append(("decode_from_to", end - pos, c, a[:2], b[:2], len(a[2]), len(b[2])))
                if a[0] != b[0] or (a[0] == 0 and a != b):
                    ok = False
                    break
                if a[0]:
                    nerr += 1
                    leaves[a[0]] = leaves.get(a[0], 0) + 1
                    sa = (o.can_collect(), o.is_finished(), o.blocks_decoded(), o.bytes_read_from_source(), o.checksum_from_data())
                    sb = (g.can_collect(), g.is_finished(), g.blocks_decoded(), g.bytes_read_from_source(), g.get_checksum_from_data())
                    trace.append(("after the error", sa, sb))
                    if sa != sb:
                        ok = False
                    break
                pos = min(len(m), pos + a[1])
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
    if not ok:
        bad += 1
        if bad <= 5:
            print("DISAGREE", it, name, len(m), trace[-3:])
print("inputs", n, "calls", ncalls, "ended in an error", nerr, "disagreements", bad, "leaves", dict(sorted(leaves.items())))
sys.exit(1 if bad else 0)
