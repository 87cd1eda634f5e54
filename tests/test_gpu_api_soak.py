"""Random call sequences on the FrameDecoder surface (init / decode_blocks with every strategy / collect / read / the counters), on mutated
corpus and dictionary frames, zgpu against the oracle call by call: tools/dev/soak_api.py with a fixed seed (the soak runs of a round use
more inputs: profiles/r06/notes.md)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_frame_decoder_call_sequences_match_the_oracle():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dev", "soak_api.py"), "600", "11"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "disagreements 0" in r.stdout


def test_decode_blocks_goes_on_behind_the_frames_end_like_the_reference():
    """decode_blocks does not look at frame_finished when it starts (frame_decoder.rs:321-359): a caller that goes on after the last block has
    its bytes read as further blocks, and every strategy ends with a last block of THAT call, a block count or a growth — not with the flag
    the earlier call left (found by tools/dev/soak_api.py: UptoBytes stopped after one block)"""
    sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    import zgpu
    from test_exact_cpu import frame, raw_block, rle_block
    z = frame(raw_block(300, 1), rle_block(50, last=True)) + raw_block(20, 2) + rle_block(7) + raw_block(5, 3, last=True) + rle_block(9, last=True)
    ctx = zgpu.Context(0)
    for strat, n in ((oracle.STRAT_UPTO_BYTES, 1000), (oracle.STRAT_UPTO_BYTES, 21), (oracle.STRAT_UPTO_BLOCKS, 2), (oracle.STRAT_ALL, 0)):
        o, g = oracle.FrameDecoder(), zgpu.FrameDecoder(ctx)
        a, b = o.init(z), g.init(z)
        assert a == b and a[0] == 0
        pos = a[1]
        a, b = o.decode_blocks(z[pos:], oracle.STRAT_ALL, 0), g.decode_blocks(z[pos:], oracle.STRAT_ALL, 0)
        assert a == b and a[0] == 0 and a[2]
        pos += a[1]
        while pos < len(z):
            a, b = o.decode_blocks(z[pos:], strat, n), g.decode_blocks(z[pos:], strat, n)
            assert a == b and a[0] == 0 and a[1] > 0, (strat, n, pos, a, b)
            pos += a[1]
            assert (o.can_collect(), o.blocks_decoded(), o.bytes_read_from_source()) == (g.can_collect(), g.blocks_decoded(), g.bytes_read_from_source())
        assert o.collect() == g.collect()
        g.close()
    ctx.close()


def test_decoder_state_after_an_error_is_the_references():
    """what the accessors say after decode_blocks returned an Err (found by the soak once it looked there): the failing block's header is
    counted when its body failed (frame_decoder.rs:325-341); a last block whose checksum is missing has finished the frame for read(),
    which hands out everything, but not for is_finished() (:347-357, :284-294, :615-627)"""
    sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    import zgpu
    from test_exact_cpu import WINDOW_LOG, frame, lit_block, raw_block, rle_block, seq_block
    with_checksum = bytes([0x28, 0xB5, 0x2F, 0xFD, 0x04, (WINDOW_LOG - 10) << 3])
    cases = [
        frame(raw_block(300, 1), raw_block(500, 2, last=True))[:-100],                       # the second block's body is cut
        frame(raw_block(300, 1))[:50],                                                      # the first block's body is cut: no block in the run
        with_checksum + raw_block(300, 1) + rle_block(4000, last=True) + b"\x01\x02",       # two of the checksum's four bytes
        frame(raw_block(300, 1), bytes([0x06 | 1, 0, 0])),                                  # reserved block type
        frame(lit_block(200), seq_block(100000, last=True)),                                # an offset beyond everything: sequence execution fails
    ]
    ctx = zgpu.Context(0)
    for i, z in enumerate(cases):
        for strat, n in ((oracle.STRAT_ALL, 0), (oracle.STRAT_UPTO_BLOCKS, 1), (oracle.STRAT_UPTO_BYTES, 100000)):
            o, g = oracle.FrameDecoder(), zgpu.FrameDecoder(ctx)
            a, b = o.init(z), g.init(z)
            assert a == b and a[0] == 0
            pos = a[1]
            err = 0
            for _ in range(4):
                a, b = o.decode_blocks(z[pos:], strat, n), g.decode_blocks(z[pos:], strat, n)
                assert a[0] == b[0], (i, strat, a, b)
                if a[0]:
                    err = a[0]
                    break
                assert a == b
                pos += a[1]
            assert err, (i, strat)
            sa = (o.is_finished(), o.blocks_decoded(), o.bytes_read_from_source(), o.checksum_from_data())
            sb = (g.is_finished(), g.blocks_decoded(), g.bytes_read_from_source(), g.get_checksum_from_data())
            assert sa == sb, (i, strat, err, sa, sb)
            assert o.can_collect() == g.can_collect(), (i, strat, err)   # (after a sequence execution error too: zg_k_partial)
            assert o.read(1 << 20) == g.read(1 << 20), (i, strat, err)
            g.close()
    ctx.close()


def test_an_error_in_front_of_the_point_where_the_walk_stops_comes_first_in_every_decode_all():
    """a frame whose block headers cannot be walked to the end (reserved block type) but which holds a block that fails in FRONT of that
    point: the reference decodes block by block and meets the block's error first (frame_decoder.rs:319-375). zgpu_decode_all did that
    since round 4; zgpu_pool_decode_all answered with the walk's error until tools/dev/soak_concat.py compared the two"""
    sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    import zgpu
    from test_exact_cpu import frame, lit_block, raw_block, rle_block, seq_block
    whole = frame(raw_block(1000, 3), rle_block(77, last=True))
    reserved = bytes([0x06, 0, 0])
    bad_body = (((3 << 3) | (2 << 1)).to_bytes(3, "little")) + b"\xff\xff\xff"        # a compressed block of three bytes of nonsense
    ctx = zgpu.Context(0)
    pool = zgpu.Pool()
    for z in (whole + frame(lit_block(200), bad_body, reserved),
              whole + frame(lit_block(200), seq_block(100000), reserved),               # sequence execution fails in front of the stop
              whole + frame(lit_block(200), reserved),                                  # nothing fails in front: the walk's error
              frame(lit_block(100), bad_body)):                                         # ... and a walk that runs out of bytes
        ost, oout = oracle.FrameDecoder().decode_all(z, 1 << 20)
        assert ost != 0
        for fn in (ctx.decode_all, pool.decode_all):
            with pytest.raises(zgpu.ZgpuError) as e:
                fn(z, 1 << 20)
            assert e.value.status == ost, (fn, e.value.status, ost)
    pool.close()
    ctx.close()


def test_concatenations_through_every_decode_all():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dev", "soak_concat.py"), "300", "12"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "disagreements 0" in r.stdout


def test_decode_from_to_against_the_oracle():
    """decode_from_to (frame_decoder.rs:439-529) on slices that end anywhere and targets of any size, call by call against the oracle's
    restatement (pinned on tests/mod.rs:129-230, :382-404 in test_oracle_golden.py); a header that cannot be read behind whole blocks: the
    blocks are decoded, then that error is the answer (zgpu answered Ok until tools/dev/soak_api.py called this function)"""
    import random
    sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import oracle
    import zgpu
    from golden_io import read_pack
    from test_exact_cpu import frame, raw_block, rle_block
    ctx = zgpu.Context(0)
    z = frame(raw_block(300, 1), rle_block(200), bytes([0x06, 0, 0]))               # two whole blocks, then a reserved block type
    for cap in (0, 100, 1 << 20):
        o, g = oracle.FrameDecoder(), zgpu.FrameDecoder(ctx)
        a, b = o.decode_from_to(z, cap), g.decode_from_to(z, cap)
        assert a[0] == b[0] == 20
        assert (o.blocks_decoded(), o.bytes_read_from_source(), o.can_collect()) == (g.blocks_decoded(), g.bytes_read_from_source(), g.can_collect())
        assert o.read(1 << 20) == g.read(1 << 20)
        g.close()
    pack = read_pack("decodecorpus.pack")
    rng = random.Random(0xF70)
    for name in sorted(k for k in pack if k.endswith(".zst"))[::4]:
        z = pack[name]
        o, g = oracle.FrameDecoder(), zgpu.FrameDecoder(ctx)
        pos = 0
        for _ in range(200):
            end = len(z) if rng.random() < 0.3 else min(len(z), pos + rng.choice([0, 2, 3, 7, 1000, 70000, 140000, 300000]))
            cap = rng.choice([0, 3, 5000, 1 << 22])
            if pos == 0:
                end = max(end, min(len(z), 32))             # (the first call reads the frame header: it must be there)
            a, b = o.decode_from_to(z[pos:end], cap), g.decode_from_to(z[pos:end], cap)
            assert a == b, (name, pos, end, cap, a[:2], b[:2])
            assert a[0] == 0
            pos = min(len(z), pos + a[1])
            assert (o.is_finished(), o.blocks_decoded(), o.bytes_read_from_source(), o.can_collect()) == \
                   (g.is_finished(), g.blocks_decoded(), g.bytes_read_from_source(), g.can_collect())
            if o.is_finished() and o.can_collect() == 0:
                break
        assert o.is_finished() and g.is_finished() and o.calculated_checksum() == g.get_calculated_checksum(), name
        g.close()
    ctx.close()


def test_good_blocks_in_front_of_an_execution_error_keep_their_bytes():
    """a mutated corpus frame (tests/golden/regress, found by tools/dev/soak_batch.py): block 5 fails in sequence execution (a match that
    starts in front of everything, DictionaryTooSmall), block 4 lies in the same unit. What the frame produced ends with its last good block
    (include/zgpu.h, zgpu_batch_sync) — and those bytes are the reference's: the sweep used to skip every unit of a frame with such an error,
    which left the match bytes of the good blocks behind the first unit unresolved"""
    sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    import zgpu
    z = open(os.path.join(ROOT, "tests", "golden", "regress", "exec_error_behind_good_blocks.zst"), "rb").read()
    o = oracle.FrameDecoder()
    st, hl, _, _ = o.init(z)
    assert st == 0
    ost, _, _ = o.decode_blocks(z[hl:])
    assert ost == 53 and o.blocks_decoded() == 5
    want = o.held()[:314443]                                   # (behind them: what block 5 wrote before it failed)
    ctx = zgpu.Context(0)
    for src in (z, z + z, z[:6] + z[6:]):                        # alone, and twice in one submit
        b = zgpu.Batch(ctx, src)
        b.run()
        b.sync()
        for f in range(b.nframes):
            fi = b.frame_info(f)
            assert (fi.status, fi.bad_block, fi.out_size) == (53, 5, 314443)
            assert b.read(fi.out_base, fi.out_size) == want
        b.close()
    # the thin boundary (a frame decoded run by run): everything the reference's buffer holds after the Err is readable
    from test_gpu_thin_boundary import parse_frame_header, walk_blocks
    hl2, window, fcs, did, _ = parse_frame_header(z)
    blocks, _ = walk_blocks(z, hl2)
    f = zgpu.BlockFrame(ctx, window, fcs, did)
    f.submit(z, blocks)
    assert f.sync() == (5, 53)
    assert f.read(1 << 22, True) == o.held()                  # ... and behind them what block 5 wrote before it failed (zg_k_partial): 1448 bytes
    assert len(o.held()) == 315891
    f.close()
    ctx.close()


def test_offset_error_in_front_of_a_sequence_the_post_pass_rejects():
    """a mutated corpus frame cut behind its second block (tests/golden/regress, found by tools/dev/soak.py seed 41): in that block sequence
    1708 reaches in front of everything (DictionaryTooSmall) and a sequence behind the first 2048 asks for more literals than are left. The
    reference executes in order and meets the offset first; zg_k_seqpost rejects the later one, zg_k_exact looks at the sequences in front of
    it — and was told the rejected sequence's index in its PASS of 2048 instead of in the block, so it stopped short of 1708"""
    sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    import zgpu
    z = open(os.path.join(ROOT, "tests", "golden", "regress", "offset_error_in_front_of_a_rejected_sequence.zst"), "rb").read()
    ost, _ = oracle.FrameDecoder().decode_all(z, 1 << 22)
    assert ost == 53
    ctx = zgpu.Context(0)
    with pytest.raises(zgpu.ZgpuError) as e:
        ctx.decode_all(z, 1 << 22)
    assert e.value.status == ost
    g = zgpu.FrameDecoder(ctx)
    st, hl, _, _ = g.init(z)
    assert st == 0 and g.decode_blocks(z[hl:])[0] == ost and g.blocks_decoded() == 1
    g.close()
    s = zgpu.CStreamingDecoder(ctx, data=z)
    with pytest.raises(zgpu.ZgpuError) as e:
        s.read(1 << 22)
    assert e.value.status == ost
    s.close()
    ctx.close()


def test_pool_entries_that_are_runs_of_frames():
    """zgpu_pool_stage takes entries that are one frame or a run of frames (skippable ones in between): zgpu_pool_frame / zgpu_pool_read speak
    for the whole entry — until tools/dev/soak_pool.py they looked at its first frame only"""
    sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import oracle
    import zgpu
    from golden_io import read_pack
    pack = read_pack("decodecorpus.pack")
    names = sorted(k for k in pack if k.endswith(".zst"))[:12]
    skip = bytes([0x5A, 0x2A, 0x4D, 0x18, 5, 0, 0, 0]) + b"12345"
    bad = bytearray(pack[names[3]])
    bad[len(bad) // 2] ^= 0x10
    entries = [pack[names[0]], pack[names[1]] + skip + pack[names[2]], skip, pack[names[4]] + bytes(bad) + pack[names[5]],
               pack[names[6]] + pack[names[7]] + pack[names[8]]]
    pool = zgpu.Pool()
    pool.stage(entries)
    pool.run()
    for i, z in enumerate(entries):
        ost, oout = oracle.FrameDecoder().decode_all(z, 1 << 25)
        gpu, size, st = pool.frame(i)
        assert st == ost, (i, st, ost)
        if ost == 0:
            assert size == len(oout) and pool.read(i, size) == oout, i
    pool.close()


def test_padding_is_looked_at_before_the_states_are_initialised():
    """a mutated corpus block (tests/golden/regress, tools/dev/soak.py seed 204): a Repeat mode without a table to repeat AND a sequence
    bitstream whose last byte is 0. The reference checks the padding (sequence_section_decoder.rs:29-40) before it initialises the FSE states
    (fse_decoder.rs:33-35): ExtraPadding, not TableIsUninitialized"""
    sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    import zgpu
    z = open(os.path.join(ROOT, "tests", "golden", "regress", "padding_before_uninitialized_table.zst"), "rb").read()
    assert oracle.FrameDecoder().decode_all(z, 1 << 22)[0] == 44
    ctx = zgpu.Context(0)
    with pytest.raises(zgpu.ZgpuError) as e:
        ctx.decode_all(z, 1 << 22)
    assert e.value.status == 44
    ctx.close()


def test_partial_output_of_a_block_on_the_in_order_path():
    """a mutated frame (tests/golden/regress, tools/dev/soak_seqbits.py seed 204) one of whose blocks regenerates more than 128 KiB — the frame
    leaves the flatten path for zg_k_lz — and whose third block fails in sequence execution: what that block wrote before it failed is in
    the buffer like in the reference's (zg_k_partial rebuilds the positions from the records' exact fields, the position fields wrap there)"""
    sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    import zgpu
    from test_gpu_thin_boundary import parse_frame_header, walk_blocks
    z = open(os.path.join(ROOT, "tests", "golden", "regress", "seqbits_240.zst"), "rb").read()
    o = oracle.FrameDecoder()
    st, hl, _, _ = o.init(z)
    assert st == 0 and o.decode_blocks(z[hl:])[0] == 53 and o.blocks_decoded() == 2
    held = o.held()
    assert len(held) == 2326
    ctx = zgpu.Context(0)
    hl2, window, fcs, did, _ = parse_frame_header(z)
    blocks, _ = walk_blocks(z, hl2)
    f = zgpu.BlockFrame(ctx, window, fcs, did)
    f.submit(z, blocks)
    assert f.sync() == (2, 53)
    assert f.read(1 << 22, True) == held
    f.close()
    ctx.close()


@pytest.mark.parametrize("tool,args", [("soak_thin.py", ["500", "13"]), ("soak_batch.py", ["300", "14"]), ("soak_seqbits.py", ["300", "15"]),
                                       ("soak_headers.py", ["300", "16"]), ("soak_stream.py", ["400", "17"])])
def test_fixed_seed_runs_of_the_other_soaks(tool, args):
    """the thin boundary in random runs, many frames per submit, bit flips inside sequence bitstreams (six surfaces, everything held after an
    Err), header edits (seven surfaces), the io::Read surface in every mode: short runs with fixed seeds (tools/dev/final_check.sh runs them long)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dev", tool)] + args, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "disagreements 0" in r.stdout
