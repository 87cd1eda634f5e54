"""The thin boundary of include/zgpu.h (zgpu_frame_begin / zgpu_blocks_submit / zgpu_sync / zgpu_read): the HOST side here — this
test — keeps the frame header and block header parse, exactly what ruzstd's FrameDecoder does before it calls
BlockDecoder::decode_block_content (frame_decoder.rs:319-375, block_decoder.rs:201-247, frame.rs:6-85), and hands block tables to
the device. Compared with the oracle on the reference's corpus, its dictionary fixtures, and on multi-submit / error cases."""
import hashlib
import struct

import pytest

import oracle
from golden_io import read_manifest, read_pack

pytestmark = pytest.mark.gpu


def parse_frame_header(z):
    """read_frame_header (frame.rs:6-85) in the caller's language: returns (header_len, window_size, content_size, dict_id, checksum flag)"""
    assert struct.unpack_from("<I", z, 0)[0] == 0xFD2FB528
    desc = z[4]
    p = 5
    single = (desc >> 5) & 1
    wd = 0
    if not single:
        wd = z[p]
        p += 1
    dl = [0, 1, 2, 4][desc & 3]
    did = int.from_bytes(z[p:p + dl], "little") if dl else 0
    p += dl
    fcs_flag = desc >> 6
    fl = [1 if single else 0, 2, 4, 8][fcs_flag]
    fcs = int.from_bytes(z[p:p + fl], "little") if fl else 0
    if fl == 2:
        fcs += 256
    p += fl
    if single:
        window = fcs
    else:
        exp, mant = wd >> 3, wd & 7
        base = 1 << (10 + exp)
        window = base + (base // 8) * mant
    return p, window, fcs, did, (desc >> 2) & 1


def walk_blocks(z, p):
    """read_block_header (block_decoder.rs:201-247) over the frame: [(src_off, src_len, type, last, raw_rle_size)], end offset"""
    out = []
    while True:
        h = z[p] | (z[p + 1] << 8) | (z[p + 2] << 16)
        last, ty, size = h & 1, (h >> 1) & 3, h >> 3
        p += 3
        clen = 1 if ty == 1 else size
        out.append((p, clen, ty, last, size if ty != 2 else 0))
        p += clen
        if last:
            return out, p


def decode_thin(ctx, z, chunk=0):
    import zgpu
    hl, window, fcs, did, has_ck = parse_frame_header(z)
    blocks, end = walk_blocks(z, hl)
    f = zgpu.BlockFrame(ctx, window, fcs, did)
    out = b""
    step = chunk or len(blocks)
    for i in range(0, len(blocks), step):
        f.submit(z, blocks[i:i + step])
        bad, st = f.sync()
        assert bad is None and st == 0, (bad, st)
        fin = i + step >= len(blocks)
        out += f.read(f.available(fin), fin)
    assert f.blocks_decoded() == len(blocks)
    ck = f.checksum()
    f.close()
    stored = struct.unpack_from("<I", z, end)[0] if has_ck else None
    return out, ck, stored


@pytest.fixture(scope="module")
def ctx():
    import zgpu
    c = zgpu.Context(0)
    yield c
    c.close()


def test_corpus_through_block_tables(ctx):
    pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
    for n in sorted(man):
        out, ck, stored = decode_thin(ctx, pack[n])
        assert len(out) == man[n]["size"] and hashlib.sha256(out).hexdigest() == man[n]["sha256"], n
        if stored is not None:
            assert ck == stored, n


def test_block_by_block_and_in_runs(ctx):
    """several submits per frame: the window, the offset history and the entropy tables carry over on the device"""
    pack, man = read_pack("synthetic.pack"), read_manifest("synthetic.json")
    for n in ("text_1m_l3.zst", "mixed_640k_l3.zst", "text_768k_l19.zst"):
        for chunk in (1, 3):
            out, ck, stored = decode_thin(ctx, pack[n], chunk)
            assert hashlib.sha256(out).hexdigest() == man[n]["sha256"], (n, chunk)


def test_dictionary_frames(ctx):
    """frames that name a dictionary: zgpu_add_dict + dict_id in zgpu_frame_begin (dict_test.rs:77-262)"""
    import zgpu
    pack, man = read_pack("dict_tests.pack"), read_manifest("dict_tests.json")
    did = ctx.add_dict(pack["dictionary"])
    names = sorted(n for n in pack if n.endswith(".zst"))[:60]
    for n in names:
        z = pack[n]
        hl, window, fcs, hdr_did, _ = parse_frame_header(z)
        assert hdr_did == did
        out, ck, stored = decode_thin(ctx, z)
        assert len(out) == man[n]["size"] and hashlib.sha256(out).hexdigest() == man[n]["sha256"], n
    # a dictionary nobody registered
    c2 = zgpu.Context(0)
    with pytest.raises(zgpu.ZgpuError) as e:
        zgpu.BlockFrame(c2, 1 << 20, 0, did)
    assert e.value.status == zgpu.E_DICT_NOT_PROVIDED
    c2.close()


def test_errors_name_the_block(ctx):
    """a corrupted block: zgpu_sync names it (frame-relative, over all submits) with the oracle's error leaf; blocks in front of it are readable"""
    import zgpu
    z = bytearray(read_pack("synthetic.pack")["text_1m_l3.zst"])
    hl, window, fcs, did, _ = parse_frame_header(z)
    blocks, end = walk_blocks(z, hl)
    assert len(blocks) >= 6
    victim = 4
    off, ln = blocks[victim][0], blocks[victim][1]
    for k in range(40, 60):
        z[off + ln - k] ^= 0x5A          # inside the sequences bitstream
    z = bytes(z)
    d = oracle.FrameDecoder()
    st, c, _, _ = d.init(z)
    assert st == 0
    ost, _, _ = d.decode_blocks(z[c:], oracle.STRAT_ALL)
    assert ost != 0
    good = d.blocks_decoded()
    f = zgpu.BlockFrame(ctx, window, fcs, did)
    f.submit(z, blocks[:2])
    assert f.sync() == (None, 0)
    f.submit(z, blocks[2:])
    bad, st = f.sync()
    assert bad == good and st == ost, (bad, st, good, ost)
    assert f.blocks_decoded() == good
    got = f.read(1 << 30, True)
    plain = oracle.decode_frame_all(read_pack("synthetic.pack")["text_1m_l3.zst"])[0]
    assert len(got) == good * (128 << 10) and got == plain[:len(got)]      # (every block of this frame but the last regenerates 128 KiB)
    f.close()
    # window limit (frame_decoder.rs:137-145)
    with pytest.raises(zgpu.ZgpuError) as e:
        zgpu.BlockFrame(ctx, 1 << 40, 0, 0)
    assert e.value.status == zgpu.E_WINDOW_SIZE_TOO_BIG


def test_execution_errors_cut_the_output_at_the_failing_block(ctx):
    """a sequence that cannot be executed (ExecuteSequencesError: zero offset, offset beyond what the buffer holds) is found by
    the LZ77 stages, after the sizes of all blocks were laid out: what zgpu_read hands out must still end with the last good block
    (what the reference's buffer holds when decode_block_content returns the error, sequence_execution.rs:5-54)"""
    import zgpu
    from test_exact_cpu import frame, lit_block, raw_block, seq_block
    for bad_block in (seq_block(1 << 22), seq_block(70000)):          # beyond everything produced so far
        z = frame(lit_block(3000), raw_block(5000, 3), lit_block(2500), bad_block, lit_block(700), lit_block(100, last=True))
        hl, window, fcs, did, _ = parse_frame_header(z)
        blocks, _ = walk_blocks(z, hl)
        o = oracle.FrameDecoder()
        st, c, _, _ = o.init(z)
        assert st == 0
        ost, _, _ = o.decode_blocks(z[c:], oracle.STRAT_ALL)
        assert ost != 0 and o.blocks_decoded() == 3
        want = oracle.decode_frame_all(frame(lit_block(3000), raw_block(5000, 3), lit_block(2500, last=True)))[0]   # the blocks in front of the failing one
        for cut in (len(blocks), 2):                                   # one submit; the failing block in a second submit
            f = zgpu.BlockFrame(ctx, window, fcs, did)
            f.submit(z, blocks[:cut])
            if cut < len(blocks):
                assert f.sync() == (None, 0)
                f.submit(z, blocks[cut:])
            bad, st = f.sync()
            assert (bad, st) == (3, ost)
            assert f.blocks_decoded() == 3
            got = f.read(1 << 20, True)
            assert len(got) == 3000 + 5000 + 2500 and got == want
            f.close()


def test_host_rejected_first_block_is_a_sticky_verdict(ctx):
    """a block the host checks reject (reserved type, block_decoder.rs:226-228) gives the same sticky verdict whether it is the first
    block of a submit or a later one: zgpu_blocks_submit returns OK, zgpu_sync names the block"""
    import zgpu
    z = read_pack("synthetic.pack")["text_1m_l3.zst"]
    hl, window, fcs, did, _ = parse_frame_header(z)
    blocks, _ = walk_blocks(z, hl)
    reserved = (blocks[2][0], blocks[2][1], 3, 0, 0)
    for first in (True, False):
        f = zgpu.BlockFrame(ctx, window, fcs, did)
        if first:
            f.submit(z, blocks[:2])
            assert f.sync() == (None, 0)
            f.submit(z, [reserved] + blocks[3:5])
        else:
            f.submit(z, blocks[:2] + [reserved] + blocks[3:5])
        bad, st = f.sync()
        assert (bad, st) == (2, zgpu.E_RESERVED_BLOCK), (first, bad, st)
        f.submit(z, blocks[3:5])                                       # the frame has failed: later submits are ignored
        assert f.sync() == (2, zgpu.E_RESERVED_BLOCK)
        assert f.blocks_decoded() == 2
        f.close()


def test_device_output_view(ctx):
    import zgpu
    z = read_pack("synthetic.pack")["text_1m_l3.zst"]
    hl, window, fcs, did, _ = parse_frame_header(z)
    blocks, _ = walk_blocks(z, hl)
    f = zgpu.BlockFrame(ctx, window, fcs, did)
    f.submit(z, blocks)
    assert f.sync() == (None, 0)
    ptr, n = f.device_output()
    assert ptr and n == read_manifest("synthetic.json")["text_1m_l3.zst"]["size"]
    f.close()


def test_dictionary_spliced_behind_drained_bytes(ctx):
    """the thin boundary's side of tests/test_gpu_exact.py::test_dictionary_spliced_behind_drained_bytes: the caller reads (drains) what three
    raw blocks produced down to the window, then submits a block whose match starts in front of what is left — the dictionary's tail, then
    the oldest byte still held (decode_buffer.rs:159-163). Same bytes as the oracle's FrameDecoder driven the same way."""
    import zgpu
    from test_exact_cpu import K, frame, raw_block, seq_block
    raw = read_pack("dict_tests.pack")["dictionary"]
    did = ctx.add_dict(raw)
    for off_extra in (1, 2, 3, 50):
        z = frame(raw_block(K, 1), raw_block(K, 2), raw_block(K, 3), seq_block(K + off_extra, last=True))
        o = oracle.FrameDecoder()
        assert o.add_dict(raw) == did
        st, c, _, _ = o.init(z)
        assert st == 0 and o.force_dict(did) == 0
        st, used, fin = o.decode_blocks(z[c:], oracle.STRAT_UPTO_BLOCKS, 3)
        want = o.collect()
        st, _, fin = o.decode_blocks(z[c + used:], oracle.STRAT_ALL)
        assert st == 0 and fin
        want += o.collect()
        hl, window, fcs, _, _ = parse_frame_header(z)
        blocks, end = walk_blocks(z, hl)
        f = zgpu.BlockFrame(ctx, window, fcs, did)
        f.submit(z, blocks[:3])
        assert f.sync() == (None, 0)
        got = f.read(f.available(False), False)
        assert len(got) == 2 * K                      # drained down to the window
        f.submit(z, blocks[3:])
        assert f.sync() == (None, 0)
        got += f.read(f.available(True), True)
        f.close()
        assert got == want, off_extra
