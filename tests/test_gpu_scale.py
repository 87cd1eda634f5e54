"""GPU parity at the sizes of BASELINE.json's configs: the generator's bytes are the ground truth (sha256 of
every frame), the oracle checks a subsample, and the size-independent property is the frame checksum (XXH64 of the
plaintext, frame_decoder.rs:263-270) which every frame carries. Everything goes through the C ABI."""
import hashlib
import os
import sys

import pytest

import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _sha(b):
    return hashlib.sha256(b).digest()


@pytest.fixture(scope="module")
def ctx():
    import zgpu
    c = zgpu.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def text1g():
    import zgdata
    plain = zgdata.text_like(1000000000, seed=0xE9)
    return plain, zgdata.zstd_compress(plain)


def _check_batch(ctx, zs, plains, oracle_on=()):
    b = ctx.prepare(b"".join(zs))
    assert b.parse_status == 0 and b.nframes == len(zs)
    b.run()
    b.sync()
    assert b.bad_status == 0, (b.bad_frame, b.bad_status)
    assert b.total_out == sum(len(p) for p in plains)
    for f, p in enumerate(plains):
        got = b.frame_bytes(f)
        assert len(got) == len(p) and _sha(got) == _sha(p), f
        if f in oracle_on:
            ref, _ = oracle.decode_frame_all(zs[f])
            assert ref == got, f
    b.close()


def test_silesia_sized_12_frames_one_submit(ctx):
    """config 3: 12 uneven frames (sizes of the Silesia files, text-like and literal-heavy content mixed), one submit"""
    import zgdata
    sizes = [51220480, 41458703, 33553445, 21606400, 10192446, 10085684, 9970564, 8474240, 7251944, 6627202, 6152192, 5345280]
    plains = [zgdata.text_like(s, seed=0x51 + i) if i % 3 else zgdata.iso_like(s, seed=0x51 + i) for i, s in enumerate(sizes)]
    zs = [zgdata.zstd_compress(p) for p in plains]
    _check_batch(ctx, zs, plains, oracle_on=(10, 11))


def test_16384_single_block_frames(ctx):
    """config 4b: tens of thousands of block-independent frames in one submit — every frame is a direct unit, resolved to bytes by the
    flatten itself (2048 distinct frames, eight times each: 2 GiB of plaintext; every frame's bytes are compared, three with the oracle)"""
    import zgdata
    big = zgdata.text_like(256 << 20, seed=0x4B)
    plains = [big[i:i + (128 << 10)] for i in range(0, len(big), 128 << 10)]
    zs = [zgdata.zstd_compress(p) for p in plains]
    assert len(zs) == 2048
    for f in (0, 1000, 2047):
        assert oracle.decode_frame_all(zs[f])[0] == plains[f]
    b = ctx.prepare(b"".join(zs) * 8)
    assert b.parse_status == 0 and b.nframes == 16384
    b.run()
    b.sync()
    assert b.bad_status == 0, (b.bad_frame, b.bad_status)
    assert b.total_out == 8 * len(big)
    want = [hashlib.sha256(p).digest() for p in plains]
    for rep in range(8):
        out = b.read(rep * len(big), len(big))
        for f in range(2048):
            assert hashlib.sha256(out[f << 17:(f + 1) << 17]).digest() == want[f], (rep, f)
    b.close()


def test_640_multi_block_frames(ctx):
    """many more multi-unit frames than the device holds workgroups for: every sweep step serves hundreds of frames"""
    import zgdata
    plains = [zgdata.text_like(640 << 10, seed=0x600 + i) for i in range(640)]
    zs = [zgdata.zstd_compress(p) for p in plains]
    os.environ["ZGPU_UNIT_BLOCKS"] = "2"       # 3 units (sweep steps) per frame
    try:
        import zgpu
        # the direct units (every frame's first) by the kernel of the pointer-mode units, and — what a submit of thousands of frames gets by
        # itself — by zg_k_flatten4 on the second stream BESIDE the pointer-mode units' kernel, in its small and its large shape
        for flat4 in ("0", "6", "1"):
            os.environ["ZGPU_FLAT4"] = flat4
            c = zgpu.Context(0, dev=True)
            _check_batch(c, zs, plains, oracle_on=(7,) if flat4 == "0" else ())
            c.close()
    finally:
        del os.environ["ZGPU_UNIT_BLOCKS"]
        os.environ.pop("ZGPU_FLAT4", None)


def test_iso_like_256mib_frame(ctx):
    """config 5: literal-heavy (ratio 1.18) single frame: 4-stream Huffman literals of 128 KiB blocks, almost no sequences"""
    import zgdata
    plain = zgdata.iso_like(256 << 20, seed=0x150)
    _check_batch(ctx, [zgdata.zstd_compress(plain)], [plain])


def test_1e9_frame_and_checksum(ctx, text1g):
    """config 2 at full size: the enwik9-like frame, bit-exact, and its content checksum through the FrameDecoder surface"""
    import zgpu
    plain, z = text1g
    _check_batch(ctx, [z], [plain])
    d = zgpu.FrameDecoder(ctx)
    st, c, _, _ = d.reset(z)
    assert st == 0
    st, used, fin = d.decode_blocks(z[c:], zgpu.STRAT_ALL)
    assert st == 0 and fin and d.is_finished()
    h = hashlib.sha256()
    while True:
        chunk = d.read(64 << 20)
        if not chunk:
            break
        h.update(chunk)
    assert h.digest() == _sha(plain)
    assert d.get_checksum_from_data() == d.get_calculated_checksum() is not None
    d.close()


def test_ramped_units_chain_beside_flatten(ctx, monkeypatch):
    """the measurement switch ZGPU_RAMP (off by default: it measured slower): one long frame in units that grow along it, the
    flatten raising a flag per unit and the sweep chain polling it from a third stream — results must not change"""
    import zgdata
    plain = zgdata.text_like(200 << 20, seed=0x7A)
    z = zgdata.zstd_compress(plain)
    import zgpu
    for ramp in ("50", "90"):
        monkeypatch.setenv("ZGPU_RAMP", ramp)    # (read when the engine is created: a context per setting)
        c = zgpu.Context(0, dev=True)
        _check_batch(c, [z], [plain])
        c.close()
    monkeypatch.delenv("ZGPU_RAMP")
    _check_batch(ctx, [z], [plain])


def test_more_than_4gib_in_one_submit(ctx, text1g):
    """5 frames of 1e9 bytes in one submit: output positions beyond 2^32, flatten scratch beyond 16 GiB"""
    plain, z = text1g
    b = ctx.prepare(z * 5)
    b.run()
    b.sync()
    assert b.bad_status == 0 and b.total_out == 5 * len(plain)
    want = _sha(plain)
    for f in range(5):
        fi = b.frame_info(f)
        assert fi.out_base == f * len(plain) and fi.out_size == len(plain)
        assert _sha(b.read(fi.out_base, fi.out_size)) == want, f
    b.close()


def test_single_frame_beyond_2gib(ctx):
    """one frame of 2.6e9 bytes: frame-relative positions beyond 2^31"""
    import zgdata
    plain = zgdata.text_like(2600000000, seed=0x26)
    z = zgdata.zstd_compress(plain)
    b = ctx.prepare(z)
    b.run()
    b.sync()
    assert b.bad_status == 0 and b.total_out == len(plain)
    h = hashlib.sha256()
    for off in range(0, len(plain), 1 << 30):
        h.update(b.read(off, min(1 << 30, len(plain) - off)))
    assert h.digest() == _sha(plain)
    b.close()


def test_pool_queue_one_gpu(ctx):
    """the library's work queue (zgpu_pool) on the one GPU of this box: decode_all over frames of very different sizes,
    and the staged form the bench uses"""
    import zgdata
    import zgpu
    import zgpu_dist
    plains = [zgdata.text_like(n, seed=0x700 + i) for i, n in enumerate((90 << 20, 1 << 20, 300, 17 << 20, 5 << 20, 0, 70 << 20))]
    zs = [zgdata.zstd_compress(p) for p in plains]
    skip = bytes([0x50, 0x2A, 0x4D, 0x18, 3, 0, 0, 0, 1, 2, 3])
    blob = zs[0] + skip + b"".join(zs[1:])
    pool = zgpu.Pool(devices=[0])
    assert pool.n_gpus == 1
    out = pool.decode_all(blob, sum(len(p) for p in plains) + 16)
    assert _sha(out) == _sha(b"".join(plains))
    with pytest.raises(zgpu.ZgpuError) as e:
        pool.decode_all(blob, 1000)
    assert e.value.status == zgpu.E_TARGET_TOO_SMALL
    with pytest.raises(zgpu.ZgpuError):
        pool.decode_all(blob[:-7], 1 << 30)       # the last frame is truncated: the error of the walk wins
    pool.stage(zs)
    for _ in range(2):
        gms, wall = pool.run()
        assert len(gms) == 1 and gms[0] > 0 and wall >= gms[0] * 0.5
    for i, p in enumerate(plains):
        gpu, size, st = pool.frame(i)
        assert (gpu, size, st) == (0, len(p), 0)
        assert pool.read(i, size) == p
    pool.close()
    fn, pool2 = zgpu_dist.gpu_decode_fn(0)
    local = zgpu_dist.decode_sharded(zs, fn, 0, 1)
    assert [local[i] for i in range(len(zs))] == plains
    pool2.close()


def test_config4_share_128_x_64mib_one_submit(ctx):
    """one GPU's share of BASELINE config 4: 128 frames of 64 MiB (8 GiB of 128 KiB blocks) in ONE submit — 16 distinct frames x 8;
    every frame's bytes against its plaintext. Each frame is two units: a direct one (resolved by the flatten itself) and one that
    goes through scratch + sweep."""
    import numpy as np
    import zgdata
    plains = [zgdata.text_like(64 << 20, seed=0xE9 + i) for i in range(16)]
    zs = [zgdata.zstd_compress(p) for p in plains]
    refs = [np.frombuffer(p, dtype=np.uint8) for p in plains]
    b = ctx.prepare(b"".join(zs * 8))
    assert b.parse_status == 0 and b.nframes == 128
    b.run()
    b.sync()
    assert b.bad_status == 0, (b.bad_frame, b.bad_status)
    assert b.total_out == 128 * (64 << 20)
    modes = [u[4] for u in b.units()]
    assert sum(1 for m in modes if m & 2) == 128 and sum(1 for m in modes if m == 0) >= 128
    for f in range(128):
        got = np.frombuffer(b.frame_bytes(f), dtype=np.uint8)
        assert np.array_equal(got, refs[f % 16]), f
    b.close()


def test_pool_over_all_visible_gpus():
    """zgpu_pool_create(0): every visible GPU gets an engine and a worker thread; frames are spread over them by the queue (LPT)
    and come back in input order. With one GPU visible the placement assertion is skipped, the bytes are checked all the same."""
    import zgdata
    import zgpu
    pool = zgpu.Pool(n_gpus=0)
    n = pool.n_gpus
    assert n >= 1
    plains = [zgdata.text_like((1 << 20) + 4099 * i, seed=0x700 + i) for i in range(4 * n + 3)]
    zs = [zgdata.zstd_compress(p) for p in plains]
    pool.stage(zs)
    gms, wall = pool.run()
    assert len(gms) == n
    gpus = set()
    for k, p in enumerate(plains):
        gpu, size, st = pool.frame(k)
        gpus.add(gpu)
        assert st == 0 and size == len(p)
        assert pool.read(k, size) == p
    if n > 1:
        assert len(gpus) == n, gpus                      # 4n + 3 frames of similar size: LPT gives every GPU some
    kern, D, Cb, nb = pool.timings(0)
    assert kern["total"] > 0 and D > 0 and Cb > 0 and nb > 0
    # decode_all over the same pool: more than two jobs per GPU (the output of a job leaves the device when the job is done)
    big = [zgdata.text_like(110 << 20, seed=0x800 + i) for i in range(3 * n + 1)]
    bz = [zgdata.zstd_compress(p) for p in big]
    skip = b"\x50\x2a\x4d\x18" + (5).to_bytes(4, "little") + b"hello"          # a skippable frame in between (frame.rs:15-23)
    nofcs = zgdata.zstd_compress(plains[0], content_size=False)                # and a frame that does not declare its size
    blob = bz[0] + skip + b"".join(bz[1:])
    want = b"".join(big)
    out = pool.decode_all(blob, len(want))
    assert _sha(out) == _sha(want)
    out = pool.decode_all(blob + nofcs, len(want) + len(plains[0]))
    assert _sha(out) == _sha(want + plains[0])
    with pytest.raises(zgpu.ZgpuError) as e:
        pool.decode_all(blob, len(want) - 1)
    assert e.value.status == zgpu.E_TARGET_TOO_SMALL
    pool.close()


def test_pool_decode_all_places_jobs_around_a_lying_content_size(ctx, monkeypatch):
    """Frame_Content_Size is never checked by FrameDecoder::decode_all (frame_decoder.rs:541-577): a frame that declares fewer (or
    more) bytes than it holds still decodes, and what follows it lands right behind what it really produced. The queue places a
    job straight into the caller's buffer when its declared size turns out right — the jobs around a lying one must keep their bytes
    while everything is moved to its final place (understated first frame: its download is longer than the gap left for it;
    overstated: shorter)."""
    import zgdata
    import zgpu
    monkeypatch.setenv("ZGPU_DA_FLOOR_MB", "1")                     # every frame below is a job of its own (> 1 MiB of input each)
    plains = [zgdata.text_like((5 << 20) + 4096 * i, seed=0x7700 + i) for i in range(5)]
    zs = [zgdata.zstd_compress(p) for p in plains]
    want = b"".join(plains)

    def lie(z, delta):                                             # Frame_Header: magic(4) descriptor(1) window(1) FCS(4): frame.rs:6-85
        assert (z[4] >> 6) == 2 and not (z[4] >> 5) & 1 and not z[4] & 3, "expected a 4-byte FCS behind a window descriptor"
        fcs = int.from_bytes(z[6:10], "little") + delta
        return z[:6] + fcs.to_bytes(4, "little") + z[10:]
    pool = zgpu.Pool(devices=[0], dev=True)
    for which, delta in ((0, -100000), (0, +100000), (2, -70000), (4, -5)):
        blob = b"".join(lie(z, delta) if k == which else z for k, z in enumerate(zs))
        out = pool.decode_all(blob, len(want) + (1 << 20))
        assert len(out) == len(want) and _sha(out) == _sha(want), (which, delta)
    pool.close()
