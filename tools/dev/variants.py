"""One data set, several engine variants (environment switches read when a Context is created): kernel times side by side.
usage: variants.py [size] [text|iso|blocks|many|isomany] -- VAR=VALUE[,VAR=VALUE] ...
(many: 64 MiB text frames, 16 distinct ones repeated up to `size` bytes, in one submit: one GPU's share of BASELINE config 4)"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import zgdata, zgpu

args = sys.argv[1:]
sep = args.index("--") if "--" in args else len(args)
size = int(args[0]) if sep > 0 else 1000000000
kind = args[1] if sep > 1 else "text"
variants = args[sep + 1:] or [""]
if kind == "blocks":     # single-block frames: every unit is a first unit, the sweep is one launch
    parts = [zgdata.text_like(128 << 10, seed=0x900 + i) for i in range(max(size >> 17, 1))]
    plain = b"".join(parts)
    z = b"".join(zgdata.zstd_compress(q) for q in parts)
    size = len(plain)
elif kind in ("many", "isomany"):
    parts = [zgdata.text_like(64 << 20, seed=0xE9 + i) for i in range(16)] if kind == "many" else [zgdata.iso_like(64 << 20, seed=0x150 + i) for i in range(16)]
    reps = max(size // (16 * (64 << 20)), 1)
    z = b"".join(zgdata.zstd_compress(q) for q in parts) * reps
    plain = None
    size = reps * 16 * (64 << 20)
    h = hashlib.sha256()
    for _ in range(reps):
        for q in parts:
            h.update(q)
    want = h.digest()
elif kind == "real":        # the real-text corpus of the image, stretched to `size` (tools/realtext.py load_tiled: bench.py's realtext1g)
    import realtext
    plain = realtext.load_tiled(size)[0]
    z = zgdata.zstd_compress(plain)
else:
    plain = zgdata.text_like(size) if kind == "text" else zgdata.iso_like(size)
    z = zgdata.zstd_compress(plain)
if plain is not None:
    want = hashlib.sha256(plain).digest()
    del plain
for v in variants:
    sets = [kv.split("=") for kv in v.split(",") if kv]
    for k, val in sets:
        os.environ[k] = val
    ctx = zgpu.Context(0, dev=True)
    b = ctx.prepare(z)
    assert b.parse_status == 0
    acc = {}
    for i in range(4):
        b.run(); b.sync()
        assert b.bad_status == 0, b.bad_status
        if i == 0:
            ok = hashlib.sha256(b.read(0, b.total_out)).digest() == want if not os.environ.get("ZGPU_SWEEP_MODE") else True
        else:
            for k, t in b.timings().items():
                acc[k] = acc.get(k, 0.0) + t / 3
    print("%-40s %s  %7.2f GB/s " % (v or "default", "OK " if ok else "BAD", size / acc["total"] / 1e6), {k: round(t, 3) for k, t in acc.items()}, flush=True)
    b.close(); ctx.close()
    for k, _ in sets:
        del os.environ[k]
