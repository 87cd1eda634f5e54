"""Quick look at the other BASELINE.json configs on one GPU (not the bench line): iso-like single frame, 12 Silesia-sized
frames, many 64 MiB frames. Prints kernel times and GB/s; every output is checked against the generator."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import zgdata, zgpu

def run(name, frames_plain):
    zs = [zgdata.zstd_compress(p) for p in frames_plain]
    blob = b"".join(zs)
    ctx = zgpu.Context(0, dev=True)
    b = ctx.prepare(blob)
    assert b.parse_status == 0
    for _ in range(2):
        b.run(); b.sync()
    assert b.bad_status == 0, b.bad_status
    out = b.read(0, b.total_out)
    assert hashlib.sha256(out).digest() == hashlib.sha256(b"".join(frames_plain)).digest(), name
    t = b.timings()
    D = sum(len(p) for p in frames_plain)
    print("%-28s frames %3d  D %6.1f MB  C %6.1f MB  %6.2f GB/s  " % (name, len(zs), D / 1e6, len(blob) / 1e6, D / t["total"] / 1e6),
          {k: round(v, 2) for k, v in t.items()}, flush=True)
    b.close(); ctx.close()

which = sys.argv[1:] or ["iso", "silesia", "many"]
if "iso" in which:
    run("iso_like 512 MiB single frame", [zgdata.iso_like(512 << 20)])
if "silesia" in which:
    sizes = [51220480, 41458703, 33553445, 21606400, 10192446, 10085684, 9970564, 8474240, 7251944, 6627202, 6152192, 5345280]
    run("silesia-sized 12 frames", [zgdata.text_like(s, seed=0x51 + i) if i % 3 else zgdata.iso_like(s, seed=0x51 + i) for i, s in enumerate(sizes)])
if "many" in which:
    run("16 x 64 MiB text frames", [zgdata.text_like(64 << 20, seed=0xE9 + i) for i in range(16)])
if "blocks" in which:
    run("2048 single-block frames", [zgdata.text_like(128 << 10, seed=0x900 + i) for i in range(2048)])
