"""One GPU's share of BASELINE config 4 in one submit: 16 distinct 64 MiB text frames, repeated to `gib` GiB of output.
usage: big_submit.py [gib]"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import zgdata, zgpu
gib = int(sys.argv[1]) if len(sys.argv) > 1 else 8
plains = [zgdata.text_like(64 << 20, seed=0xE9 + i) for i in range(16)]
zs = [zgdata.zstd_compress(p) for p in plains]
digests = [hashlib.sha256(p).digest() for p in plains]
reps = gib * 16 // 16
blob = b"".join(zs) * reps
ctx = zgpu.Context(0, dev=True)
b = ctx.prepare(blob)
assert b.parse_status == 0
for _ in range(2):
    t0 = time.perf_counter(); b.run(); b.sync(); dt = time.perf_counter() - t0
assert b.bad_status == 0
D = b.total_out
ok = True
for k in (0, 5, 16 * reps - 1):                      # spot checks: first, one in the middle, last frame
    off = (k // 16) * (16 * (64 << 20)) + (k % 16) * (64 << 20)
    ok = ok and hashlib.sha256(b.read(off, 64 << 20)).digest() == digests[k % 16]
print("submit of %d frames, %.2f GiB out, %.2f GiB in: %.1f ms wall, %.1f GB/s; kernels %s; spot checks %s" % (16 * reps, D / 2**30, len(blob) / 2**30, dt * 1e3, D / dt / 1e9, {k: round(v, 2) for k, v in b.timings().items()}, "OK" if ok else "BAD"))
