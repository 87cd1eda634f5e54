"""kernel times of the 1e9-byte frame with the library ZGPU_LIB names (a variant build of tools/dev/mkvariant.sh) — the plaintext is checked"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import zgdata, zgpu
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000000
plain = zgdata.text_like(size)
z = zgdata.zstd_compress(plain)
want = hashlib.sha256(plain).digest()
ctx = zgpu.Context(0)
b = ctx.prepare(z)
acc = {}
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for i in range(n + 1):
    b.run(); b.sync()
    assert b.bad_status == 0
    if i == 0:
        ok = hashlib.sha256(b.read(0, b.total_out)).digest() == want
    else:
        for k, t in b.timings().items():
            acc[k] = acc.get(k, 0.0) + t / n
print(os.environ.get("ZGPU_LIB", "libzgpu.so"), "OK" if ok else "BAD", {k: round(t, 3) for k, t in acc.items()}, flush=True)
