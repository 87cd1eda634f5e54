"""Throughput with several submits in flight on one GPU (one engine + host thread each), against one at a time.
usage: two_in_flight.py [size] [kind] [engines] [passes]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import zgdata, zgpu
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000000
kind = sys.argv[2] if len(sys.argv) > 2 else "text"
neng = int(sys.argv[3]) if len(sys.argv) > 3 else 2
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 20
plain = zgdata.text_like(size) if kind == "text" else zgdata.iso_like(size)
z = zgdata.zstd_compress(plain)
ctxs = [zgpu.Context(0, dev=True) for _ in range(neng)]
bs = [c.prepare(z) for c in ctxs]
for b in bs:
    b.run(); b.sync(); assert b.bad_status == 0
def loop(b, n):
    for _ in range(n):
        b.run(); b.sync()
for k in (1, neng):
    th = [threading.Thread(target=loop, args=(bs[i], passes)) for i in range(k)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print("%d in flight: %d decodes in %.1f ms -> %.2f ms per decode, %.1f GB/s" % (k, k * passes, dt * 1e3, dt * 1e3 / (k * passes), size * k * passes / dt / 1e9), flush=True)
