"""Phase split of zg_k_flatten from a build with -DZG_PROFILE_FLAT (make EXTRA=-DZG_PROFILE_FLAT OUT=../libzgpu_prof.so):
ZGPU_LIB=zstd-rs_amd/libzgpu_prof.so python tools/dev/flat_phases.py [size]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ['ZGPU_DEBUG_TIMERS'] = '1'
import zgdata, zgpu
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256 << 20
d = zgdata.text_like(size); z = zgdata.zstd_compress(d)
for var in (sys.argv[2:] or ["1024"]):
    os.environ["ZGPU_FLAT_T"] = var
    c = zgpu.Context(0, dev=True); b = c.prepare(z)
    for _ in range(2): b.run(); b.sync()
    t = b.debug_timers(); tot = sum(t[:6]) or 1
    names = ["tile setup", "S1a+S1b (records, marks, prefix)", "S1c (walk, gathers issued)", "S2 (pointer jumping)", "S3a (root words, literals out)", "S3b (offsets out) + end"]
    print("T =", var, "thread-0 cycles per phase:", {n: round(x / tot, 3) for n, x in zip(names, t[:6])}, "total Mcycles", round(tot / 1e6, 1),
          "flat ms", round(b.timings()["flat"], 3), flush=True)
    b.close(); c.close()
