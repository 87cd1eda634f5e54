import sys, os, time, ctypes as C
sys.path.insert(0,'zstd-rs_amd'); sys.path.insert(0,'tools')
import zgdata, zgpu
d = zgdata.text_like(1000000000, seed=0xE9); z = zgdata.zstd_compress(d, level=3)
c = zgpu.Context(0); L = c.L
buf = C.create_string_buffer(len(d)); w = C.c_size_t()
for i in range(3):
    t0 = time.perf_counter(); st = L.zgpu_decode_all(c.h, z, len(z), buf, len(d), C.byref(w)); dt = time.perf_counter() - t0
    print("C-level decode_all: %.1f ms  %.2f GB/s st %d" % (dt * 1e3, len(d) / dt / 1e9, st))
for i in range(2):
    t0 = time.perf_counter(); b = c.prepare(z); t1 = time.perf_counter(); b.run(); b.sync(); t2 = time.perf_counter()
    st = L.zgpu_batch_read(b.h, 0, buf, len(d)); t3 = time.perf_counter(); b.close(); t4 = time.perf_counter()
    print("prepare %.1f  run+sync %.1f  read(D2H) %.1f  destroy %.1f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3))
