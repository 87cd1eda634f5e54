"""(GPU) zgpu_decode_all host to host, by frame size and by the kind of host memory the caller hands over: pageable fresh (np.empty per call:
first touch inside the call), pageable reused (touched), pinned (torch pin_memory) source and destination.   usage: latency.py [max MiB]"""
import ctypes as C, sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import zgpu, zgdata
top = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = zgpu.Context(0)
L = ctx.L
def call(src_ptr, n_src, dst_ptr, cap):
    w = C.c_size_t()
    t = time.perf_counter()
    st = L.zgpu_decode_all(ctx.h, C.cast(C.c_void_p(src_ptr), C.c_char_p), n_src, C.c_void_p(dst_ptr), cap, C.byref(w))
    dt = time.perf_counter() - t
    assert st == 0 and w.value == cap, (st, w.value)
    return dt
def med(f, k):
    ts = sorted(f() for _ in range(k))
    return ts[len(ts) // 2]
for n in (4096, 131072, 1 << 20, 16 << 20, 64 << 20, 256 << 20, 1000000000):
    if n > top << 20: break
    data = zgdata.text_like(n, seed=0x11)
    z = zgdata.zstd_compress(data)
    k = 15 if n <= (64 << 20) else 5
    src = np.frombuffer(z, dtype=np.uint8)
    def fresh():
        out = np.empty(n, dtype=np.uint8)
        return call(src.ctypes.data, len(z), out.ctypes.data, n)
    out = np.zeros(n, dtype=np.uint8)
    call(src.ctypes.data, len(z), out.ctypes.data, n); assert out.tobytes() == data
    def reused():
        return call(src.ctypes.data, len(z), out.ctypes.data, n)
    ps = torch.empty(len(z), dtype=torch.uint8, pin_memory=True); ps.numpy()[:] = src
    pd = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    def pinned():
        return call(ps.data_ptr(), len(z), pd.data_ptr(), n)
    def pinned_dst():
        return call(src.ctypes.data, len(z), pd.data_ptr(), n)
    r = [med(f, k) for f in (fresh, reused, pinned_dst, pinned)]
    assert bytes(pd.numpy()) == data
    print("decode_all %10d B (%9d compressed): fresh pageable %8.3f ms %6.2f GB/s | reused pageable %8.3f ms %6.2f | pinned dst %8.3f ms %6.2f | pinned both %8.3f ms %6.2f GB/s" %
          (n, len(z), r[0] * 1e3, n / r[0] / 1e9, r[1] * 1e3, n / r[1] / 1e9, r[2] * 1e3, n / r[2] / 1e9, r[3] * 1e3, n / r[3] / 1e9), flush=True)
