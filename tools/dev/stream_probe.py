"""where a streamed frame's time goes (zgpu_streaming_stats): python tools/dev/stream_probe.py [MiB of text] — 8 KiB / 1 MiB / 64 MiB io::copy loops,
one read of the whole frame; slice source in pinned memory"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, zgdata, zgpu
n = (int(sys.argv[1]) << 20) if len(sys.argv) > 1 else 1000000000
ra = (int(sys.argv[2]) << 20) if len(sys.argv) > 2 else 0
plain = zgdata.text_like(n, seed=0xE9)
z = zgdata.zstd_compress(plain, level=3)
ctx = zgpu.Context(0)
src = torch.frombuffer(bytearray(z), dtype=torch.uint8).pin_memory()
dst = torch.empty(n, dtype=torch.uint8).pin_memory()
for rep in range(1):
    for label, buf, ck in (("8KiB", 8192, True), ("8KiB-nohash", 8192, False), ("1MiB", 1 << 20, False), ("64MiB", 64 << 20, False), ("one-read", 0, False)):
        s = zgpu.CStreamingDecoder(ctx, data=(src.data_ptr(), len(z)), checksum=ck, read_ahead=ra)
        t0 = time.perf_counter()
        got = s.copy_to_sink(buf) if buf else s.read_into(dst.data_ptr(), n)
        dt = time.perf_counter() - t0
        st = s.stats()
        print("%-12s %6.1f ms %6.2f GB/s runs %d | worker: idle %.1f run %.1f land %.1f commit %.1f ringfull %.1f | reader: wait %.1f copy %.1f pull %.1f" % (
            label, dt * 1e3, n / dt / 1e9, st["runs"], st["us_worker_idle"] / 1e3, st["us_run"] / 1e3, st["us_land"] / 1e3, st["us_commit"] / 1e3,
            st["us_ring_full"] / 1e3, st["us_reader_wait"] / 1e3, st["us_reader_copy"] / 1e3, st["us_pull"] / 1e3), flush=True)
        print("             prepare %.1f run+sync %.1f | kernels: %s" % (st["us_prepare"] / 1e3, st["us_kernels"] / 1e3, " ".join("%s %.2f" % (k[2:], st[k] / 1e3) for k in st if k.startswith("k_"))), flush=True)
        assert got == n
        s.close()
