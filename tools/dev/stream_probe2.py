"""does what ran before in the process change the stream's download rate? (hardware queues are shared between streams)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, zgdata, zgpu
mode = sys.argv[1]
n = 1000000000
plain = zgdata.text_like(n, seed=0xE9)
z = zgdata.zstd_compress(plain, level=3)
if mode in ("pool_closed", "pool_alive", "pool_run_closed"):
    pool = zgpu.Pool(devices=[0])
    if mode == "pool_run_closed":
        pool.stage([z]); pool.run(); pool.run()
    if mode != "pool_alive":
        pool.close()
if mode == "torch_sync":
    torch.cuda.set_device(0); torch.cuda.synchronize(); x = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
ctx = zgpu.Context(0)
src = torch.frombuffer(bytearray(z), dtype=torch.uint8).pin_memory()
for rep in range(3):
    s = zgpu.CStreamingDecoder(ctx, data=(src.data_ptr(), len(z)), checksum=False)
    t0 = time.perf_counter(); got = s.copy_to_sink(64 << 20); dt = time.perf_counter() - t0
    st = s.stats()
    print("%-16s %6.1f ms %6.2f GB/s | worker run %.1f land %.1f | reader wait %.1f copy %.1f" % (mode, dt * 1e3, n / dt / 1e9, st["us_run"] / 1e3, st["us_land"] / 1e3, st["us_reader_wait"] / 1e3, st["us_reader_copy"] / 1e3), flush=True)
    s.close()
