#!/usr/bin/env python3
"""bench.py — decompressed GB/s of the zstd block-decode path on MI355X.

A "step" is one pass of the hot path (FSE tables -> Huffman literals || FSE sequence chains -> scan -> LZ77 flatten ->
sweep) over everything staged on this rank's GPU: the compressed frames and the host-parsed block tables are already in
HBM when the timed region starts, the plaintext stays in HBM. The PCIe-inclusive rate (C ABI zgpu_decode_all, host buffer
in, host buffer out) is measured separately and reported as `e2e_GBps`; it is never `value`.

Workloads (BASELINE.json configs):
  enwik9like  configs[1]  enwik9.zst as ONE frame per GPU. $ZGPU_DATA/enwik9 (1e9 bytes) is compressed with libzstd -3 when
              present; else the stand-in text_like(1e9 B, seed 0xE9 + rank, V=14000) | libzstd -3 (ratio 3.19, 7630 blocks).
              N > 1: every rank decodes its own frame (a single frame does not shard: replicas, weak scaling).
  silesia12   configs[2]  12 independent frames ($ZGPU_DATA/silesia/* when present, else 12 synthetic frames of the Silesia
              sizes), sharded over the ranks in LPT order (strong scaling, bounded by the largest frame).
  blocks      configs[3]  16 x 64 MiB text frames (1 GiB of 128 KiB blocks, ratio 3.19) per GPU (weak scaling).
  blocks4b    variant 4b  the same plaintext as 2048 single-block frames per GPU.
  iso         configs[4]  iso_like frames, ratio 1.18 (Huffman-literal dominated), 8 x 64 MiB per GPU (weak scaling).

The frames go through the library's work queue (zgpu_pool: LPT order, one worker + engine per GPU); under
torch.distributed.run there is one process per GPU and each rank's pool holds its own GPU (frames -> ranks by the same LPT rule,
zgpu_dist.shard_frames). Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
SILESIA_SIZES = [51220480, 41458703, 33553445, 21606400, 10192446, 10085684, 9970564, 8474240, 7251944, 6627202, 6152192, 5345280]
SILESIA_NAMES = ["mozilla", "webster", "nci", "samba", "dickens", "osdb", "mr", "x-ray", "sao", "reymont", "ooffice", "xml"]


def build_workload(name, rank, world, size):
    """returns (description, [(plaintext, zst)] of the WHOLE job's frames for sharded workloads / of this rank for replicated
    ones, sharded?, data tag)"""
    import zgdata
    data_dir = os.environ.get("ZGPU_DATA")
    if name == "enwik9like":
        path = os.path.join(data_dir, "enwik9") if data_dir else None
        if path and os.path.exists(path):
            plain = open(path, "rb").read()
            return "enwik9 (%d B from $ZGPU_DATA) | libzstd %s -3, one frame per GPU" % (len(plain), zgdata.zstd_version()), [plain], False, "file"
        plain = zgdata.text_like(size, seed=0xE9 + rank)
        return ("enwik9-like single frame: text_like(%d B, seed 0xE9+rank, V=14000) | libzstd %s -3, one frame per GPU"
                % (size, zgdata.zstd_version())), [plain], False, "synthetic"
    if name == "silesia12":
        sdir = os.path.join(data_dir, "silesia") if data_dir else None
        if sdir and os.path.isdir(sdir) and len(os.listdir(sdir)) >= 12:
            plains = [open(os.path.join(sdir, f), "rb").read() for f in sorted(os.listdir(sdir))]
            return "Silesia (%d files from $ZGPU_DATA) | libzstd -3, one frame per file" % len(plains), plains, True, "file"
        plains = [zgdata.text_like(s, seed=0x51 + i) if i % 3 else zgdata.iso_like(s, seed=0x51 + i) for i, s in enumerate(SILESIA_SIZES)]
        return "Silesia-sized stand-in: 12 frames (text_like / iso_like, sizes of the 12 Silesia files) | libzstd -3", plains, True, "synthetic"
    if name == "blocks":
        plains = [zgdata.text_like(64 << 20, seed=0xE9 + 16 * rank + i) for i in range(16)]
        return "16 x 64 MiB text_like frames (1 GiB of 128 KiB blocks) per GPU | libzstd -3", plains, False, "synthetic"
    if name == "blocks4b":
        big = zgdata.text_like(256 << 20, seed=0xE9 + rank)
        plains = [big[i:i + (128 << 10)] for i in range(0, len(big), 128 << 10)]
        return "2048 single-block frames (128 KiB each) per GPU | libzstd -3", plains, False, "synthetic"
    if name == "iso":
        plains = [zgdata.iso_like(64 << 20, seed=0x150 + 8 * rank + i) for i in range(8)]
        return "8 x 64 MiB iso_like frames (ratio 1.18) per GPU | libzstd -3", plains, False, "synthetic"
    raise SystemExit("unknown workload " + name)


def _best_of(fn, n=3):
    best = None
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best


def cpu_baseline(zs, plain_len, cores):
    """CPU decoders timed on this box's host cores, same run: the oracle (a C port of ruzstd's decode path; ruzstd itself cannot
    be built here: no Rust toolchain) and libzstd (the C library the reference's Readme compares itself with), one thread and
    `cores` threads (one frame per thread), best of 3. zs: one compressed sample frame (bounded: this leg stays in seconds)."""
    import oracle
    import zgdata
    L = oracle.lib()
    zl = zgdata.libzstd()

    def oracle_one(out):
        d = L.zor_new()
        w = C.c_size_t()
        st = L.zor_decode_all(d, zs, len(zs), out, plain_len, C.byref(w))
        L.zor_free(d)
        assert st == 0 and w.value == plain_len

    def zstd_one(out):
        assert zl.ZSTD_decompress(out, plain_len, zs, len(zs)) == plain_len

    def threaded(fn, n):
        bufs = [C.create_string_buffer(plain_len) for _ in range(n)]

        def go():
            th = [threading.Thread(target=fn, args=(b,)) for b in bufs]      # ctypes releases the GIL during the call
            for t in th:
                t.start()
            for t in th:
                t.join()
        return go

    one = C.create_string_buffer(plain_len)
    t_o1 = _best_of(lambda: oracle_one(one))
    t_z1 = _best_of(lambda: zstd_one(one))
    t_on = _best_of(threaded(oracle_one, cores))
    t_zn = _best_of(threaded(zstd_one, cores))
    return {"value": round(plain_len / t_o1 / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": "%d-byte frame of the same workload through the oracle's decode_all (FrameDecoder::decode_all semantics), best of 3" % plain_len,
            "oracle_nt_GBps": round(cores * plain_len / t_on / 1e9, 4), "nt_cores": cores,
            "libzstd_1t_GBps": round(plain_len / t_z1 / 1e9, 4), "libzstd_nt_GBps": round(cores * plain_len / t_zn / 1e9, 4),
            "libzstd_version": zgdata.zstd_version(), "host_cores_available": os.cpu_count(),
            "note": "nt = one frame per thread on nt_cores threads; ruzstd itself is not buildable here (no rustc/cargo)"}


def other_workload(name, local_rank, steps=20):
    import zgdata
    import zgpu
    desc, plains, _, _ = build_workload(name, 0, 1, 0)
    zs = [zgdata.zstd_compress(p, level=3) for p in plains]
    pool = zgpu.Pool(devices=[local_rank])
    pool.stage(zs)
    for _ in range(2):
        pool.run()
    for k, p in enumerate(plains):
        gpu, size, st = pool.frame(k)
        assert st == 0 and size == len(p), (name, k, st, size)
        assert hashlib.sha256(pool.read(k, size)).digest() == hashlib.sha256(p).digest(), "GPU output differs (%s frame %d)" % (name, k)
    t0 = time.perf_counter()
    busy = 0.0
    for _ in range(steps):
        gms, _ = pool.run()
        busy += gms[0]
    dt = time.perf_counter() - t0
    pool.close()
    D = sum(len(p) for p in plains)
    return {"workload": desc, "plaintext_bytes": D, "compressed_bytes": sum(len(z) for z in zs), "frames": len(zs), "steps": steps,
            "GBps": round(D * steps / dt / 1e9, 3), "ms_per_step": round(dt / steps * 1e3, 3), "kernel_ms_per_step": round(busy / steps, 3)}


def e2e_rate(ctx, z, plain_len):
    """C ABI zgpu_decode_all: host buffer in, host buffer out (H2D + host block walk + kernels + D2H), pinned host buffers"""
    import torch
    src = torch.frombuffer(bytearray(z), dtype=torch.uint8).pin_memory()
    dst = torch.empty(plain_len, dtype=torch.uint8).pin_memory()
    w = C.c_size_t()

    def go():
        st = ctx.L.zgpu_decode_all(ctx.h, C.c_char_p(src.data_ptr()), len(z), C.c_void_p(dst.data_ptr()), plain_len, C.byref(w))
        assert st == 0 and w.value == plain_len
    go()
    return plain_len / _best_of(go) / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=250)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="enwik9like", choices=["enwik9like", "silesia12", "blocks", "blocks4b", "iso"])
    ap.add_argument("--size", type=int, default=int(os.environ.get("ZGPU_BENCH_SIZE", 1000000000)), help="plaintext bytes per GPU (enwik9like)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-other", action="store_true", help="skip the short runs of the other BASELINE.json configurations")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)

    import zgdata
    import zgpu
    import zgpu_dist
    desc, plains, sharded, data_tag = build_workload(args.workload, rank, world, args.size)
    zs = [zgdata.zstd_compress(p, level=3) for p in plains]
    job_lens = [len(z) for z in zs]
    mine = zgpu_dist.shard_frames(job_lens, world)[rank] if sharded else list(range(len(zs)))
    order, _, loads = zgpu.plan(job_lens, world if sharded else 1)
    pool = zgpu.Pool(devices=[local_rank])            # this rank's GPU behind the library's work queue
    t0 = time.perf_counter()
    pool.stage([zs[i] for i in mine])                 # host block walk + H2D: the submission, outside the timed region
    prep_s = time.perf_counter() - t0

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        pool.run()
    # parity gate on a warm run: the full plaintext of every frame must match the generator's bytes
    for k, i in enumerate(mine):
        gpu, size, st = pool.frame(k)
        assert st == 0 and size == len(plains[i]), (i, st, size)
        got = pool.read(k, size)
        assert hashlib.sha256(got).digest() == hashlib.sha256(plains[i]).digest(), "GPU output differs (frame %d)" % i
        del got

    kern = {}
    barrier()
    t0 = time.perf_counter()
    busy = 0.0
    for _ in range(args.steps):
        gms, _wall = pool.run()                        # blocks until this rank's GPU has finished the pass
        busy += gms[0]
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    D_mine, C_mine = sum(len(plains[i]) for i in mine), sum(len(zs[i]) for i in mine)
    if world > 1:
        tot = torch.tensor([float(D_mine), float(C_mine), busy], device="cuda", dtype=torch.float64)
        allt = [torch.zeros_like(tot) for _ in range(world)]
        dist.all_gather(allt, tot)
        D_job, C_job = sum(float(t[0]) for t in allt), sum(float(t[1]) for t in allt)
        per_gpu_busy = [float(t[2]) / args.steps for t in allt]
    else:
        D_job, C_job, per_gpu_busy = float(D_mine), float(C_mine), [busy / args.steps]
    value = D_job * args.steps / dt / 1e9

    out = None
    if rank == 0:
        # kernel times of one more pass on this rank, per kernel (HIP events on the engine's own streams)
        ctx = zgpu.Context(local_rank)
        b = ctx.prepare(b"".join(zs[i] for i in mine))
        for _ in range(2):
            b.run(); b.sync()
        acc = {}
        reps = 5
        for _ in range(reps):
            b.run(); b.sync()
            for k, v in b.timings().items():
                acc[k] = acc.get(k, 0.0) + v / reps
        kern = acc
        nblocks = b.nblocks
        b.close()
        dom = max(("tables", "huf", "seq", "seqpost", "scan", "lit", "flat", "sweep", "lz"), key=lambda k: kern[k])
        A = C_mine + D_mine            # algorithmic bytes of one pass: every compressed byte read once + every plaintext byte written once
        pipe = A / (kern["total"] / 1e3) / 1e9 if kern["total"] > 0 else 0.0
        ach_dom = A / (kern[dom] / 1e3) / 1e9 if kern[dom] > 0 else 0.0
        traffic, traffic_src = None, None
        try:   # HBM bytes from the committed rocprofv3 PMC passes; refused when the kernels changed since they were taken
            pm = json.load(open(os.path.join(ROOT, "profiles", "r02", "bench_pmc.json")))
            cur = hashlib.sha256(open(os.path.join(ROOT, "zstd-rs_amd", "csrc", "zg_kernels.hip"), "rb").read()).hexdigest()
            if pm.get("kernels_sha256") == cur and pm.get("workload") == args.workload and pm.get("plaintext_bytes") == D_mine:
                traffic = pm["pipeline_hbm_bytes_per_pass"]
                traffic_src = "profiles/r02/bench_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, per-kernel calibration; all kernels of one pass)"
            else:
                traffic_src = "profiles/r02/bench_pmc.json is stale for this build/workload: not reported"
        except Exception:
            pass
        out = {
            "metric": "decompressed_GB_per_s", "value": round(value, 4), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "u8", "data": data_tag,
            "config": {"workload": desc, "name": args.workload, "plaintext_bytes_job": int(D_job), "compressed_bytes_job": int(C_job),
                       "frames_job": len(zs) if sharded else len(zs) * world, "frames_this_gpu": len(mine), "blocks_this_gpu": nblocks,
                       "timed_region": "kernels only, inputs + block tables resident in HBM, output left in HBM; one pool.run() per step",
                       "queue": "zgpu_pool (LPT order, one worker + engine per GPU); ranks by zgpu_dist.shard_frames",
                       "host_prepare_s": round(prep_s, 4)},
            "roofline": {"bound": "hbm", "achieved": round(pipe, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(pipe / HBM_PEAK_GBS, 6),
                         "scope": "whole pipeline of one pass on one GPU: (C + D) / t_kernels (SURVEY 8d)",
                         "kernel": "zg_k_" + dom, "kernel_ms": round(kern[dom], 4), "achieved_dominant": round(ach_dom, 3),
                         "frac_dominant": round(ach_dom / HBM_PEAK_GBS, 6), "algorithmic_bytes": int(A), "traffic": traffic,
                         "traffic_source": traffic_src},
            "read_GBps": round(C_mine / (kern["total"] / 1e3) / 1e9, 3) if kern["total"] > 0 else 0.0,
            "kernel_ms": {k: round(v, 4) for k, v in kern.items()},
            "per_gpu_busy_ms": [round(x, 3) for x in per_gpu_busy],
            "lpt": {"loads_compressed_bytes": loads, "bound_speedup": round(sum(job_lens) / max(loads), 3) if sharded and max(loads) else float(world)},
        }
        if not args.no_e2e:
            i0 = mine[0]
            n = min(len(plains[i0]), 256 << 20) if args.workload == "enwik9like" else len(plains[i0])
            ze = zs[i0] if n == len(plains[i0]) else zgdata.zstd_compress(plains[i0][:n], level=3)
            out["e2e_GBps"] = round(e2e_rate(ctx, ze, n), 3)
            out["e2e_note"] = "C ABI zgpu_decode_all on a %d-byte frame: pinned host buffer in, pinned host buffer out (H2D + host walk + kernels + D2H), best of 3" % n
        ctx.close()
        if not args.no_cpu:
            # bounded CPU sample of the same workload: one frame of at most 64 MiB of plaintext
            i0 = mine[0]
            n = min(len(plains[i0]), 64 << 20)
            zc = zs[i0] if n == len(plains[i0]) else zgdata.zstd_compress(plains[i0][:n], level=3)
            out["cpu_baseline"] = cpu_baseline(zc, n, max(1, min(os.cpu_count() or 1, 64)))
        if not args.no_other and world == 1 and args.workload == "enwik9like":
            # the other BASELINE.json configurations on this GPU, 20 passes each (parity gate on every frame first): context for the
            # headline, not part of `value`
            pool.close(); pool = None
            out["other_workloads"] = {w: other_workload(w, local_rank) for w in ("silesia12", "blocks", "iso")}
        print(json.dumps(out), flush=True)
    if pool is not None:
        pool.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
