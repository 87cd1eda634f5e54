#!/usr/bin/env python3
"""bench.py — decompressed GB/s of the zstd block-decode path on MI355X.

A "step" is one pass of the hot path (tables -> Huffman literals -> FSE sequences -> scan -> literal copy -> LZ77
match copy) over one resident batch: the compressed frame(s) and the host-parsed block table are already in HBM when
the timed region starts, the plaintext stays in HBM (the PCIe-inclusive rate is reported separately in DESIGN.md).

Workload (BASELINE.json configs[1]): enwik9.zst as ONE frame. enwik9 is not on the box and there is no network, so the
stand-in of SURVEY.md Appendix C is used: text_like(1e9 bytes, seed 0xE9, V=14000) compressed by libzstd -3
(ratio ~3.19, ~7630 blocks of 128 KiB, window 2 MiB). With --gpus N (one process per GPU under torch.distributed.run)
every rank decodes its own frame (seed 0xE9 + rank): independent frames shard with no data-path collective, so
scaling is weak and `value` is the whole-job aggregate.

Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def cpu_baseline(z, plain_len, budget_s=20.0):
    """the CPU oracle (a C port of ruzstd's decode path) timed on this box's host cores, 1 thread, on a bounded sample"""
    import oracle
    import zgdata
    L = oracle.lib()
    out = C.create_string_buffer(plain_len)
    w = C.c_size_t()
    d = L.zor_new()
    t0 = time.perf_counter()
    st = L.zor_decode_all(d, z, len(z), out, plain_len, C.byref(w))
    dt = time.perf_counter() - t0
    L.zor_free(d)
    assert st == 0 and w.value == plain_len
    res = {"value": round(plain_len / dt / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
           "sample": "%d-byte text_like frame (zstd -3), whole frame through the oracle's decode_all, 1 run" % plain_len}
    try:  # context only: the C libzstd the reference's Readme compares itself against
        zl = zgdata.libzstd()
        t0 = time.perf_counter()
        n = zl.ZSTD_decompress(out, plain_len, z, len(z))
        dt = time.perf_counter() - t0
        if n == plain_len:
            res["libzstd_1t_GBps"] = round(plain_len / dt / 1e9, 4)
            res["libzstd_version"] = zgdata.zstd_version()
    except Exception:
        pass
    res["host_cores_available"] = os.cpu_count()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=int(os.environ.get("ZGPU_BENCH_SIZE", 1000000000)), help="plaintext bytes per GPU")
    ap.add_argument("--kind", default="text", choices=["text", "iso"])
    ap.add_argument("--no-cpu", action="store_true")
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
    plain = zgdata.text_like(args.size, seed=0xE9 + rank) if args.kind == "text" else zgdata.iso_like(args.size, seed=0x150 + rank)
    z = zgdata.zstd_compress(plain, level=3)
    ctx = zgpu.Context(local_rank)
    t0 = time.perf_counter()
    batch = ctx.prepare(z)          # host block walk + H2D: the submission, outside the timed region
    prep_s = time.perf_counter() - t0
    assert batch.parse_status == 0

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.run()
        batch.sync()
    assert batch.bad_status == 0, batch.bad_status
    # parity gate on the warm run: full plaintext must match (checksum of the whole output vs the generator's bytes)
    import hashlib
    got = batch.read(0, batch.total_out)
    assert batch.total_out == len(plain) and hashlib.sha256(got).digest() == hashlib.sha256(plain).digest(), "GPU output differs"
    del got

    kern = {k: 0.0 for k in ("tables", "huf", "seq", "seqpost", "scan", "lit", "flat", "sweep", "lz", "total")}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.run()
        batch.sync()               # hipStreamSynchronize on the engine's stream
        for k, v in batch.timings().items():
            kern[k] += v
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    for k in kern:
        kern[k] /= args.steps

    D, Cb = len(plain), len(z)
    value = world * D * args.steps / dt / 1e9
    dom = max(("tables", "huf", "seq", "seqpost", "scan", "lit", "flat", "sweep", "lz"), key=lambda k: kern[k])
    # algorithmic bytes of one pass: every compressed byte read once + every plaintext byte written once
    achieved = (Cb + D) / (kern[dom] / 1e3) / 1e9 if kern[dom] > 0 else 0.0
    # HBM bytes of the dominant kernel from the committed rocprofv3 PMC passes of this same command (profiles/r01/):
    # FETCH_SIZE and WRITE_SIZE need separate profiler passes, so they cannot be read live here
    traffic, traffic_src = None, None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r01", "bench_1e9_pmc.json")))
        if args.kind == "text" and D == 1000000000:
            for kname, rec in pm["kernels"].items():
                if kname.split("<")[0] == "zg_k_" + dom:
                    traffic = rec["hbm_bytes_per_launch_corrected"]
                    traffic_src = "profiles/r01/bench_1e9_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, x2 fetch correction calibrated on zg_k_calib_copy)"
    except Exception:
        pass
    out = {
        "metric": "decompressed_GB_per_s", "value": round(value, 4), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "enwik9-like single frame: text_like(%d B, seed 0xE9+rank, V=14000) | libzstd %s -3, one frame per GPU"
                   % (D, zgdata.zstd_version()) if args.kind == "text" else "iso_like(%d B) | libzstd -3, one frame per GPU" % D,
                   "plaintext_bytes": D, "compressed_bytes": Cb, "blocks": batch.nblocks, "frames_per_gpu": 1,
                   "timed_region": "kernels only, inputs + block table resident in HBM, output left in HBM",
                   "host_prepare_s": round(prep_s, 4)},
        "roofline": {"bound": "hbm", "kernel": "zg_k_" + dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes": Cb + D, "kernel_ms": round(kern[dom], 4),
                     "pipeline_achieved": round((Cb + D) / (kern["total"] / 1e3) / 1e9, 3) if kern["total"] > 0 else 0.0},
        "kernel_ms": {k: round(v, 4) for k, v in kern.items()},
    }
    if rank == 0:
        if not args.no_cpu:
            # bounded CPU sample: at most ~256 MiB of the same workload
            n = min(D, 256 << 20)
            zs = z if n == D else zgdata.zstd_compress(plain[:n], level=3)
            out["cpu_baseline"] = cpu_baseline(zs, n)
        print(json.dumps(out), flush=True)
    batch.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
