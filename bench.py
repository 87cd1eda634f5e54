#!/usr/bin/env python3
"""bench.py — decompressed GB/s of the zstd block-decode path on MI355X.

A "step" is one pass of the hot path (FSE tables -> Huffman literals || FSE sequence chains -> scan -> LZ77 flatten ->
sweep) over everything staged on the GPUs: the compressed frames and the host-parsed block tables are already in HBM when
the timed region starts, the plaintext stays in HBM. The PCIe-inclusive rate (C ABI zgpu_pool_decode_all, pinned host buffer
in, pinned host buffer out, uploads / kernels / downloads overlapped) is measured separately and reported as `e2e_GBps`; it is
never `value`.

Workloads (BASELINE.json configs). Sizes are per GPU; frames beyond the distinct ones are repeats of them (host
generation and compression stay in seconds), every frame's output is compared with its plaintext before anything is timed:
  enwik9like  configs[1]  enwik9.zst as ONE frame per GPU. $ZGPU_DATA/enwik9 (1e9 bytes) is compressed with libzstd -3 when
              present; else the stand-in text_like(1e9 B, seed 0xE9 + gpu, V=14000) | libzstd -3 (ratio 3.19, 7630 blocks).
              N > 1: every GPU decodes its own frame (a single frame does not shard: replicas, weak scaling).
  realtext1g  the same real text stretched to 1e9 bytes (its 4 MiB pieces in seeded permutations, tools/realtext.py load_tiled: no match reaches
              a piece's earlier copy through a 2 MiB window) as ONE frame: real-text statistics at enwik9's size. $ZGPU_DATA/enwik9 replaces
              the stand-in of `enwik9like` when it exists; this workload is the real-data line when it does not.
  realtext    a real-text single frame per GPU: source code and documentation found in this image (tools/realtext.py: 393 MB from the
              directories listed in tools/realtext_manifest.json — the ones every box of the pool holds with the same bytes; a directory
              that differs is skipped and named) | libzstd -3: what the stand-in is a stand-in for.
  silesia12   configs[2]  12 independent frames ($ZGPU_DATA/silesia/* when present, else 12 synthetic frames of the Silesia
              sizes), sharded over the GPUs in LPT order (strong scaling, bounded by the largest frame).
  blocks      configs[3]  128 x 64 MiB text frames = 8 GiB of 128 KiB blocks (ratio 3.19) per GPU: one GPU's share of the 64 GiB
              run (16 distinct frames x 8). Weak scaling.
  blocks4b    variant 4b  65536 single-block frames (128 KiB each, 8 GiB: one GPU's share of 524288) per GPU (2048 distinct x 32).
  iso         configs[4]  128 x 64 MiB iso_like frames, ratio 1.18 (Huffman-literal dominated), 8 GiB per GPU (8 distinct x 16).

--gpus N: under torch.distributed.run (WORLD_SIZE set) there is one process per GPU and each rank's pool holds its own GPU
(frames -> ranks by the LPT rule of the library's queue, zgpu_dist.shard_frames); without it ONE process drives N engines
through the library's work queue (zgpu_pool_create(N): one worker thread + engine per GPU). Prints ONE JSON line (rank 0).
The timed region lasts at least --min-seconds whatever --steps says: a step then consists of `passes_per_step` passes.
"""
import argparse
import ctypes as C
import hashlib
import json
import math
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
KERNELS = ("tables", "huf", "seq", "seqpost", "scan", "lit", "flat", "sweep", "lz")
PROFILE_DIR = os.path.join(ROOT, "profiles", "r06")
KERNEL_SOURCES = ("zg_kernels.hip", "zg_flat1.h", "zg_flat4.h", "zg_huf.h", "zg_exact.h", "zg_dev.h", "zg_types.h")   # what the device code is built from


def pin_to_gpu_node(idx):
    """Run this process on the CPUs of the NUMA node the GPU hangs off (sysfs local_cpulist of its PCI function): pinned host buffers are then
    allocated next to the GPU and the host-to-host figures (e2e, stream) stop depending on where the scheduler happened to put the process
    (27 vs 40 GB/s between two runs of round 6). Returns (description, the affinity to restore for the CPU-baseline legs)."""
    try:
        import torch
        before = os.sched_getaffinity(0)
        pr = torch.cuda.get_device_properties(idx)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        txt = open("/sys/bus/pci/devices/%s/local_cpulist" % bdf).read().strip()
        cpus = set()
        for part in txt.split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= before
        if not cpus:
            return "unchanged (empty local_cpulist)", before
        os.sched_setaffinity(0, cpus)
        return "GPU %d at %s: CPUs %s" % (idx, bdf, txt), before
    except Exception as e:   # no sysfs entry, no permission ...: run where the scheduler puts us
        return "unchanged (%s)" % type(e).__name__, None


def pmap(fn, items):
    """fn over items on host threads (the generators and libzstd are C behind ctypes: the GIL is released while they run), results in
    order: eight GPUs' worth of frames are generated and compressed side by side instead of one after the other"""
    items = list(items)
    if len(items) <= 1:
        return [fn(x) for x in items]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(len(items), os.cpu_count() or 1, 64))) as ex:
        return list(ex.map(fn, items))


def build_workload(name, gpu, size, small=False):
    """returns (description, distinct plaintexts, repeat, sharded?, data tag). The job's frames are distinct x repeat for
    per-GPU workloads (gpu = index of the GPU they are built for), the whole job's frames for sharded ones."""
    import zgdata
    data_dir = os.environ.get("ZGPU_DATA")
    if name == "enwik9like":
        path = os.path.join(data_dir, "enwik9") if data_dir else None
        if path and os.path.exists(path):
            plain = open(path, "rb").read()
            return "enwik9 (%d B from $ZGPU_DATA) | libzstd %s -3, one frame per GPU" % (len(plain), zgdata.zstd_version()), [plain], 1, False, "file"
        plain = zgdata.text_like(size, seed=0xE9 + gpu)
        return ("enwik9-like single frame: text_like(%d B, seed 0xE9+gpu, V=14000) | libzstd %s -3, one frame per GPU"
                % (size, zgdata.zstd_version())), [plain], 1, False, "synthetic"
    if name == "realtext":
        import realtext
        plain, info = realtext.load()
        return ("real text found in this image (%d files, %d B, sha256 %s.., manifest %s%s) | libzstd %s -3, one frame per GPU"
                % (info["files"], info["bytes"], info["sha256"][:16], info["manifest"],
                   "".join("; skipped %s: %s" % (k, why) for k, why in info["skipped"][:4]), zgdata.zstd_version())), [plain], 1, False, "file"
    if name == "realtext1g":
        import realtext
        plain, info = realtext.load_tiled(1000000000)
        return ("real text found in this image stretched to %d B (%.2f passes over %d B in 4 MiB pieces, seeded permutations; sha256 %s.., corpus manifest %s) "
                "| libzstd %s -3, one frame per GPU" % (info["tiled_bytes"], info["passes"], info["bytes"], info["tiled_sha256"][:16], info["manifest"],
                                                       zgdata.zstd_version())), [plain], 1, False, "file"
    if name == "silesia12":
        sdir = os.path.join(data_dir, "silesia") if data_dir else None
        if sdir and os.path.isdir(sdir) and len(os.listdir(sdir)) >= 12:
            plains = [open(os.path.join(sdir, f), "rb").read() for f in sorted(os.listdir(sdir))]
            return "Silesia (%d files from $ZGPU_DATA) | libzstd -3, one frame per file" % len(plains), plains, 1, True, "file"
        plains = pmap(lambda a: zgdata.text_like(a[1], seed=0x51 + a[0]) if a[0] % 3 else zgdata.iso_like(a[1], seed=0x51 + a[0]), enumerate(SILESIA_SIZES))
        return "Silesia-sized stand-in: 12 frames (text_like / iso_like, sizes of the 12 Silesia files) | libzstd -3", plains, 1, True, "synthetic"
    if name == "blocks":
        rep = 1 if small else 8
        plains = pmap(lambda i: zgdata.text_like(64 << 20, seed=0xE9 + 16 * gpu + i), range(16))
        return "%d x 64 MiB text_like frames (%d GiB of 128 KiB blocks: 16 distinct x %d) per GPU | libzstd -3" % (16 * rep, rep, rep), plains, rep, False, "synthetic"
    if name == "blocks4b":
        rep = 1 if small else 32
        big = zgdata.text_like(256 << 20, seed=0xE9 + gpu)
        plains = [big[i:i + (128 << 10)] for i in range(0, len(big), 128 << 10)]
        return "%d single-block frames (128 KiB each: 2048 distinct x %d) per GPU | libzstd -3" % (2048 * rep, rep), plains, rep, False, "synthetic"
    if name == "iso":
        rep = 1 if small else 16
        plains = pmap(lambda i: zgdata.iso_like(64 << 20, seed=0x150 + 8 * gpu + i), range(8))
        return "%d x 64 MiB iso_like frames (ratio 1.18: 8 distinct x %d) per GPU | libzstd -3" % (8 * rep, rep), plains, rep, False, "synthetic"
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
    `cores` threads (one frame per thread), 3 runs each. zs: one compressed sample frame (bounded: this leg stays in seconds)."""
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

    def runs(fn, n=5):
        out = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            out.append(time.perf_counter() - t0)
        return out

    def med(ts):
        return sorted(ts)[len(ts) // 2]

    one = C.create_string_buffer(plain_len)
    t_o1, t_z1 = runs(lambda: oracle_one(one)), runs(lambda: zstd_one(one))
    t_on, t_zn = runs(threaded(oracle_one, cores)), runs(threaded(zstd_one, cores))
    gb = plain_len / 1e9
    # ... and on every core the box has ("N = nproc, state N"; decode_all.rs:6-11 is the reference's own bench loop)
    allc = os.cpu_count() or cores
    if allc > cores:
        t_oa, t_za = runs(threaded(oracle_one, allc), 3), runs(threaded(zstd_one, allc), 3)
    else:
        t_oa, t_za = t_on, t_zn
    rate = lambda k, ts: [round(k * gb / t, 2) for t in ts]
    # every figure is the MEDIAN of its runs (five; three on all cores), all runs listed: the multi-thread legs spread by 2-3 x between runs
    # and boxes (unpinned threads that each fault in a fresh 64 MiB buffer) — they are context, not a baseline to divide by
    return {"value": round(gb / med(t_o1), 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": "%d-byte frame of the same workload through the oracle's decode_all (FrameDecoder::decode_all semantics), median of 5 runs" % plain_len,
            "runs_GBps": rate(1, t_o1),
            "oracle_nt_GBps": round(cores * gb / med(t_on), 4), "nt_cores": cores, "oracle_nt_GBps_runs": rate(cores, t_on),
            "libzstd_1t_GBps": round(gb / med(t_z1), 4), "libzstd_1t_GBps_runs": rate(1, t_z1),
            "libzstd_nt_GBps": round(cores * gb / med(t_zn), 4), "libzstd_nt_GBps_runs": rate(cores, t_zn),
            "all_cores": allc, "oracle_all_GBps": round(allc * gb / med(t_oa), 4), "oracle_all_GBps_runs": rate(allc, t_oa),
            "libzstd_all_GBps": round(allc * gb / med(t_za), 4), "libzstd_all_GBps_runs": rate(allc, t_za),
            "libzstd_version": zgdata.zstd_version(), "host_cores_available": os.cpu_count(),
            "note": "medians; nt = one frame per thread on nt_cores threads, all = the same on every hardware thread of the box (threads not pinned: the "
                    "multi-thread figures are context only, their runs are listed); ruzstd itself is not buildable here (no rustc/cargo): "
                    "the oracle is its C port"}


def check_frames(pool, plains, repeat, first=0):
    """every staged frame's output against the plaintext it was made from (byte for byte)"""
    import numpy as np
    refs = [np.frombuffer(p, dtype=np.uint8) for p in plains]
    n = len(plains) * repeat
    for k in range(n):
        ref = refs[k % len(plains)]
        gpu, size, st = pool.frame(first + k)
        assert st == 0 and size == len(ref), ("frame", k, "status", st, "size", size)
        got = np.frombuffer(pool.read(first + k, size), dtype=np.uint8)
        assert np.array_equal(got, ref), "GPU output differs (frame %d)" % k


def kernels_sha256():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "zstd-rs_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def roofline(kern, A, workload, pass_ms=None, njobs=1, plaintext_bytes=None, plan=None):
    """roofline block of one workload on one GPU. kern: per-kernel ms of a pass (HIP events on the engines' streams, summed over
    the GPU's resident jobs); A: algorithmic bytes of the pass (every compressed byte read once + every plaintext byte written
    once); pass_ms: how long the GPU took for the pass — with one job that is the sum of its kernels, with several jobs
    in flight on two engines their kernels overlap and the pass is shorter than that sum."""
    dom = max(KERNELS, key=lambda k: kern[k])
    t_pipe = pass_ms if (pass_ms and njobs > 1) else kern["total"]
    pipe = A / (t_pipe / 1e3) / 1e9 if t_pipe > 0 else 0.0
    ach_dom = A / (kern[dom] / 1e3) / 1e9 if kern[dom] > 0 else 0.0
    # the kernel behind the "flat" time: the direct units of a submit with many of them have a kernel of their own (zg_k_flatten4, round 5)
    kname = "zg_k_" + dom
    if dom == "flat":
        kname = "zg_k_flatten4" if (plan and plan.get("pointer_units", 1) == 0 and plan.get("direct_units", 0) >= 1024) else "zg_k_flatten"
    traffic, traffic_src = None, None
    try:   # HBM bytes from the committed rocprofv3 PMC passes; refused when the kernels or the workload differ from what they were taken on
        pm = json.load(open(os.path.join(PROFILE_DIR, "%s_pmc.json" % workload)))
        if pm.get("kernels_sha256") != kernels_sha256():
            traffic_src = "profiles/r06/%s_pmc.json is stale for this build: not reported" % workload
        elif plaintext_bytes is not None and pm.get("plaintext_bytes") != plaintext_bytes:
            traffic_src = "profiles/r06/%s_pmc.json was taken on %s plaintext bytes, this run decodes %s: not reported" % (workload, pm.get("plaintext_bytes"), plaintext_bytes)
        else:
            traffic = pm["pipeline_hbm_bytes_per_pass"]
            traffic_src = "profiles/r06/%s_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, per-kernel calibration; all kernels of one pass)" % workload
    except Exception:
        pass
    # the same time against the bytes the counters saw: how far the HBM is really loaded (frac counts only C + D, SURVEY 8d: the gap between
    # the two is the pipeline's own traffic — scratch words, sequence records, gathers)
    frac_traffic = round(traffic / (t_pipe / 1e3) / 1e9 / HBM_PEAK_GBS, 6) if (traffic and t_pipe > 0) else None
    return {"bound": "hbm", "achieved": round(pipe, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(pipe / HBM_PEAK_GBS, 6),
            "frac_of_traffic": frac_traffic,
            "scope": "whole pipeline of one pass on one GPU: (C + D) / t_pass (SURVEY 8d); t_pass = sum of the kernels of the pass (one job) "
                     "or the time the GPU's two engines took for their jobs (several jobs in flight: kernels of different jobs overlap)",
            "jobs_in_flight": njobs, "t_pass_ms": round(t_pipe, 4),
            "kernel": kname, "kernel_ms": round(kern[dom], 4), "achieved_dominant": round(ach_dom, 3),
            "frac_dominant": round(ach_dom / HBM_PEAK_GBS, 6), "algorithmic_bytes": int(A), "traffic": traffic, "traffic_source": traffic_src}


def timed_passes(pool, steps, min_seconds, barrier=None):
    """the timed region: `steps` steps of `pps` passes each, pps chosen so that the region lasts >= min_seconds"""
    t0 = time.perf_counter()
    pool.run()
    t1 = time.perf_counter() - t0
    pps = max(1, int(math.ceil(min_seconds / max(steps * t1, 1e-9))))
    if barrier:
        barrier()
    t0 = time.perf_counter()
    busy = [0.0] * pool.n_gpus
    for _ in range(steps * pps):
        gms, _wall = pool.run()                        # blocks until every GPU of the pool has finished the pass
        for g, v in enumerate(gms):
            busy[g] += v
    if barrier:
        barrier()
    return time.perf_counter() - t0, pps, [b / (steps * pps) for b in busy]


def other_workload(name, device, min_seconds):
    """one of the other BASELINE.json configurations on one GPU, at one GPU's share of its size"""
    import zgdata
    import zgpu
    t_prep = time.perf_counter()
    desc, plains, rep, _, _ = build_workload(name, 0, 0)
    zs = pmap(lambda p: zgdata.zstd_compress(p, level=3), plains)
    pool = zgpu.Pool(devices=[device])
    pool.stage(zs * rep)
    prep = time.perf_counter() - t_prep
    for _ in range(2):
        pool.run()
    check_frames(pool, plains, rep)
    dt, pps, busy = timed_passes(pool, 1, min_seconds)
    kern, D, Cb, nb = pool.timings(0)
    pool_jobs = pool.last_njobs
    plan = pool.plan_stats(0)
    pool.close()
    return {"workload": desc, "plaintext_bytes": D, "compressed_bytes": Cb, "frames": len(zs) * rep, "blocks": nb, "passes": pps,
            "GBps": round(D * pps / dt / 1e9, 3), "ms_per_pass": round(dt / pps * 1e3, 3), "kernel_ms_per_pass": round(busy[0], 3),
            "kernel_ms": {k: round(v, 4) for k, v in kern.items()}, "lz77_plan": plan, "roofline": roofline(kern, Cb + D, name, busy[0], pool_jobs, D, plan), "host_prepare_s": round(prep, 2)}


def e2e_rate(device, zs_list, plain_total):
    """C ABI zgpu_pool_decode_all: pinned host buffer in, pinned host buffer out (host walk + H2D + kernels + D2H, jobs overlapped)"""
    import torch
    import zgpu
    blob = b"".join(zs_list)
    src = torch.frombuffer(bytearray(blob), dtype=torch.uint8).pin_memory()
    dst = torch.empty(plain_total, dtype=torch.uint8).pin_memory()
    pool = zgpu.Pool(devices=[device])
    w = C.c_size_t()

    def go():
        st = pool.L.zgpu_pool_decode_all(pool.h, C.c_char_p(src.data_ptr()), len(blob), C.c_void_p(dst.data_ptr()), plain_total, C.byref(w))
        assert st == 0 and w.value == plain_total, (st, w.value)
    go()
    t = _best_of(go)
    pool.close()
    return plain_total / t / 1e9, hashlib.sha256(dst.numpy().tobytes()).digest()


def stream_rates(device, z, plain):
    """The io::Read surface (zgpu_streaming_*: StreamingDecoder::read with read-ahead, zg_stream.h) on ONE frame, host to host: the
    compressed frame lies in pinned host memory, std::io::copy-style loops with caller buffers of 8 KiB (the reference CLI's), 1 MiB and
    64 MiB drain it (zgpu_streaming_copy into a sink: every byte is copied into the caller's buffer and dropped), with the content
    checksum computed (ruzstd's default `hash` feature) and without; and the whole frame read into one pinned buffer of its size
    (e2e_single_frame). Source walk + H2D + kernels + D2H + the reader's copy, best of 3 after one warm-up (the pinned ring of the
    first stream is kept by the library). Parity: byte counts, calculated == stored checksum, sha256 of the single-buffer read."""
    import torch
    import zgpu
    ctx = zgpu.Context(device)
    src = torch.frombuffer(bytearray(z), dtype=torch.uint8).pin_memory()
    dst = torch.empty(len(plain), dtype=torch.uint8).pin_memory()
    want_sha = hashlib.sha256(plain).digest()

    def one_copy(buf, checksum, callback=False):
        if callback:
            s = zgpu.CStreamingDecoder(ctx, _PinnedReader(src), checksum=checksum)
        else:
            s = zgpu.CStreamingDecoder(ctx, data=(src.data_ptr(), len(z)), checksum=checksum)
        t0 = time.perf_counter()
        n = s.copy_to_sink(buf)
        dt = time.perf_counter() - t0
        st = s.stats()
        assert n == len(plain) and s.is_finished() and st["dropped"] == 0, (n, st)
        if checksum:
            assert s.get_calculated_checksum() == s.get_checksum_from_data()
        s.close()
        last_stats.clear(); last_stats.update(st)
        return dt

    last_stats = {}

    def one_read(checksum):
        s = zgpu.CStreamingDecoder(ctx, data=(src.data_ptr(), len(z)), checksum=checksum)
        t0 = time.perf_counter()
        n = s.read_into(dst.data_ptr(), len(plain))
        dt = time.perf_counter() - t0
        assert n == len(plain)
        assert s.read_into(dst.data_ptr(), 16) == 0 and s.is_finished()
        s.close()
        return dt
    one_copy(1 << 20, True)                              # warm-up: the ring, the engine's buffers
    gb = len(plain) / 1e9
    out = {"stream_GBps": {}, "stream_nohash_GBps": {}}
    for label, buf in (("8KiB", 8192), ("1MiB", 1 << 20), ("64MiB", 64 << 20)):
        out["stream_GBps"][label] = round(gb / min(one_copy(buf, True) for _ in range(3)), 3)
        out["stream_nohash_GBps"][label] = round(gb / min(one_copy(buf, False) for _ in range(3)), 3)
    # where the 64 MiB loop's time went (zgpu_streaming_stats of its last run; ms): the worker thread's host prepare + kernels per run, the reader
    out["stream_breakdown_ms"] = {k[3:]: round(v / 1e3, 2) for k, v in last_stats.items() if k.startswith("us_")}
    out["stream_breakdown_ms"].update({"runs": last_stats.get("runs"), "kernels_total": round(last_stats.get("k_total", 0) / 1e3, 2),
                                       "k_seq": round(last_stats.get("k_seq", 0) / 1e3, 2), "k_flat": round(last_stats.get("k_flat", 0) / 1e3, 2),
                                       "k_sweep": round(last_stats.get("k_sweep", 0) / 1e3, 2)})
    out["stream_callback_GBps"] = {"1MiB": round(gb / min(one_copy(1 << 20, True, True) for _ in range(2)), 3)}
    t = min(one_read(False) for _ in range(3))
    assert hashlib.sha256(dst.numpy().tobytes()).digest() == want_sha, "stream output differs"
    out["e2e_single_frame_GBps"] = round(gb / t, 3)
    out["e2e_single_frame_hash_GBps"] = round(gb / min(one_read(True) for _ in range(2)), 3)
    out["stream_note"] = ("zgpu_streaming_* (StreamingDecoder::read with read-ahead) on the %d-byte frame, compressed bytes in pinned host memory (slice source; "
                          "stream_callback: an io::Read callback that copies from it), zgpu_streaming_copy with caller buffers of 8 KiB / 1 MiB / 64 MiB into a sink; "
                          "stream_GBps with the XXH64 of ruzstd's default `hash` feature (hasher thread), stream_nohash without; e2e_single_frame: one read() of "
                          "the whole frame into a pinned buffer. Walk + H2D + kernels + D2H + the reader's copy; best of 3; one GPU" % len(plain))
    ctx.close()
    return out


class _PinnedReader:
    """an io::Read-like source over pinned memory (a torch uint8 tensor): read(n) -> bytes"""

    def __init__(self, t):
        self.t, self.pos, self.n = t, 0, t.numel()

    def read(self, n):
        k = min(n, self.n - self.pos)
        b = C.string_at(self.t.data_ptr() + self.pos, k)
        self.pos += k
        return b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="enwik9like", choices=["enwik9like", "realtext", "realtext1g", "silesia12", "blocks", "blocks4b", "iso"])
    ap.add_argument("--size", type=int, default=int(os.environ.get("ZGPU_BENCH_SIZE", 1000000000)), help="plaintext bytes per GPU (enwik9like)")
    ap.add_argument("--min-seconds", type=float, default=2.0, help="lower bound of the timed region (passes per step are raised to reach it)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-other", action="store_true", help="skip the runs of the other BASELINE.json configurations")
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
    affinity_note, affinity_before = pin_to_gpu_node(local_rank) if args.gpus == 1 or world > 1 else ("unchanged (one process drives several GPUs)", None)

    import zgdata
    import zgpu
    import zgpu_dist
    # one process per GPU (torch.distributed.run), or one process driving --gpus engines through the library's queue
    in_process = world == 1 and args.gpus > 1
    pool = zgpu.Pool(n_gpus=args.gpus) if in_process else zgpu.Pool(devices=[local_rank])
    ngpu_here = pool.n_gpus                                   # (zgpu_pool_num_gpus: what the node really has)
    n_gpus = world if world > 1 else ngpu_here
    t0 = time.perf_counter()
    # (host preparation of all GPUs side by side: eight 1e9-byte generations + compressions are ~90 s one after the other)
    built = pmap(lambda g: build_workload(args.workload, (rank if world > 1 else g), args.size), range(ngpu_here if args.workload != "silesia12" else 1))
    desc, _, _, sharded, data_tag = built[0]
    per_gpu = [(b[1], b[2]) for b in built]                  # (plains, repeat) per GPU of this process
    flat = [(g, i) for g, (plains, _) in enumerate(per_gpu) for i in range(len(plains))]
    zflat = pmap(lambda gi: zgdata.zstd_compress(per_gpu[gi[0]][0][gi[1]], level=3), flat)
    zs_gpu = [[] for _ in per_gpu]
    for (g, _), z in zip(flat, zflat):
        zs_gpu[g].append(z)
    if sharded:
        job_lens = [len(z) for z in zs_gpu[0]]
        mine = zgpu_dist.shard_frames(job_lens, world)[rank] if world > 1 else list(range(len(job_lens)))
        staged_z = [zs_gpu[0][i] for i in mine]
        staged_p = [per_gpu[0][0][i] for i in mine]
        order, _, loads = zgpu.plan(job_lens, n_gpus)
    else:
        staged_z, staged_p = [], []
        for (plains, rep), zs in zip(per_gpu, zs_gpu):
            staged_z += zs * rep
            staged_p += plains * rep
        job_lens, loads = [len(z) for z in staged_z], None
    pool.stage(staged_z)                                     # host block walk + H2D: the submission, outside the timed region
    prep_s = time.perf_counter() - t0

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        pool.run()
    # parity gate on a warm run: the full plaintext of every frame must match the generator's bytes
    import numpy as np
    used_gpus = set()
    for k, p in enumerate(staged_p):
        gpu, size, st = pool.frame(k)
        used_gpus.add(gpu)
        assert st == 0 and size == len(p), (k, st, size)
        assert np.array_equal(np.frombuffer(pool.read(k, size), dtype=np.uint8), np.frombuffer(p, dtype=np.uint8)), "GPU output differs (frame %d)" % k

    # every GPU this process drives must have taken part (one frame per GPU, or at least as many sharded frames as GPUs)
    if world == 1 and len(staged_p) >= ngpu_here:
        assert len(used_gpus) == ngpu_here, ("GPUs used", sorted(used_gpus), "GPUs driven", ngpu_here)

    dt, pps, busy = timed_passes(pool, args.steps, args.min_seconds, barrier)
    if world > 1:
        tmax = torch.tensor([dt], device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    D_mine, C_mine = sum(len(p) for p in staged_p), sum(len(z) for z in staged_z)
    if world > 1:
        tot = torch.tensor([float(D_mine), float(C_mine), busy[0]], device="cuda", dtype=torch.float64)
        allt = [torch.zeros_like(tot) for _ in range(world)]
        dist.all_gather(allt, tot)
        D_job, C_job = sum(float(t[0]) for t in allt), sum(float(t[1]) for t in allt)
        per_gpu_busy = [float(t[2]) for t in allt]
    else:
        D_job, C_job, per_gpu_busy = float(D_mine), float(C_mine), busy
    value = D_job * args.steps * pps / dt / 1e9

    if rank == 0:
        kern, D0, C0, nblocks = pool.timings(0)               # GPU 0's last pass, per kernel (HIP events on the engine's own streams)
        out = {
            "metric": "decompressed_GB_per_s", "value": round(value, 4), "unit": "GB/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "u8", "data": data_tag,
            "config": {"workload": desc, "name": args.workload, "plaintext_bytes_job": int(D_job), "compressed_bytes_job": int(C_job),
                       "passes_per_step": pps, "plaintext_bytes_per_step": int(D_job) * pps, "ms_per_pass": round(dt / (args.steps * pps) * 1e3, 3),
                       "frames_job": len(staged_z) if world == 1 else None, "frames_gpu0_blocks": nblocks,
                       "timed_region": "kernels only, inputs + block tables resident in HBM, output left in HBM; steps x passes_per_step "
                                       "passes (one zgpu_pool_run each), >= %.1f s" % args.min_seconds,
                       "queue": ("one process, zgpu_pool_create(%d): LPT order, one worker thread + engine per GPU; GPUs used: %s" % (args.gpus, sorted(used_gpus)))
                                if world == 1 else "one process per GPU (torch.distributed.run); frames -> ranks by zgpu_dist.shard_frames (the queue's LPT rule)",
                       "gpus_requested": args.gpus, "host_prepare_s": round(prep_s, 3), "host_affinity": affinity_note},
            "roofline": roofline(kern, C0 + D0, args.workload, per_gpu_busy[0], pool.last_njobs, D0, pool.plan_stats(0)),
            "read_GBps": round(C0 / (kern["total"] / 1e3) / 1e9, 3) if kern["total"] > 0 else 0.0,
            "kernel_ms": {k: round(v, 4) for k, v in kern.items()},
            "lz77_plan": pool.plan_stats(0),
            "per_gpu_busy_ms": [round(x, 3) for x in per_gpu_busy],
        }
        if world == 1 and ngpu_here > 1:   # one process drives all GPUs: the last pass of each, per kernel
            out["per_gpu_kernel_ms"] = [{k: round(v, 4) for k, v in pool.timings(g)[0].items()} for g in range(ngpu_here)]
        if sharded:
            out["lpt"] = {"loads_compressed_bytes": loads, "bound_speedup": round(sum(job_lens) / max(loads), 3) if max(loads) else float(n_gpus)}
        pool.close(); pool = None
        if not args.no_e2e:
            # the path a caller of the ABI sees: 1 GiB of 64 MiB frames (or the workload's own frames when they are many), host to host
            if len(staged_z) >= 8:
                ez, ep = staged_z[:16], staged_p[:16]
            else:
                ep = [zgdata.text_like(64 << 20, seed=0xE9 + i) for i in range(16)]
                ez = [zgdata.zstd_compress(p, level=3) for p in ep]
            rate, digest = e2e_rate(local_rank, ez, sum(len(p) for p in ep))
            assert digest == hashlib.sha256(b"".join(ep)).digest()
            out["e2e_GBps"] = round(rate, 3)
            out["e2e_note"] = ("C ABI zgpu_pool_decode_all on %d frames (%d B of plaintext): pinned host buffer in, pinned host buffer out; host walk + H2D "
                               "+ kernels + D2H with the jobs of two engines overlapped, best of 3; per GPU" % (len(ez), sum(len(p) for p in ep)))
        if not args.no_e2e and n_gpus == 1 and args.workload == "enwik9like" and len(staged_z) == 1:
            out.update(stream_rates(local_rank, staged_z[0], staged_p[0]))
        if not args.no_cpu:
            if affinity_before:
                os.sched_setaffinity(0, affinity_before)         # the CPU legs use every core of the box
            # bounded CPU sample of the same workload: one frame of at most 64 MiB of plaintext
            n = min(len(staged_p[0]), 64 << 20)
            zc = staged_z[0] if n == len(staged_p[0]) else zgdata.zstd_compress(staged_p[0][:n], level=3)
            out["cpu_baseline"] = cpu_baseline(zc, n, max(1, min(os.cpu_count() or 1, 64)))
        if not args.no_other and n_gpus == 1 and args.workload == "enwik9like":
            # the other BASELINE.json configurations on this GPU, at one GPU's share of their size (parity gate on every frame first):
            # context for the headline, not part of `value`
            del staged_p, staged_z, per_gpu, zs_gpu
            out["other_workloads"] = {w: other_workload(w, local_rank, args.min_seconds) for w in ("realtext1g", "silesia12", "blocks", "blocks4b", "iso", "realtext")}
        print(json.dumps(out), flush=True)
    if pool is not None:
        pool.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
