"""Sharding independent frames over the GPUs of one node.

Frames are independent units: no data-path collective. With one process per GPU (torch.distributed), every rank takes the
frames of its shard — the same longest-processing-time-first rule the library's own work queue uses (zgpu_pool_plan) —
decodes them on its own GPU through the queue (zgpu.Pool over that one GPU) and keeps the plaintext there. The only
communication is what a caller wants to know afterwards (sizes / digests): one all_gather of small tensors over RCCL
(backend "nccl", tensors on the rank's GPU) or gloo (CPU tests).
"""


def shard_frames(sizes, world_size):
    """LPT assignment. sizes[i] = compressed size of frame i. Returns a list of world_size lists of frame indices.
    Identical to zgpu_pool_plan (tests/test_dist_cpu.py checks it)."""
    order = sorted(range(len(sizes)), key=lambda i: (-sizes[i], i))
    loads = [0] * world_size
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += sizes[i]
    for s in shards:
        s.sort()
    return shards


def decode_sharded(frames, decode_fn, rank, world_size):
    """frames: list of bytes (one .zst frame each). decode_fn(frame_bytes) -> plaintext bytes (this rank's engine).
    Returns {frame_index: plaintext} for the frames of this rank's shard."""
    mine = shard_frames([len(f) for f in frames], world_size)[rank]
    return {i: decode_fn(frames[i]) for i in mine}


def gpu_decode_fn(local_rank):
    """decode_fn for decode_sharded on a real GPU: this rank's device behind the library's work queue. Returns (fn, pool)."""
    import zgpu
    pool = zgpu.Pool(devices=[local_rank])

    def fn(z):
        return pool.decode_all(z, _bound(z))
    return fn, pool


def _bound(z):
    """upper bound of the plaintext of a run of frames: 128 KiB per block is the format's limit for conforming frames; the
    pool sizes the device side exactly, this only sizes the host buffer"""
    n, p, total = len(z), 0, 0
    while p + 4 <= n:
        magic = int.from_bytes(z[p:p + 4], "little")
        if 0x184D2A50 <= magic <= 0x184D2A5F:
            p += 8 + int.from_bytes(z[p + 4:p + 8], "little")
            continue
        desc = z[p + 4]
        single = (desc >> 5) & 1
        p += 5 + (0 if single else 1) + (0, 1, 2, 4)[desc & 3] + ((1 if single else 0), 2, 4, 8)[desc >> 6]
        while True:
            h = z[p] | (z[p + 1] << 8) | (z[p + 2] << 16)
            btype, size = (h >> 1) & 3, h >> 3
            p += 3 + (1 if btype == 1 else size)
            total += size if btype != 2 else 128 << 10
            if h & 1:
                p += 4 if (desc >> 2) & 1 else 0
                break
    return total


def gather_digests(local, nframes, dist):
    """all_gather of (index, length, 56-bit digest) so that every rank can check the whole job.
    local: {frame_index: plaintext}. dist: torch.distributed (initialised). Returns {index: (length, digest)}."""
    import hashlib
    import torch
    # RCCL ("nccl") collectives take device tensors only; gloo takes CPU tensors
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    rows = torch.zeros((nframes, 3), dtype=torch.int64)
    for i, b in local.items():
        rows[i, 0] = 1
        rows[i, 1] = len(b)
        rows[i, 2] = int.from_bytes(hashlib.sha256(b).digest()[:7], "little")
    rows = rows.to(dev)
    out = [torch.zeros_like(rows) for _ in range(dist.get_world_size())]
    dist.all_gather(out, rows)
    res = {}
    for t in out:
        t = t.cpu()
        for i in range(nframes):
            if int(t[i, 0]):
                assert i not in res, "frame decoded by two ranks"
                res[i] = (int(t[i, 1]), int(t[i, 2]))
    return res
