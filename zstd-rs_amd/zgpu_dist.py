"""Sharding independent frames over the GPUs of one node.

Frames are independent units: no data-path collective. Every rank (one process per GPU, torch.distributed) takes the
frames of its shard from a list ordered by descending compressed size (LPT: longest processing time first), decodes
them on its own GPU and keeps the plaintext there. The only communication is what a caller wants to know afterwards
(sizes / checksums): one all_gather of small tensors over RCCL (backend "nccl") or gloo (CPU tests).
"""


def shard_frames(sizes, world_size):
    """LPT assignment. sizes[i] = compressed size of frame i. Returns a list of world_size lists of frame indices."""
    order = sorted(range(len(sizes)), key=lambda i: (-sizes[i], i))
    loads = [0] * world_size
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += sizes[i]
    return shards


def decode_sharded(frames, decode_fn, rank, world_size):
    """frames: list of bytes (one .zst frame each). decode_fn(frame_bytes) -> plaintext bytes (this rank's engine).
    Returns {frame_index: plaintext} for the frames of this rank's shard."""
    mine = shard_frames([len(f) for f in frames], world_size)[rank]
    return {i: decode_fn(frames[i]) for i in mine}


def gather_digests(local, nframes, dist):
    """all_gather of (index, length, xxh-like 64-bit digest) so that every rank can check the whole job.
    local: {frame_index: plaintext}. dist: torch.distributed (initialised). Returns {index: (length, digest)}."""
    import hashlib
    import torch
    rows = torch.zeros((nframes, 3), dtype=torch.int64)
    for i, b in local.items():
        rows[i, 0] = 1
        rows[i, 1] = len(b)
        rows[i, 2] = int.from_bytes(hashlib.sha256(b).digest()[:7], "little")
    out = [torch.zeros_like(rows) for _ in range(dist.get_world_size())]
    dist.all_gather(out, rows)
    res = {}
    for t in out:
        for i in range(nframes):
            if int(t[i, 0]):
                assert i not in res, "frame decoded by two ranks"
                res[i] = (int(t[i, 1]), int(t[i, 2]))
    return res
