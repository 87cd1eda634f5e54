"""The N>1 path on CPU: world_size-2 gloo. Frames shard with no data-path collective; the ranks only
exchange digests. The decode itself is stood in by the CPU harness of the engine's lane routines (tests/emu)."""
import hashlib
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, names, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd"))
    import emu
    import zgpu_dist
    from golden_io import read_pack
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pack = read_pack("decodecorpus.pack")
    frames = [pack[n] for n in names]

    def decode(z):
        out, st = emu.EmuBatch(z).frame_bytes(0)
        assert st == 0
        return out

    local = zgpu_dist.decode_sharded(frames, decode, rank, world)
    dist.barrier()
    res = zgpu_dist.gather_digests(local, len(frames), dist)
    t = torch.tensor([float(len(local))])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)           # same reduction bench.py uses for the step time
    if rank == 0:
        q.put((res, sorted(local), float(t.item())))
    dist.destroy_process_group()


def test_lpt_sharding_properties():
    sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd"))
    import zgpu_dist
    sizes = [51220480, 41458703, 33553445, 21606400, 10192446, 10085684, 9970564, 8474240, 7251944, 6627202, 6152192, 5345280]  # Silesia
    for w in (1, 2, 4, 8):
        sh = zgpu_dist.shard_frames(sizes, w)
        assert sorted(i for s in sh for i in s) == list(range(12))
        loads = [sum(sizes[i] for i in s) for s in sh]
        assert max(loads) <= max(sum(sizes) / w * 1.34, max(sizes))      # LPT bound 4/3 - 1/(3w), or one huge frame
    assert zgpu_dist.shard_frames([], 2) == [[], []]


def test_python_sharder_equals_library_queue_plan():
    """the ranks' static shards (zgpu_dist.shard_frames) and the library's work-queue plan (zgpu_pool_plan, host only) are the same
    LPT rule: a job that runs as N processes x 1 GPU and one that runs as 1 process x N GPUs place frames identically"""
    import random
    sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd"))
    import zgpu
    import zgpu_dist
    rnd = random.Random(5)
    for trial in range(40):
        n, w = rnd.randrange(0, 40), rnd.randrange(1, 9)
        sizes = [rnd.choice([rnd.randrange(1, 1 << 26), 12345]) for _ in range(n)]
        order, worker, load = zgpu.plan(sizes, w)
        sh = zgpu_dist.shard_frames(sizes, w)
        assert [sorted(i for i in range(n) if worker[i] == r) for r in range(w)] == sh
        assert load == [sum(sizes[i] for i in s) for s in sh]
        assert sorted(order) == list(range(n)) and all(sizes[order[k]] >= sizes[order[k + 1]] for k in range(n - 1))


def test_bound_of_a_run_of_frames():
    from golden_io import read_manifest, read_pack
    sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd"))
    import zgpu_dist
    pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
    for n in sorted(man)[:20]:
        assert zgpu_dist._bound(pack[n]) >= man[n]["size"]


def test_two_ranks_gloo():
    from golden_io import read_manifest
    man = read_manifest("decodecorpus.json")
    names = sorted(man)[:12]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, names, q)) for r in range(2)]
    for p in procs:
        p.start()
    res, mine0, tmax = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(res) == list(range(len(names)))        # every frame decoded exactly once across the two ranks
    for i, n in enumerate(names):
        assert res[i][0] == man[n]["size"]
        assert res[i][1] == int.from_bytes(bytes.fromhex(man[n]["sha256"])[:7], "little")
    assert 0 < len(mine0) < len(names) and tmax >= len(mine0)
