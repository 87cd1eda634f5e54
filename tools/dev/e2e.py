#!/usr/bin/env python3
"""PCIe-inclusive rate of zgpu_pool_decode_all on 1 GiB of 64 MiB text frames (pinned host buffers in and out), for a list of
environment variants: e2e.py -- "" ZGPU_DA_SPLIT=16 "ZGPU_DA_SPLIT=16 ZGPU_DA_FLOOR_MB=16" """
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd"))
import bench, zgdata
variants = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else [""]
ep = [zgdata.text_like(64 << 20, seed=0xE9 + i) for i in range(16)]
ez = [zgdata.zstd_compress(p, level=3) for p in ep]
want = hashlib.sha256(b"".join(ep)).digest()
for v in variants:
    kv = dict(x.split("=", 1) for x in v.split()) if v else {}
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update(kv)
    rate, digest = bench.e2e_rate(0, ez, sum(len(p) for p in ep))
    for k, o in old.items():
        if o is None: del os.environ[k]
        else: os.environ[k] = o
    print("%-50s %s %7.2f GB/s" % (v or "default", "OK " if digest == want else "BAD", rate), flush=True)
