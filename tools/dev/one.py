"""decode one golden frame (debugging aid): one.py [name]"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import zgpu
from golden_io import read_manifest, read_pack
pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
name = sys.argv[1] if len(sys.argv) > 1 else sorted(man)[0]
ctx = zgpu.Context(0, dev=True)
b = ctx.prepare(pack[name])
print("frames", b.nframes, "blocks", b.nblocks, flush=True)
b.run(); b.sync()
print("status", b.bad_status, "total", b.total_out, flush=True)
out = b.read(0, b.total_out)
print("match", hashlib.sha256(out).hexdigest() == man[name]["sha256"], flush=True)
