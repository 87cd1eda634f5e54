"""decode every golden corpus frame one by one, printing the name first (debugging aid)"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zstd-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import zgpu
from golden_io import read_manifest, read_pack
pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
ctx = zgpu.Context(0, dev=True)
for name in sorted(man):
    b = ctx.prepare(pack[name])
    print("FRAME", name, "blocks", b.nblocks, flush=True)
    b.run(); b.sync()
    out = b.read(0, b.total_out)
    print("RESULT", name, b.bad_status, hashlib.sha256(out).hexdigest() == man[name]["sha256"], flush=True)
    b.close()
