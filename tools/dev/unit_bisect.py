"""bisecting aid: decode a workload in a child process per set of engine switches, report OK / wrong / crash"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, os, hashlib
sys.path.insert(0, os.path.join(%r, "zstd-rs_amd")); sys.path.insert(0, os.path.join(%r, "tools")); sys.path.insert(0, os.path.join(%r, "tests"))
import zgpu, zgdata
from golden_io import read_pack, read_manifest
which = sys.argv[1]
if which == "corpus":
    pack, man = read_pack("decodecorpus.pack"), read_manifest("decodecorpus.json")
    names = sorted(man)
    z = b"".join(pack[n] for n in names); size = sum(man[n]["size"] for n in names)
    want = None
elif which == "many":
    plains = [zgdata.text_like(8 << 20, seed=0xE9 + i) for i in range(16)]
    z = b"".join(zgdata.zstd_compress(p) for p in plains); size = sum(map(len, plains)); want = hashlib.sha256(b"".join(plains)).digest()
elif which == "one":
    p = zgdata.text_like(64 << 20, seed=5); z = zgdata.zstd_compress(p); size = len(p); want = hashlib.sha256(p).digest()
c = zgpu.Context(0, dev=True)
out = c.decode_all(z, size)
ok = len(out) == size and (want is None or hashlib.sha256(out).digest() == want)
print("RESULT", "OK" if ok else "WRONG", flush=True)
''' % (ROOT, ROOT, ROOT)
for which in sys.argv[1:] or ["corpus", "many", "one"]:
    for sw in ["", "ZGPU_DIRECT=0", "ZGPU_DEBUG_NO_SWEEP=1", "ZGPU_DIRECT=0,ZGPU_DEBUG_NO_SWEEP=1", "ZGPU_UNIT_BLOCKS=15",
               "ZGPU_UNIT_BLOCKS=15,ZGPU_DEBUG_NO_SWEEP=1", "ZGPU_FLAT_T=512"]:
        env = dict(os.environ)
        for kv in sw.split(","):
            if kv:
                k, v = kv.split("="); env[k] = v
        r = subprocess.run([sys.executable, "-c", CHILD, which], env=env, capture_output=True, text=True)
        res = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        print("%-8s %-70s %s" % (which, sw or "default", res[0] if res else "CRASH rc=%d %s" % (r.returncode, (r.stderr.strip().splitlines() or [""])[0][:100])), flush=True)
