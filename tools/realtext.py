"""A real-text corpus from files that are part of this image (no network: enwik9 cannot be fetched): source code and documentation
under the Python site-packages and the system / ROCm include directories, concatenated in sorted path order until `want` bytes are
reached. The GPU box runs the same image, so the same bytes come out there; tools/realtext_manifest.json holds the file count, the byte
count and the sha256 of the concatenation as built here, and load() says whether what it built matches it.

usage: python tools/realtext.py [--write-manifest]"""
import hashlib
import json
import os
import sys

ROOTS = ["/usr/local/lib/python3.10/dist-packages", "/opt/rocm/include", "/usr/include", "/usr/lib/python3.10"]
EXTS = (".py", ".pyi", ".h", ".hpp", ".cuh", ".c", ".cpp", ".inc", ".md", ".rst", ".txt")
MANIFEST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "realtext_manifest.json")
WANT = 512 << 20


def files(want=WANT):
    out, tot = [], 0
    for root in ROOTS:
        cand = []
        for dp, dn, fn in os.walk(root):
            dn.sort()
            for f in sorted(fn):
                if f.endswith(EXTS):
                    p = os.path.join(dp, f)
                    if os.path.islink(p):
                        continue
                    try:
                        s = os.path.getsize(p)
                    except OSError:
                        continue
                    if 1024 < s < (8 << 20):
                        cand.append((p, s))
        for p, s in sorted(cand):
            out.append(p)
            tot += s
            if tot >= want:
                return out
    return out


def load(want=WANT):
    """returns (bytes, info). info: files, bytes, sha256, manifest ("match" / "differs" / "absent")"""
    parts = []
    tot = 0
    flist = files(want)
    for p in flist:
        try:
            b = open(p, "rb").read()
        except OSError:
            continue
        parts.append(b)
        tot += len(b)
    data = b"".join(parts)[:want]
    info = {"files": len(flist), "bytes": len(data), "sha256": hashlib.sha256(data).hexdigest(), "roots": ROOTS}
    try:
        m = json.load(open(MANIFEST))
        info["manifest"] = "match" if (m["sha256"] == info["sha256"] and m["bytes"] == info["bytes"]) else "differs"
    except Exception:
        info["manifest"] = "absent"
    return data, info


if __name__ == "__main__":
    data, info = load()
    print(json.dumps(info))
    if "--write-manifest" in sys.argv:
        json.dump({k: info[k] for k in ("files", "bytes", "sha256", "roots")}, open(MANIFEST, "w"), indent=1)
