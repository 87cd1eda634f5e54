"""A real-text corpus from files that are part of this image (no network: enwik9 cannot be fetched): source code and documentation
under the Python site-packages and the system / ROCm include directories, concatenated in sorted path order.

Deterministic across boxes (round 5): the boxes of this pool do not hold the same site-packages (round 4: 393 MB on the builder's GPU boxes,
475 MB on the driver's — twelve packages of the offline wheelhouse exist on one and not on the other). tools/realtext_manifest.json therefore
lists the TOP-LEVEL DIRECTORIES that go in, each with its file count, byte count and a sha256 prefix, as found both in the build container
and on a GPU box (tools/dev/call2.sh census, round 5): load() walks exactly those, skips one whose files differ and says so
(info["skipped"]: name and reason), and reports whether the concatenation is the manifest's ("match") or not ("differs").

usage: python tools/realtext.py [--write-manifest [census.json ...]]   (census: only directories that every given census also holds)"""
import hashlib
import json
import os
import sys

ROOTS = ["/usr/local/lib/python3.10/dist-packages", "/opt/rocm/include", "/usr/include", "/usr/lib/python3.10"]
EXTS = (".py", ".pyi", ".h", ".hpp", ".cuh", ".c", ".cpp", ".inc", ".md", ".rst", ".txt")
MANIFEST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "realtext_manifest.json")
WANT = 512 << 20


def census():
    """{root/top-level-entry: [(path, size) ...]} of every candidate file, paths sorted"""
    out = {}
    for root in ROOTS:
        for dp, dn, fn in os.walk(root):
            dn.sort()
            for f in sorted(fn):
                if not f.endswith(EXTS):
                    continue
                p = os.path.join(dp, f)
                if os.path.islink(p):
                    continue
                try:
                    s = os.path.getsize(p)
                except OSError:
                    continue
                if 1024 < s < (8 << 20):
                    rel = os.path.relpath(p, root).split(os.sep)
                    out.setdefault(root + "/" + (rel[0] if len(rel) > 1 else "."), []).append((p, s))
    return out


def dir_digest(files):
    h = hashlib.sha256()
    n = tot = 0
    for p, _ in files:
        try:
            b = open(p, "rb").read()
        except OSError:
            continue
        h.update(b); n += 1; tot += len(b)
    return [n, tot, h.hexdigest()[:16]]


def load(want=WANT):
    """returns (bytes, info). info: files, bytes, sha256, manifest ("match" / "differs" / "absent"), skipped [(directory, reason) ...]"""
    try:
        man = json.load(open(MANIFEST))
    except Exception:
        man = None
    cen = census()
    keys = sorted(cen, key=lambda k: (ROOTS.index(k[:k.rindex("/")]) if k[:k.rindex("/")] in ROOTS else 99, k))
    parts, skipped, nfiles = [], [], 0
    for k in (keys if man is None else [k for k in man["dirs"]]):
        if k not in cen:
            skipped.append((k, "missing on this box"))
            continue
        fl = cen[k]                                   # (walk order: a directory's files, then its subdirectories, names sorted)
        bs = []
        for p, _ in fl:
            try:
                bs.append(open(p, "rb").read())
            except OSError:
                pass
        if man is not None:
            h = hashlib.sha256()
            for b in bs:
                h.update(b)
            if [len(bs), sum(len(b) for b in bs), h.hexdigest()[:16]] != man["dirs"][k]:
                skipped.append((k, "its files differ from the manifest's"))
                continue
        parts.extend(bs)
        nfiles += len(bs)
    data = b"".join(parts)[:want]
    info = {"files": nfiles, "bytes": len(data), "sha256": hashlib.sha256(data).hexdigest(), "roots": ROOTS, "skipped": skipped}
    info["manifest"] = "absent" if man is None else ("match" if (man["sha256"] == info["sha256"] and man["bytes"] == info["bytes"]) else "differs")
    return data, info


def load_tiled(target=1000000000, chunk=4 << 20, seed=0xE9):
    """the corpus stretched to `target` bytes: the first pass in its own order, every further pass with its 4 MiB pieces in a seeded
    permutation. A piece meets its earlier copy hundreds of MB back — far beyond the 2 MiB window of `zstd -3` — so the tiling adds no
    match the compressor can use: per byte this is the corpus's own statistics, at enwik9's size. Returns (bytes, info)."""
    import random
    data, info = load()
    pieces = [data[i:i + chunk] for i in range(0, len(data), chunk)]
    out, n, rng = [data], len(data), random.Random(seed)
    while n < target:
        order = list(range(len(pieces)))
        rng.shuffle(order)
        for i in order:
            out.append(pieces[i]); n += len(pieces[i])
            if n >= target:
                break
    big = b"".join(out)[:target]
    info = dict(info)
    info.update({"tiled_bytes": len(big), "tiled_sha256": hashlib.sha256(big).hexdigest(), "passes": round(len(big) / max(len(data), 1), 2)})
    return big, info


if __name__ == "__main__":
    if "--write-manifest" in sys.argv:
        cen = census()
        others = [json.load(open(a)) for a in sys.argv[sys.argv.index("--write-manifest") + 1:]]
        dirs = {}
        keys = sorted(cen, key=lambda k: (ROOTS.index(k[:k.rindex("/")]), k))
        for k in keys:
            d = dir_digest(cen[k])
            if all(o.get(k) == d for o in others):
                dirs[k] = d
        json.dump({"dirs": dirs, "bytes": 0, "sha256": "", "files": 0, "roots": ROOTS}, open(MANIFEST, "w"), indent=0)
        data, info = load()
        json.dump({"dirs": dirs, "bytes": info["bytes"], "sha256": info["sha256"], "files": info["files"], "roots": ROOTS}, open(MANIFEST, "w"), indent=0)
    data, info = load()
    print(json.dumps({k: v for k, v in info.items() if k != "skipped"}), "skipped:", info["skipped"][:5])
