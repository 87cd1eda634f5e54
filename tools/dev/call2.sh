#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c9
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
python - > $OUT/realtext_dirs.json <<'PY'
import os, sys, json, hashlib
sys.path.insert(0, "tools")
import realtext
res = {}
for root in realtext.ROOTS:
    for dp, dn, fn in os.walk(root):
        dn.sort()
        for f in sorted(fn):
            if f.endswith(realtext.EXTS):
                p = os.path.join(dp, f)
                if os.path.islink(p): continue
                try: s = os.path.getsize(p)
                except OSError: continue
                if 1024 < s < (8 << 20):
                    rel = os.path.relpath(p, root).split(os.sep)
                    key = root + "/" + (rel[0] if len(rel) > 1 else ".")
                    h = res.setdefault(key, [0, 0, hashlib.sha256()])
                    h[0] += 1; h[1] += s; h[2].update(open(p, "rb").read())
print(json.dumps({k: [v[0], v[1], v[2].hexdigest()[:16]] for k, v in res.items()}))
PY
( timeout 900 python -m pytest tests -x -q -m gpu -k "trailing_frame or lies_by_gigabytes or inorder_fallback or ramped or sized_in_advance" ) > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
