#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
( time bash tools/dev/profile.sh blocks4b ) > gpurun_out/profile_4b.log 2>&1
tail -5 gpurun_out/profile_4b.log | cut -c1-300
python - <<'PY'
import csv, collections, glob
for k in (1,2,3):
    f = "gpurun_out/prof/blocks4b_sq%d.csv" % k
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(r["Kernel_Name"], r["Counter_Name"])] += 1
    for kn in acc:
        if "flatten" in kn or "seqpost" in kn or kn.startswith("zg_k_seq") or "huf" in kn:
            print(k, kn[:40], {c: int(v / cnt[(kn, c)]) for c, v in acc[kn].items()})
PY
