#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c14
mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -x -q -m gpu ) > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
( time timeout 1500 python bench.py ) > $OUT/bench.log 2> $OUT/bench.err
tail -c 600 $OUT/bench.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/c14/bench.log") if x.startswith('{"metric"')]
d=json.loads(l[-1])
print("headline", d["value"], d["kernel_ms"], d["roofline"]["frac"])
print("e2e", d.get("e2e_GBps"), "cpu", {k:v for k,v in d.get("cpu_baseline",{}).items() if "GBps" in k or k in ("value","all_cores")})
for w,o in d.get("other_workloads",{}).items(): print(w, o["GBps"], o["kernel_ms"], o["host_prepare_s"])
PY
