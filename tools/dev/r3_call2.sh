#!/bin/bash
# round 3, second GPU call: parity with the bounded resources, thin boundary, old vs new kernel timings
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/c2; mkdir -p $O
cd $ROOT
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_thin_boundary.py -x -q -m gpu ) > $O/parity.log 2>&1
tail -5 $O/parity.log
python tools/dev/variants.py 1000000000 text -- ZGPU_FLAT=old "" ZGPU_DIRECT=0 ZGPU_FLAT_T=512 > $O/var_text.log 2>&1
cat $O/var_text.log
python tools/dev/variants.py 268435456 blocks -- ZGPU_FLAT=old "" > $O/var_blocks.log 2>&1
cat $O/var_blocks.log
python tools/dev/configs.py many silesia iso > $O/cfg_new.log 2>&1
ZGPU_FLAT=old python tools/dev/configs.py many silesia iso > $O/cfg_old.log 2>&1
cat $O/cfg_new.log $O/cfg_old.log
cd /tmp; export TMPDIR=/tmp
set="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
timeout 300 rocprofv3 --output-format csv --pmc $set -d $O/pmc1 -o p -- python $ROOT/tools/dev/variants.py 1000000000 text > $O/pmc1.log 2>&1
f=$(find $O/pmc1 -name "*counter_collection.csv" | head -1)
python - "$f" > $O/pmc1_summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    if "flat" in k or "lit" in k or "sweep" in k: print(k[-40:], {c: round(v / cnt[(k, c)]) for c, v in acc[k].items()})
PY
cat $O/pmc1_summary.txt
rm -rf $O/pmc1
