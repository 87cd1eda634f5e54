#!/bin/bash
# SQ counters of the LZ77 kernels on a 256 MiB text frame (runs on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
  n=$(echo $set | md5sum | cut -c1-6)
  timeout 200 rocprofv3 --output-format csv --pmc $set -d $OUT/$n -o p -- python $ROOT/tools/dev/variants.py 1000000000 text > $OUT/$n.log 2>&1
  f=$(find $OUT/$n -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    if "flat" in k or "sweep" in k or "seq" in k:
        print(k[-30:], {c: round(v / cnt[(k, c)]) for c, v in acc[k].items()})
PY
  rm -rf $OUT/$n
done
