#!/bin/bash
# L2 / fabric counters of the LZ77 kernels (runs on the GPU box): pmc_sweep.sh [size] [ZGPU_FLAT_T]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc
SIZE=${1:-1000000000}; SHAPE=${2:-1024}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum"; do
  n=$(echo $set | md5sum | cut -c1-6)
  timeout 200 rocprofv3 --output-format csv --pmc $set -d $OUT/$n -o p -- python $ROOT/tools/dev/variants.py $SIZE text -- ZGPU_FLAT_T=$SHAPE > $OUT/$n.log 2>&1
  f=$(find $OUT/$n -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); launches = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    if any(s in k for s in ("flat", "sweep", "seqpost", "calib")):
        print(k[-30:], {c: "%.4g total over %d launches" % (v, cnt[(k, c)]) for c, v in acc[k].items()})
PY
  rm -rf $OUT/$n
done
