#!/bin/bash
# round 5, GPU call 1: the chain microbenchmark; zg_k_huf's stores plain vs nontemporal (time + WRITE_SIZE); text baseline
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c1
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
( time timeout 300 $ROOT/tools/dev/chain_bench 255 ) > $OUT/chain_bench.log 2>&1
for lib in libzgpu.so libzgpu_nt.so; do
  ZGPU_LIB=$ROOT/zstd-rs_amd/$lib timeout 300 python $ROOT/tools/dev/variants.py 4294967296 isomany > $OUT/iso_$lib.log 2>&1
  ZGPU_LIB=$ROOT/zstd-rs_amd/$lib timeout 300 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/w_$lib -o pmc -- python $ROOT/tools/dev/variants.py 1073741824 isomany > $OUT/isow_$lib.log 2>&1
  f=$(find $OUT/w_$lib -name "*counter_collection.csv" | head -1)
  python - "$f" > $OUT/write_$lib.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    acc[k] += float(r["Counter_Value"]); cnt[k] += 1
for k in sorted(acc, key=lambda k: -acc[k])[:8]:
    print("%-40s launches %4d  WRITE_SIZE sum %.1f  per launch %.1f" % (k[-40:], cnt[k], acc[k], acc[k] / cnt[k]))
PY
  rm -rf $OUT/w_$lib
done
ZGPU_LIB=$ROOT/zstd-rs_amd/libzgpu.so timeout 300 python $ROOT/tools/dev/variants.py 1000000000 text > $OUT/text_plain.log 2>&1
ZGPU_LIB=$ROOT/zstd-rs_amd/libzgpu_nt.so timeout 300 python $ROOT/tools/dev/variants.py 1000000000 text > $OUT/text_nt.log 2>&1
tail -n 30 $OUT/chain_bench.log; cat $OUT/iso_*.log $OUT/write_*.txt $OUT/text_*.log | cut -c1-400
