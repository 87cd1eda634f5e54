#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/c5; mkdir -p $O
cd $ROOT
python tools/dev/bisect.py many one > $O/bisect.log 2>&1; cat $O/bisect.log
python tools/dev/variants.py 1000000000 text -- ZGPU_FLAT=old "" ZGPU_DIRECT=0 > $O/var_text.log 2>&1
cat $O/var_text.log
ZGPU_DEBUG_NO_LITRUN=1 ZGPU_DEBUG_NO_SWEEP=1 ZGPU_SWEEP_MODE=1 python tools/dev/variants.py 1000000000 text -- ZGPU_FLAT=old "" > $O/var_flatonly.log 2>&1
cat $O/var_flatonly.log
python tools/dev/variants.py 268435456 blocks -- ZGPU_FLAT=old "" > $O/var_blocks.log 2>&1
cat $O/var_blocks.log
python tools/dev/configs.py many silesia iso > $O/cfg_new.log 2>&1
ZGPU_FLAT=old python tools/dev/configs.py many silesia iso > $O/cfg_old.log 2>&1
cat $O/cfg_new.log $O/cfg_old.log
cd /tmp; export TMPDIR=/tmp
for v in new old; do
  if [ $v = old ]; then export ZGPU_FLAT=old; else unset ZGPU_FLAT; fi
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
    n=$(echo $set | md5sum | cut -c1-6)
    timeout 200 rocprofv3 --output-format csv --pmc $set -d $O/p$n -o p -- python $ROOT/tools/dev/variants.py 1000000000 text > $O/p$v$n.log 2>&1
    f=$(find $O/p$n -name "*counter_collection.csv" | head -1)
    python - "$f" $v >> $O/pmc_summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    if "flat" in k: print(sys.argv[2], k[-34:], {c: round(v / cnt[(k, c)]) for c, v in acc[k].items()})
PY
    rm -rf $O/p$n
  done
done
cat $O/pmc_summary.txt
