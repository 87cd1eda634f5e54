#!/bin/bash
# what zg_k_huf waits for (SQ counters on 1 GiB of iso-like frames)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT; mkdir -p gpurun_out/huf_sq
cd /tmp && export TMPDIR=/tmp
k=1
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES_EQ_64 SQ_THREAD_CYCLES_VALU"; do
  timeout 300 rocprofv3 --output-format csv --pmc $set -d $ROOT/gpurun_out/huf_sq/sq$k -o pmc -- python $ROOT/tools/dev/variants.py 1073741824 isomany -- "" > $ROOT/gpurun_out/huf_sq/sq$k.log 2>&1
  k=$((k+1))
done
cd $ROOT
python - <<'PY'
import csv, glob, collections
for k in (1, 2, 3, 4):
    for f in glob.glob("gpurun_out/huf_sq/sq%d/**/*counter_collection.csv" % k, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"].split("(")[0]
            acc[kn][r["Counter_Name"]] += float(r["Counter_Value"])
        for kn in acc:
            if "huf" in kn or "flatten" in kn:
                print(k, kn, {c: "%.4g" % v for c, v in acc[kn].items()})
PY
find gpurun_out/huf_sq -name "*.csv" -size +2M -delete
