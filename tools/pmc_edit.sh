#!/bin/bash
# SQ / memory counters of the edit kernels at C3 (run through gpurun): tools/pmc_edit.sh <outdir> [bench args]
set -u
OUT=${1:-gpurun_out/pmce}; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
	tag=$(echo $set | cut -d' ' -f1)
	timeout 900 rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -o c -- python bench.py --workload edit --steps 1 --warmup 0 --cpu-pairs -1 "$@" > $OUT/$tag.log 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    if "k_edit" in k:
        print(k, {a: "%.4g" % b for a, b in sorted(v.items())})
PY
find $OUT -name '*.db' -delete
