#!/bin/bash
# SQ instruction / cycle counters of the 8-bit forward kernels in one rocprofv3 --pmc pass per setting (run through gpurun):
#   tools/pmc_align8.sh <outdir> [bench args...]      env: BSA_ALIGN8_X_LANES, BSA_ALIGN8_FWD as for the library
set -u
OUT=${1:-gpurun_out/pmc}; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY \
	--output-format csv -d $OUT -o sq -- python bench.py --steps 1 --warmup 0 --cpu-pairs -1 "$@" > $OUT/bench.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    if v.get("SQ_INSTS_VALU", 0) > 1e8:
        print(k, {a: "%.4g" % b for a, b in sorted(v.items())})
PY
find $OUT -name '*.db' -delete
