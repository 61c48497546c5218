#!/bin/bash
# A/B of two builds of the library on one bench workload (GPU box):  tools/ab_lib.sh <tag> <lib path | ""> [bench args...]
# prints the bench line's figures and the WRITE_SIZE / FETCH_SIZE totals per kernel of one untimed step.
set -u
TAG=$1; LIB=$2; shift 2
OUT=gpurun_out/ab_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
[ -n "$LIB" ] && export BSA_LIB_PATH="$GRAFT_REPO_ROOT/$LIB"
python bench.py "$@" --steps 5 --warmup 2 --cpu-pairs -1 2>/dev/null | python tools/sumline.py
for ctr in WRITE_SIZE FETCH_SIZE; do
	timeout -s KILL 600 rocprofv3 --pmc $ctr --output-format csv -d $OUT -o pmc_$ctr -- python bench.py "$@" --steps 1 --warmup 0 --cpu-pairs -1 > $OUT/pmc_$ctr.log 2>&1 < /dev/null
	python - "$OUT" "$ctr" <<'PY'
import sys, glob, csv, collections
out, ctr = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/**/pmc_%s_counter_collection.csv" % ctr, recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
for k, (v, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:4]:
    print("   %s %-60s total %.3f GB-units (x1024 if KB)  dispatch rows %d" % (ctr, k[:60], v / 1e6, n))
PY
done
find $OUT -name '*.db' -delete
