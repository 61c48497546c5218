#!/bin/bash
# Instruction counts of the dominant kernels per launch (SQ_INSTS_VALU / SQ_INSTS_SALU, one rocprofv3 --pmc pass per workload, run through gpurun):
#   tools/issue_counts.sh <outdir>        -> <outdir>/issue_counts.json, to be copied to profiles/issue_counts.json
# bench.py divides them by the cells of the launch it was measured on and prices them at the measured issue cost
# (profiles/r02f_valu_rate_probe.txt: 4 cycles per packed / three-operand VALU instruction and per scalar instruction of a SIMD's turn).
set -u
OUT=${1:-gpurun_out/issue}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for wl in "align8" "edit" "align8 --scoring 2,-6,-3,-2,-8,-1"; do
	tag=$(echo $wl | tr ' ,-' '___')
	timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/$tag -o sq -- python bench.py --workload $wl --steps 1 --warmup 0 --cpu-pairs -1 > $OUT/$tag.log 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, json, os, sys, collections
out = {}
for d in sorted(glob.glob(sys.argv[1] + "/*/")):
    tag = os.path.basename(d.rstrip("/"))
    line = None
    try:
        line = json.loads([l for l in open(sys.argv[1] + "/" + tag + ".log") if l.startswith("{")][-1])
    except Exception:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    n, L, bw = line["config"]["pairs_per_gpu"], line["config"]["length"], line["config"]["bandwidth"]
    for k, v in agg.items():
        if v.get("SQ_INSTS_VALU", 0) + v.get("SQ_INSTS_SALU", 0) < 1e8:
            continue
        nd = max(1, len(disp[k]))
        out["%s|%s" % (line["config"]["workload"].split(":")[0] + ("_2piece" if "2piece" in tag or "_8__1" in tag else ""), k)] = {
            "valu_per_launch": v.get("SQ_INSTS_VALU", 0) / nd, "salu_per_launch": v.get("SQ_INSTS_SALU", 0) / nd, "launches": nd,
            "pairs": n, "length": L, "bandwidth": bw, "source": "tools/issue_counts.sh (rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU) on bench.py --workload %s" % line["config"]["workload"].split(":")[0]}
json.dump(out, open(sys.argv[1] + "/issue_counts.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $OUT -name '*.db' -delete
