#!/bin/bash
# tiled (row format 1) against format 0 of the edit path under the profiler: bytes, instruction counts, kernel times (8192 pairs)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in 1 0; do
  BSA_EDIT_TILED=$t timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d gpurun_out/ed -o t${t}_SQ -- python bench.py --workload edit --pairs 8192 --steps 1 --warmup 0 --cpu-pairs -1 --no-exchange > /dev/null 2>&1
done
python3 - <<'PY'
import csv,glob,collections
for t in (1,0):
    acc=collections.OrderedDict()
    for fn in glob.glob("gpurun_out/ed/**/t%d_SQ_counter_collection.csv"%t,recursive=True):
        for r in csv.DictReader(open(fn)):
            k=r["Kernel_Name"].split("(")[0][:34]
            if "trace_wave" in k or "grp32" in k: acc[(k,r["Counter_Name"])]=acc.get((k,r["Counter_Name"]),0)+float(r["Counter_Value"])
    for k,v in acc.items(): print("tiled",t,k,"%.3e"%v)
PY
