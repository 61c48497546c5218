#!/bin/bash
# PMC passes over the traceback kernels (issue mix, busy / wait cycles): tools/pmc_trace_kernels.sh <tag>
# writes gpurun_out/<tag>_{align8,edit}_pmcN.csv (kernel rows of the counter_collection summary)
cd "$(dirname "$0")/.." ; REPO=$PWD
tag=${1:-pmc}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
P1="SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM"
P2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU"
P3="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_IFETCH SQ_WAVES SQ_ACTIVE_INST_LDS"
for wl in align8 edit; do
	args="--no-secondary --no-exchange"; [ $wl = edit ] && args="--workload edit --no-exchange"
	i=0
	for P in "$P1" "$P2" "$P3"; do
		i=$((i+1))
		rm -rf /tmp/pmc_$wl$i
		timeout -s KILL 600 rocprofv3 --pmc $P --output-format csv -d /tmp/pmc_$wl$i -o p -- python $REPO/bench.py $args --steps 1 --warmup 0 --cpu-pairs -1 > $OUT/${tag}_${wl}_pmc$i.log 2>&1 < /dev/null
		f=$(find /tmp/pmc_$wl$i -name "*counter_collection.csv" | head -1)
		[ -n "$f" ] && python3 - "$f" > $OUT/${tag}_${wl}_pmc$i.txt <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'][:60]
    acc[k][r['Counter_Name']]+=float(r['Counter_Value']); 
for k,v in acc.items():
    if 'trace' in k or 'fwd' in k:
        print(k, {c: '%.4g'%x for c,x in v.items()})
PY
	done
done
cat $OUT/${tag}_*_pmc?.txt
