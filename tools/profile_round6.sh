#!/bin/bash
# Round-6 rocprofv3 evidence (same recipe as round 4) (run on the GPU box through gpurun):  tools/profile_round5.sh <tag> [workload names...]
# Per workload: kernel-trace statistics of the bench command, then -- each in a pass of its own, as the MI355X guide prescribes --
# FETCH_SIZE, WRITE_SIZE and the SQ instruction counters, and the bench line itself outside the profiler.
# tools/summarize_round4.py <tag> turns the output into profiles/<tag>_* and the shape-keyed profiles/counters.json.
set -u
TAG=${1:-r06}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
declare -A WL
WL[align8]=""
WL[align8_2piece]="--scoring 2,-6,-3,-2,-8,-1"
WL[align8_linear]="--scoring 2,-6,0,-3,0,0"
WL[align8_overlap]="--mode overlap"
WL[align8wq]="--length 10000 --bw -1 --pairs 2048"
WL[align8_gen]="--length 10000 --bw -1 --pairs 2048 --scoring 10,-30,-20,-10,0,0"
WL[edit]="--workload edit"
WL[poarec]="--workload poa"
WL[poarec4096]="--workload poa --pairs 4096"
NAMES=${@:-align8 align8_2piece edit align8wq poarec poarec4096}
for name in $NAMES; do
	args=${WL[$name]}
	timeout -s KILL 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${name} -- python bench.py $args --steps 3 --warmup 1 --cpu-pairs -1 > $OUT/${name}_stats.log 2>&1 < /dev/null
	for ctr in FETCH_SIZE WRITE_SIZE; do
		timeout -s KILL 900 rocprofv3 --pmc $ctr --output-format csv -d $OUT -o ${name}_pmc_${ctr} -- python bench.py $args --steps 1 --warmup 0 --cpu-pairs -1 > $OUT/${name}_pmc_${ctr}.log 2>&1 < /dev/null
	done
	timeout -s KILL 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT -o ${name}_pmc_SQ -- python bench.py $args --steps 1 --warmup 0 --cpu-pairs -1 > $OUT/${name}_pmc_SQ.log 2>&1 < /dev/null
	timeout -s KILL 900 python bench.py $args --steps 5 --warmup 2 > $OUT/${name}_bench_line.json 2> $OUT/${name}_bench_line.err < /dev/null
done
find $OUT -name '*.db' -delete
find $OUT -name '*_kernel_trace.csv' -delete
find $OUT -name '*agent_info.csv' -delete
ls $OUT | head -80
