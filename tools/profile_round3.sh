#!/bin/bash
# Round-3 rocprofv3 evidence (run on the GPU box through gpurun):  tools/profile_round3.sh <tag>   e.g. r03d
# kernel-trace stats of the bench commands and, in separate --pmc passes, FETCH_SIZE / WRITE_SIZE of the dominant kernels:
#   align8    the default bench (C2: 100 k x 10 kbp, bandwidth 128)
#   align8wq  whole-query bands, `bsalign align -W 0` on 10 kbp pairs (the systolic kernel)
#   poarec    recorded POA windows, 256 and 4096 in flight (row-at-a-time forward pass + ring traceback)
#   edit      the default edit bench (C3), stats only
set -u
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run_stats(){ local name=$1; shift
	timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${name} -- python bench.py "$@" --cpu-pairs -1 > $OUT/${name}_bench.log 2>&1 < /dev/null; }
run_pmc(){ local name=$1 ctr=$2; shift 2
	timeout -s KILL 600 rocprofv3 --pmc $ctr --output-format csv -d $OUT -o ${name}_pmc_${ctr} -- python bench.py "$@" --cpu-pairs -1 > $OUT/${name}_pmc_${ctr}.log 2>&1 < /dev/null; }
WQ="--length 10000 --bw -1 --pairs 2048"
run_stats align8 --steps 3 --warmup 1
run_pmc align8 FETCH_SIZE --steps 1 --warmup 0
run_pmc align8 WRITE_SIZE --steps 1 --warmup 0
run_stats align8wq $WQ --steps 3 --warmup 1
run_pmc align8wq FETCH_SIZE $WQ --steps 1 --warmup 0
run_pmc align8wq WRITE_SIZE $WQ --steps 1 --warmup 0
run_stats poarec --workload poa --steps 2 --warmup 1
run_pmc poarec FETCH_SIZE --workload poa --steps 1 --warmup 0
run_pmc poarec WRITE_SIZE --workload poa --steps 1 --warmup 0
run_stats poarec4096 --workload poa --pairs 4096 --steps 2 --warmup 1
run_stats align8wq4096 --length 10000 --bw -1 --pairs 4096 --steps 2 --warmup 1
run_stats edit --workload edit --steps 3 --warmup 1
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
	timeout -s KILL 600 rocprofv3 --pmc $set --output-format csv -d $OUT -o align8wq_pmc_sq -- python bench.py $WQ --steps 1 --warmup 0 --cpu-pairs -1 > $OUT/align8wq_pmc_sq.log 2>&1 < /dev/null
	timeout -s KILL 600 rocprofv3 --pmc $set --output-format csv -d $OUT -o poarec_pmc_sq -- python bench.py --workload poa --steps 1 --warmup 0 --cpu-pairs -1 > $OUT/poarec_pmc_sq.log 2>&1 < /dev/null
done
# the bench lines themselves (with the CPU baseline), outside the profiler
timeout -s KILL 600 python bench.py --steps 3 --warmup 1 > $OUT/align8_bench_line.json 2> $OUT/align8_bench_line.err < /dev/null
timeout -s KILL 600 python bench.py $WQ --steps 3 --warmup 1 --cpu-pairs 40 > $OUT/align8wq_bench_line.json 2> $OUT/align8wq_bench_line.err < /dev/null
timeout -s KILL 600 python bench.py --workload poa --steps 2 --warmup 1 > $OUT/poarec_bench_line.json 2> $OUT/poarec_bench_line.err < /dev/null
timeout -s KILL 600 python bench.py --workload poa --pairs 4096 --steps 2 --warmup 1 > $OUT/poarec4096_bench_line.json 2> $OUT/poarec4096_bench_line.err < /dev/null
timeout -s KILL 600 python bench.py --workload edit --steps 3 --warmup 1 > $OUT/edit_bench_line.json 2> $OUT/edit_bench_line.err < /dev/null
find $OUT -name '*.db' -delete
find $OUT -name '*_kernel_trace.csv' -delete
find $OUT -name '*agent_info.csv' -delete
ls -la $OUT | head -60
