#!/bin/bash
# Reproduce the rocprofv3 evidence kept under profiles/ (run on the GPU box through gpurun):
#   tools/profile_round.sh <tag>     e.g. r01d
# writes gpurun_out/prof_<tag>/{align8,edit,poa}_* : kernel-trace stats of the default bench commands, and for the
# dominant kernels FETCH_SIZE / WRITE_SIZE in separate --pmc passes (never combined with tracing, see the task notes).
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run_stats(){ # name, bench args...
	local name=$1; shift
	timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${name} -- python bench.py "$@" --cpu-pairs -1 > $OUT/${name}_bench.log 2>&1
}
run_pmc(){ # name, counter, bench args...
	local name=$1 ctr=$2; shift 2
	timeout 900 rocprofv3 --pmc $ctr --output-format csv -d $OUT -o ${name}_pmc_${ctr} -- python bench.py "$@" --cpu-pairs -1 > $OUT/${name}_pmc_${ctr}.log 2>&1
}
run_stats align8 --steps 3 --warmup 1
run_pmc align8 FETCH_SIZE --steps 1 --warmup 0
run_pmc align8 WRITE_SIZE --steps 1 --warmup 0
run_stats edit --workload edit --steps 3 --warmup 1
run_pmc edit FETCH_SIZE --workload edit --steps 1 --warmup 0
run_pmc edit WRITE_SIZE --workload edit --steps 1 --warmup 0
run_stats poa --workload poa --steps 2 --warmup 1
run_pmc poa FETCH_SIZE --workload poa --steps 1 --warmup 0
run_pmc poa WRITE_SIZE --workload poa --steps 1 --warmup 0
# full-width edit bands (extend mode: the wave-per-pair kernel k_edit_fwd_wide) and the k-mer anchored edit alignment
EF="--workload edit --mode extend --bw -1 --pairs 8192 --length 15000"
run_stats editfull $EF --steps 3 --warmup 1
run_pmc editfull FETCH_SIZE $EF --steps 1 --warmup 0
run_pmc editfull WRITE_SIZE $EF --steps 1 --warmup 0
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kmer -- python tools/bench_kmer.py 8192 10000 13 noplain > $OUT/kmer_bench.log 2>&1 < /dev/null
BSA_KMER_TIMING=1 BSA_BATCH_TIMING=1 timeout 600 python tools/bench_kmer.py 8192 10000 13 cpu > $OUT/kmer_timing.log 2>&1 < /dev/null
# SQ instruction counters of the align8 kernels (VALU / SALU / LDS / VMEM per launch): three more --pmc passes
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM"; do
	tag=$(echo $set | cut -d' ' -f1)
	timeout 900 rocprofv3 --pmc $set --output-format csv -d $OUT -o align8_pmc_sq_$tag -- python bench.py --steps 1 --warmup 0 --cpu-pairs -1 > $OUT/align8_pmc_sq_$tag.log 2>&1
done
# the bench lines themselves (with the CPU baseline), outside the profiler
timeout 900 python bench.py --steps 3 --warmup 1 > $OUT/align8_bench_line.json 2> $OUT/align8_bench_line.err
timeout 900 python bench.py --workload edit --steps 3 --warmup 1 > $OUT/edit_bench_line.json 2> $OUT/edit_bench_line.err
timeout 900 python bench.py --workload poa --steps 2 --warmup 1 > $OUT/poa_bench_line.json 2> $OUT/poa_bench_line.err
timeout 900 python bench.py $EF --steps 3 --warmup 1 --cpu-pairs 40 > $OUT/editfull_bench_line.json 2> $OUT/editfull_bench_line.err
# keep only the small summaries (the merge back is capped at 64 MiB)
find $OUT -name '*.db' -delete
find $OUT -name '*_kernel_trace.csv' -delete
find $OUT -name '*agent_info.csv' -delete
ls -la $OUT
