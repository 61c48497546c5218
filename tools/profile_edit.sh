set -u
OUT=gpurun_out/prof_r01g
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o edit -- python bench.py --workload edit --steps 3 --warmup 1 --cpu-pairs -1 > $OUT/edit_bench.log 2>&1 < /dev/null
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $ctr --output-format csv -d $OUT -o edit_pmc_$ctr -- python bench.py --workload edit --steps 1 --warmup 0 --cpu-pairs -1 > $OUT/edit_pmc_$ctr.log 2>&1 < /dev/null
done
timeout 900 python bench.py --workload edit --steps 3 --warmup 1 > $OUT/edit_bench_line.json 2> $OUT/edit_bench_line.err < /dev/null
timeout 900 python bench.py --steps 3 --warmup 1 > $OUT/align8_bench_line.json 2> $OUT/align8_bench_line.err < /dev/null
find $OUT -name '*.db' -delete; find $OUT -name '*_kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete
ls $OUT
