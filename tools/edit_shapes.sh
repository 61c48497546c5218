run(){ python bench.py --workload edit --steps 2 --warmup 1 --cpu-pairs -1 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; o=r['other_kernel']; print(d['value'], d['ms_per_step'], r['kernel'][:22], r['kernel_ms_avg'], o['kernel'][:18], o['kernel_ms_avg'])"; }
for shape in "--pairs 65536 --length 20000 --bw 512" "--pairs 65536 --length 20000 --bw 256" "--pairs 262144 --length 3000 --bw 128" "--pairs 32768 --length 50000 --bw 256"; do
  echo "== $shape"
  for g in 0 1; do for w in 0 1; do echo -n "grp32=$g wave=$w: "; BSA_EDIT_GRP32=$g BSA_EDIT_TRACE_WAVE=$w run $shape; done; done
done
