# 8-bit compact path: traceback kernel choice over batch shapes (run through gpurun)
run(){ python bench.py --steps 2 --warmup 1 --cpu-pairs -1 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; o=r['other_kernel']; print(d['value'], d['ms_per_step'], r['kernel_ms_avg'], o['kernel'][:28], o['kernel_ms_avg'])"; }
for shape in "--pairs 1000000 --length 1000" "--pairs 400000 --length 2500" "--pairs 2000000 --length 300 --bw 64" "--pairs 20000 --length 50000"; do
  echo "== $shape"
  for w in 0 1; do echo -n "wave=$w: "; BSA_ALIGN8_TRACE_WAVE=$w run $shape; done
done
# whole-query bands of short reads (the reference CLI's default -W 0): widened dispatch + static-band kernels against the run-time-width kernel
for shape in "--pairs 2000000 --length 150 --bw -1" "--pairs 2000000 --length 100 --bw -1" "--pairs 4000000 --length 50 --bw -1"; do
  echo "== $shape"
  echo -n "default:           "; run $shape
  echo -n "moving-band kernel: "; BSA_ALIGN8_NO_STATIC=1 run $shape
  echo -n "run-time width:     "; BSA_ALIGN8_WIDEN=0 run $shape
done
