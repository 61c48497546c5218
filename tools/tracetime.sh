# kernel times of the edit path for several batch shapes (run on the GPU box): bash tools/tracetime.sh
cd /tmp && export TMPDIR=/tmp
for cfg in "16384 100000" "8192 100000" "4096 100000" "16384 50000" "32768 100000"; do
  set -- $cfg
  rm -rf /tmp/tt; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tt -o t -- python /root/repo/bench.py --workload edit --pairs $1 --length $2 --steps 2 --warmup 1 --cpu-pairs -1 > /dev/null 2>&1 < /dev/null
  python - "$1" "$2" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open('/tmp/tt/t_kernel_stats.csv')) if 'k_edit_trace' in r['Name'] or 'k_edit_fwd' in r['Name']]
print("pairs %s L %s: " % (sys.argv[1], sys.argv[2]) + "; ".join("%s %.1f ms" % (r['Name'][5:22], float(r['AverageNs']) / 1e6) for r in rows))
PY
done
