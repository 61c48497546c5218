# End-to-end parity campaign through the two command lines: bsalign-hip against the reference's own CLI (oracle/_ref/bsalign_ref_cli) on the
# same FASTA, option set by option set, compared by checksum of the whole stdout.  Run through gpurun:  bash tools/parity_cli.sh
D=${TMPDIR:-/tmp}/parity_cli; mkdir -p $D
python - <<PY
import numpy as np, sys
sys.path.insert(0, "tests")
import support as S
rng = np.random.default_rng(99)
def mk(path, n, lens, eps):
    with open(path, "w") as f:
        for k in range(n):
            T = rng.integers(0, 4, size=int(rng.choice(lens))).astype(np.uint8)
            Q = S.mutate(rng, T, float(rng.choice(eps)))
            if k % 7 == 0 and len(Q) > 20: Q = Q[len(Q) // 4:]
            if len(Q) == 0: Q = T[:1]
            f.write(">q%d\n%s\n>t%d\n%s\n" % (k, "".join("ACGT"[b] for b in Q), k, "".join("ACGT"[b] for b in T)))
mk("$D/short.fa", 50000, [30, 50, 75, 100, 125, 150, 200, 250], [0.0, 0.02, 0.1, 0.2])
mk("$D/mid.fa", 4000, [300, 700, 1500, 3000], [0.05, 0.1, 0.15])
PY
REF=oracle/_ref/bsalign_ref_cli; HIP=bsalign_amd/bsalign-hip
bad=0; tot=0
# (the reference's traceback crashes or hangs on rare inputs, HISTORY section 2: then the records it printed before are compared, and bsalign-hip --
# which reports such a pair on stderr and goes on -- must agree with them)
chk(){
	$HIP "$@" > $D/a.txt 2>/dev/null; timeout 600 $REF "$@" > $D/b.txt 2>/dev/null; local rc=$?
	tot=$((tot+1))
	if [ $rc = 0 ]; then
		if cmp -s $D/a.txt $D/b.txt; then echo "same   $(md5sum < $D/a.txt | cut -c1-16)   $*"; else echo "DIFF   $*"; bad=$((bad+1)); fi
	else
		local n=$(( $(wc -l < $D/b.txt) / 4 * 4 ))
		if [ "$(head -n $n $D/a.txt | md5sum)" = "$(head -n $n $D/b.txt | md5sum)" ]; then echo "same   up to the record where the reference stops (exit $rc after $((n / 4)) records)   $*"; refdied=$((refdied+1)); else echo "DIFF   (reference exit $rc)   $*"; bad=$((bad+1)); fi
	fi
}
refdied=0
for f in short mid; do
  for m in global overlap extend; do
    chk align -m $m $D/$f.fa
    chk align -m $m -W 128 $D/$f.fa
    chk align -m $m -W 100 $D/$f.fa
    chk align -m $m -M 2 -X 2 -O 4 -E 2 $D/$f.fa
    chk align -m $m -O 0 -E 3 -W 64 $D/$f.fa
    chk edit -m $m $D/$f.fa
    chk edit -m $m -W 64 $D/$f.fa
  done
  chk align -m global -Q 8 -P 1 $D/$f.fa
  chk align -m global -Q 8 -P 1 -W 128 $D/$f.fa
  chk edit -m kmer -k 11 $D/$f.fa
done
echo "option sets: $tot, different: $bad, reference crashed or hung in: $refdied"
