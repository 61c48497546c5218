# bsalign-hip (batched command line) against the reference's own CLI (oracle/_ref/bsalign_ref_cli, one core) and against the
# single-pair path (-B 1), on synthetic short and long pairs; run through gpurun:  bash tools/bench_cli.sh
set -e
D=${TMPDIR:-/tmp}/bench_cli; mkdir -p $D
python - <<PY
import numpy as np, sys
sys.path.insert(0, "tests")
import support as S
rng = np.random.default_rng(5)
def mk(path, n, L):
    with open(path, "w") as f:
        for k in range(n):
            T = rng.integers(0, 4, size=L).astype(np.uint8)
            Q = S.mutate(rng, T, 0.1)
            f.write(">q%d\n%s\n>t%d\n%s\n" % (k, "".join("ACGT"[b] for b in Q), k, "".join("ACGT"[b] for b in T)))
mk("$D/short.fa", 100000, 150)
mk("$D/long.fa", 2000, 10000)
PY
t(){ local s=$(date +%s.%N); "$@" > $D/out.txt; local e=$(date +%s.%N); echo "$(awk -v a=$s -v b=$e 'BEGIN{printf "%7.2f", b - a}') s   $(md5sum < $D/out.txt | cut -c1-12)   $*"; }
REF=oracle/_ref/bsalign_ref_cli; HIP=bsalign_amd/bsalign-hip
echo "== 100000 pairs x 150 bp, global, whole-query band (the CLI default)"
t $HIP align -m global $D/short.fa
[ -x $REF ] && t $REF align -m global $D/short.fa
echo "== the same pairs with no option at all (overlap mode, whole-query band: main.c:262-266)"
t $HIP align $D/short.fa
[ -x $REF ] && t $REF align $D/short.fa
head -c 3100000 $D/short.fa > $D/short_10k.fa
t $HIP align -m global -B 1 $D/short_10k.fa
echo "== 2000 pairs x 10 kbp, global, -W 128"
t $HIP align -m global -W 128 $D/long.fa
[ -x $REF ] && t $REF align -m global -W 128 $D/long.fa
echo "== 2000 pairs x 10 kbp, edit, global, -W 256"
t $HIP edit -m global -W 256 $D/long.fa
[ -x $REF ] && t $REF edit -m global -W 256 $D/long.fa
echo "== 2000 pairs x 10 kbp with no option at all (overlap, whole-query band: example/run.sh 'NoBand' without its scoring)"
t $HIP align $D/long.fa
[ -x $REF ] && t $REF align $D/long.fa
echo "== example/run.sh: NoBand, Band64, Edit0"
t $HIP align -M 2 -X 2 -O 4 -E 2 -Q 0 -P 0 $D/long.fa
[ -x $REF ] && t $REF align -M 2 -X 2 -O 4 -E 2 -Q 0 -P 0 $D/long.fa
t $HIP align -W 64 -M 2 -X 2 -O 4 -E 2 -Q 0 -P 0 -m overlap $D/long.fa
[ -x $REF ] && t $REF align -W 64 -M 2 -X 2 -O 4 -E 2 -Q 0 -P 0 -m overlap $D/long.fa
t $HIP edit -W 0 $D/long.fa
[ -x $REF ] && t $REF edit -W 0 $D/long.fa
