#!/bin/bash
# interleaved A/B of two library builds on one bench workload (GPU box): tools/abab.sh <libB> <reps> [bench args...]; A = the in-tree library
LIBB=$1; REPS=$2; shift 2
for r in $(seq 1 $REPS); do
	for which in A B; do
		if [ $which = B ]; then export BSA_LIB_PATH="$GRAFT_REPO_ROOT/$LIBB"; else unset BSA_LIB_PATH; fi
		echo -n "$which: "; python bench.py "$@" --steps 5 --warmup 2 --cpu-pairs -1 2>/dev/null | python tools/sumline.py | cut -c1-200
	done
done
