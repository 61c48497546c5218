"""Randomised GPU-vs-oracle campaign for the 8-bit path (run on the GPU box: gpurun -- python tools/stress_align8.py SEED N).
Every batch draws a mode, a bandwidth and a scoring, aligns a few hundred random pairs through the C-ABI and compares
result structs, CIGARs and status bits with the literal oracle.  Prints one summary line per batch and a final tally."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bsalign_amd as B  # noqa: E402
import support as S  # noqa: E402

SCORINGS = [(2, -6, -3, -2, 0, 0), (2, -2, -4, -2, 0, 0), (2, -6, 0, -3, 0, 0), (1, -1, -1, -1, 0, 0), (3, -4, -6, -1, 0, 0),
            (5, -10, -8, -4, 0, 0), (4, -8, 0, -6, 0, 0), (2, -6, -3, -2, -8, -1), (10, -30, -20, -10, 0, 0)]


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    nbatch = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rng = np.random.default_rng(seed)
    ctx = B.Context(0)
    tot = bad = flagged = 0
    for b in range(nbatch):
        mode = int(rng.integers(3))
        bw = int(rng.choice([0, 16, 32, 48, 64, 128, 256]))
        sc = SCORINGS[int(rng.integers(len(SCORINGS)))]
        # STRESS_MODE / STRESS_BW / STRESS_SC (indices into SCORINGS) narrow the campaign, e.g. to one forward kernel's domain
        if os.environ.get("STRESS_MODE"):
            mode = int(rng.choice([int(x) for x in os.environ["STRESS_MODE"].split(",")]))
        if os.environ.get("STRESS_BW"):
            bw = int(rng.choice([int(x) for x in os.environ["STRESS_BW"].split(",")]))
        if os.environ.get("STRESS_SC"):
            sc = SCORINGS[int(rng.choice([int(x) for x in os.environ["STRESS_SC"].split(",")]))]
        pairs = []
        for _ in range(int(rng.integers(100, 400))):
            L = int(rng.choice([1, 2, 15, 16, 17, 33, 64, 100, 300, 700, 1500, 3000]))
            T = rng.integers(0, 4, size=L).astype(np.uint8)
            Q = S.mutate(rng, T, float(rng.choice([0.0, 0.02, 0.1, 0.2, 0.4])))
            r = float(rng.choice([1.0, 1.0, 1.0, 0.8, 1.25, 2.0, 0.5, 0.3]))
            if r != 1.0:
                Lq = max(1, int(len(Q) * r))
                Q = Q[:Lq] if Lq <= len(Q) else np.concatenate([Q, rng.integers(0, 4, size=Lq - len(Q)).astype(np.uint8)])
            if mode and rng.random() < 0.3 and len(Q) > 10:
                Q = Q[int(len(Q) * 0.3):]
            if len(Q) == 0:
                Q = np.array([1], np.uint8)
            if os.environ.get("STRESS_MAXQ"):          # cap the query length (whole-query bands of at most 256 columns run widened, bsa_api.hip)
                Q = Q[:int(os.environ["STRESS_MAXQ"])]
                T = T[:4 * int(os.environ["STRESS_MAXQ"])]
            pairs.append((Q, T))
        out, cigs, status = ctx.align_batch(pairs, B.make_params(mode, bw, *sc))
        nb = nf = 0
        for k, (q, t) in enumerate(pairs):
            res, cig, n = S.oracle_align(q, t, mode, bw, *sc)
            if n == S.ORC_ERR_TRACE:
                ok = bool(status[k] & B.ST_TRACE)
                nf += 1
            else:
                got = np.array([out[k][f] for f in out.dtype.names], dtype=np.int32)
                ok = status[k] == 0 and np.array_equal(got, res) and np.array_equal(cigs[k], cig)
            if not ok:
                nb += 1
                if bad + nb <= 5:
                    print("DIFF mode", mode, "bw", bw, "sc", sc, "qlen", len(q), "tlen", len(t), "status", status[k], out[k], res)
        tot += len(pairs)
        bad += nb
        flagged += nf
        print("batch %d mode %d bw %d sc %s pairs %d diff %d reference-nonterminating %d" % (b, mode, bw, sc, len(pairs), nb, nf), flush=True)
    print("TOTAL pairs %d diff %d reference-nonterminating %d" % (tot, bad, flagged))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
