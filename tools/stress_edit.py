"""Randomised GPU-vs-oracle campaign for the 2-bit edit path (run on the GPU box: gpurun -- python tools/stress_edit.py SEED N)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bsalign_amd as B  # noqa: E402
import support as S  # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    nbatch = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rng = np.random.default_rng(seed)
    ctx = B.Context(0)
    tot = bad = 0
    for b in range(nbatch):
        mode = int(rng.integers(3))
        bw = int(rng.choice([0, 64, 128, 256, 512, 1024, 2048])) if mode == 0 else 0
        pairs = []
        for _ in range(int(rng.integers(60, 200))):
            L = int(rng.choice([1, 2, 63, 64, 65, 100, 300, 700, 1100, 2500, 5000]))
            T = rng.integers(0, 4, size=L).astype(np.uint8)
            Q = S.mutate(rng, T, float(rng.choice([0.0, 0.02, 0.1, 0.2, 0.35])))
            r = float(rng.choice([1.0, 1.0, 1.0, 0.85, 1.15, 1.5, 0.7]))
            if r != 1.0:
                Lq = max(1, int(len(Q) * r))
                Q = Q[:Lq] if Lq <= len(Q) else np.concatenate([Q, rng.integers(0, 4, size=Lq - len(Q)).astype(np.uint8)])
            if len(Q) == 0:
                Q = np.array([1], np.uint8)
            pairs.append((Q, T))
        out, cigs, status = ctx.edit_batch(pairs, mode, bw)
        nb = 0
        for k, (q, t) in enumerate(pairs):
            res, cig, n = S.oracle_edit(q, t, mode, bw)
            got = np.array([out[k][f] for f in out.dtype.names], dtype=np.int32)
            if n < 0:
                ok = status[k] != 0
            else:
                ok = status[k] == 0 and np.array_equal(got, res) and np.array_equal(cigs[k], cig)
            if not ok:
                nb += 1
                if bad + nb <= 5:
                    print("DIFF mode", mode, "bw", bw, "qlen", len(q), "tlen", len(t), "status", status[k], got, res, n)
        tot += len(pairs)
        bad += nb
        print("batch %d mode %d bw %d pairs %d diff %d" % (b, mode, bw, len(pairs), nb), flush=True)
    print("TOTAL pairs %d diff %d" % (tot, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
