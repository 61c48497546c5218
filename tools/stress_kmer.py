"""Randomised campaign for the k-mer anchored edit alignment (run on the GPU box: gpurun -- python tools/stress_kmer.py SEED BATCHES).
Device batches against the host pieces driven with the oracle's edit DP (tests/kmer_support.py), and, where the
reference library was built (oracle/_ref), a sample of every batch against the reference itself."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bsalign_amd as B  # noqa: E402
import support as S  # noqa: E402
import kmer_support as K  # noqa: E402


def oracle_segment(qs, ts, mode):
    r, c, n = S.oracle_edit(qs, ts, mode, 0)
    return r, c


def make_pair(rng):
    L = int(rng.choice([30, 200, 800, 2500, 6000, 12000]))
    T = rng.integers(0, 4, L).astype(np.uint8)
    eps = float(rng.choice([0.0, 0.01, 0.05, 0.1, 0.18, 0.3]))
    Q = S.mutate(rng, T, eps)
    kind = int(rng.integers(8))
    if kind == 1 and len(Q) > 300:
        a = int(rng.integers(50, len(Q) - 200))
        Q = np.concatenate([Q[:a], Q[a + int(rng.integers(30, 200)):]])
    elif kind == 2 and len(Q) > 300:
        a = int(rng.integers(50, len(Q) - 100))
        Q = np.concatenate([Q[:a], rng.integers(0, 4, int(rng.integers(30, 600))).astype(np.uint8), Q[a:]])
    elif kind == 3 and len(Q) > 600:
        a = int(rng.integers(50, len(Q) // 2))
        b = a + int(rng.integers(50, 200))
        c = int(rng.integers(b, len(Q) - 10))
        Q = np.concatenate([Q[:a], Q[b:c], Q[a:b], Q[c:]])
    elif kind == 4:
        unit = rng.integers(0, 4, int(rng.integers(1, 40))).astype(np.uint8)
        a = int(rng.integers(0, len(T)))
        T = np.concatenate([T[:a], np.tile(unit, int(rng.integers(3, 40))), T[a:]])
        Q = S.mutate(rng, T, eps)
    elif kind == 5:
        Q = Q[int(rng.integers(0, min(80, len(Q) // 2 + 1))):]
        T = T[:len(T) - int(rng.integers(0, min(80, len(T) // 2 + 1)))]
    elif kind == 6:
        Q = np.concatenate([rng.integers(0, 4, int(rng.integers(0, 3000))).astype(np.uint8), Q])      # a long head
    elif kind == 7:
        T = np.concatenate([T, rng.integers(0, 4, int(rng.integers(0, 3000))).astype(np.uint8)])      # a long tail
    if len(Q) == 0:
        Q = T[:1].copy()
    return Q, T


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    nbatch = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    rng = np.random.default_rng(seed)
    ctx = B.Context(0)
    tot = bad = refbad = reftot = 0
    for b in range(nbatch):
        ksz = int(rng.choice([5, 8, 11, 13, 15]))
        pairs = [make_pair(rng) for _ in range(int(rng.integers(40, 120)))]
        out, cigs, st = ctx.kmer_edit_batch(pairs, ksz=ksz, threads=int(rng.integers(0, 3)))
        nb = 0
        for k, (q, t) in enumerate(pairs):
            r, c, _ = K.kmer_host(ksz, q, t, oracle_segment)
            ok = st[k] == 0 and np.array_equal(np.array(out[k].tolist(), dtype=np.int32), r) and np.array_equal(cigs[k], c)
            nb += not ok
            if S.have_ref() and k % 8 == 0:
                r0, c0 = K.ref_kmer_edit(ksz, q, t)
                reftot += 1
                refbad += not (np.array_equal(r0, r) and np.array_equal(c0, c))
        tot += len(pairs)
        bad += nb
        print("batch %d ksz %d pairs %d diff %d" % (b, ksz, len(pairs), nb), flush=True)
    print("TOTAL pairs %d diff %d; host pieces vs reference: %d of %d differ" % (tot, bad, refbad, reftot))
    return 1 if bad or refbad else 0


if __name__ == "__main__":
    sys.exit(main())
