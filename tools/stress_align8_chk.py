"""Randomised campaign for the CHECKED whole-query kernel (k_align8_fwd_sys<CHK>, bsa_align8_sys.hip): scores drawn from outside the
static exact-arithmetic guard, whole-query bands above 256 columns, all three modes and gap models; every pair's result struct, CIGAR
and status against the lane-exact oracle.  Reports how many pairs the kernel flagged (re-run by the literal kernels).
Run on the GPU box: gpurun -- python tools/stress_align8_chk.py SEED NBATCH"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bsalign_amd as B  # noqa: E402
import support as S  # noqa: E402


def draw_scoring(rng):
    """a scoring the checked kernel accepts (bsa_align8_sys_supported() == 2) or, now and then, one inside the guard"""
    while True:
        kind = int(rng.integers(3))            # 0 linear, 1 affine, 2 two-piece
        m = int(rng.integers(1, 40)); n = int(rng.integers(1, 60))
        ge = int(rng.integers(1, 40)); go = 0 if kind == 0 else int(rng.integers(1, 50))
        g = go + ge
        if g > 63 or g + n + m > 128:
            continue
        if g > 31 and rng.random() < 0.75:      # above 31 the -63 restart of the running blocks binds and nearly every pair is handed over: mostly draw what the checked kernel decides itself
            continue
        if kind == 2:
            ge2 = int(rng.integers(0, ge)); go2 = int(rng.integers(go + 1, 64))
            if go2 + ge2 > 63 or go2 + ge2 <= g or go2 + ge2 + n + m > 128 or ge2 >= ge:
                continue
            return (m, -n, -go, -ge, -go2, -ge2)
        return (m, -n, -go, -ge, 0, 0)


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    nbatch = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rng = np.random.default_rng(seed)
    ctx = B.Context(0)
    tot = bad = nonterm = handed = chk = 0
    for b in range(nbatch):
        mode = int(rng.integers(3))
        sc = draw_scoring(rng)
        pairs = []
        for _ in range(int(rng.integers(40, 120))):
            L = int(rng.choice([260, 300, 400, 513, 700, 1000, 1500, 2500]))
            kind = rng.random()
            if kind < 0.1:                           # low-complexity target: long homopolymer / repeat runs
                T = np.repeat(rng.integers(0, 4, size=L // 8 + 1), 8)[:L].astype(np.uint8)
            else:
                T = rng.integers(0, 4, size=L).astype(np.uint8)
            Q = S.mutate(rng, T, float(rng.choice([0.0, 0.02, 0.1, 0.2, 0.4, 0.75])))
            r = float(rng.choice([1.0, 1.0, 1.0, 0.8, 1.25, 0.5]))
            if r != 1.0:
                Lq = max(1, int(len(Q) * r))
                Q = Q[:Lq] if Lq <= len(Q) else np.concatenate([Q, rng.integers(0, 4, size=Lq - len(Q)).astype(np.uint8)])
            if len(Q) <= 256:
                Q = np.concatenate([Q, rng.integers(0, 4, size=257 - len(Q)).astype(np.uint8)])
            if rng.random() < 0.15:
                T = T[:int(rng.integers(1, 200))]
            pairs.append((Q, T))
        out, cigs, status = ctx.align_batch(pairs, B.make_params(mode, 0, *sc))
        name = ctx.last_kernel_names()[0]
        ho = ctx.last_handover()
        nb = nf = 0
        for k, (q, t) in enumerate(pairs):
            res, cig, n = S.oracle_align(q, t, mode, 0, *sc)
            if n == S.ORC_ERR_TRACE:
                ok = bool(status[k] & B.ST_TRACE)
                nf += 1
            else:
                got = np.array([out[k][f] for f in out.dtype.names], dtype=np.int32)
                ok = status[k] == 0 and np.array_equal(got, res) and np.array_equal(cigs[k], cig)
            if not ok:
                nb += 1
                if bad + nb <= 8:
                    print("DIFF mode", mode, "sc", sc, "qlen", len(q), "tlen", len(t), "status", status[k], out[k], res, flush=True)
        tot += len(pairs); bad += nb; nonterm += nf; handed += ho
        chk += len(pairs) if "CHK" in name else 0
        print("batch %d mode %d sc %s pairs %d diff %d reference-nonterminating %d handed-over %d kernel %s" % (b, mode, sc, len(pairs), nb, nf, ho, name.split(" ")[0]), flush=True)
    print("TOTAL pairs %d (checked kernel: %d) diff %d reference-nonterminating %d handed-over %d" % (tot, chk, bad, nonterm, handed))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
