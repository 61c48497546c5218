#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ from the REAL reference (oracle/_ref/libbsref.so, built by
oracle/Makefile from /root/reference).  Runs only in the build container; the fixtures (inputs + expected
outputs, no reference text) are committed and travel to the GPU box.

    python tests/golden/make_golden.py

align8.npz : cases of banded_striped_epi8_seqalign_pairwise (bsalign.h:3854)
edit.npz   : cases of striped_seqedit_pairwise               (bsalign.h:1046)
Each case k: q_k, t_k (uint8 codes), meta_k = [mode, bw, M, X, O, E, Q, P], res_k (seqalign_result_t as 10 int32),
cig_k (uint32 CIGAR words).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import support as S  # noqa: E402

SCORINGS = [(2, -6, -3, -2, 0, 0), (2, -2, -4, -2, 0, 0), (2, -6, 0, -3, 0, 0), (2, -6, -3, -2, -8, -1)]


def mk(rng, L, eps, ratio):
    T = rng.integers(0, 4, size=L).astype(np.uint8)
    Q = S.mutate(rng, T, eps)
    if ratio != 1.0:
        Lq = max(1, int(len(Q) * ratio))
        Q = Q[:Lq] if Lq <= len(Q) else np.concatenate([Q, rng.integers(0, 4, size=Lq - len(Q)).astype(np.uint8)])
    if len(Q) == 0:
        Q = np.array([int(rng.integers(4))], dtype=np.uint8)
    return Q, T


def gen_align(rng):
    cases = []
    lens = [1, 15, 16, 17, 63, 64, 65, 100, 1000]
    for L in lens:
        for bw in (0, 16, 64, 128, 256):
            for mode in (0, 1, 2):
                for si, sc in enumerate(SCORINGS):
                    eps = [0.20, 0.10, 0.01][(L + bw + mode + si) % 3]
                    ratio = [1.0, 0.9, 1.1, 2.0][(L + si + mode) % 4]
                    if L >= 1000 and (si + mode + bw // 16) % 3:
                        continue   # keep the fixture small
                    q, t = mk(rng, L, eps, ratio)
                    cases.append((q, t, mode, bw, sc))
    # degenerate shapes
    same = np.zeros(200, dtype=np.uint8)
    cases.append((same, same.copy(), 0, 64, SCORINGS[0]))
    cases.append((same[:50], same.copy(), 0, 16, SCORINGS[0]))                # qlen < tlen, narrow band
    q, t = mk(rng, 300, 0.1, 1.0)
    cases.append((q[:40], t, 0, 64, SCORINGS[0]))                              # qlen < bandwidth
    cases.append((q, t[:60], 0, 32, SCORINGS[0]))                              # tlen << qlen: rush-to-end branch, band jumps
    cases.append((q, t[:60], 1, 32, SCORINGS[0]))
    cases.append((q[:10], t[:10], 0, 128, SCORINGS[3]))                        # bandwidth > qlen
    # benchmark shapes: the synthetic generator's pairs
    for k, L in ((0, 10000), (1, 10000), (2, 15000)):
        q, t = S.synth_pair(k, L)
        cases.append((q, t, 0, 128, SCORINGS[0]))
    q, t = S.synth_pair(3, 10000)
    cases.append((q, t, 0, 128, SCORINGS[1]))
    cases.append((q, t, 1, 64, SCORINGS[3]))
    out = {}
    kept = 0
    for q, t, mode, bw, sc in cases:
        # the oracle flags inputs on which the reference does not terminate; those cannot be goldens
        o = S.oracle_align(q, t, mode, bw, *sc)
        if o[2] == S.ORC_ERR_TRACE:
            # kept in the fixture as dropped_* so that the filter can be audited (tests/test_oracle.py checks that the
            # oracle still flags every one of them; the reference itself cannot be asked: it would not return)
            nd = sum(1 for k in out if k.startswith("dropped_q_"))
            out["dropped_q_%d" % nd] = q
            out["dropped_t_%d" % nd] = t
            out["dropped_meta_%d" % nd] = np.array([mode, bw] + list(sc), dtype=np.int32)
            continue
        res, cig, n = S.ref_align(q, t, mode, bw, *sc)
        out["q_%d" % kept] = q
        out["t_%d" % kept] = t
        out["meta_%d" % kept] = np.array([mode, bw] + list(sc), dtype=np.int32)
        out["res_%d" % kept] = res
        out["cig_%d" % kept] = cig
        kept += 1
    out["n"] = np.array([kept])
    out["ndropped"] = np.array([sum(1 for k in out if k.startswith("dropped_q_"))])
    return out


def gen_edit(rng):
    out = {}
    kept = 0
    cases = []
    for L in [1, 2, 63, 64, 65, 129, 1000, 3000]:
        for bw in (0, 64, 256):
            for mode in (0, 1, 2):
                eps = [0.20, 0.10, 0.01][(L + bw + mode) % 3]
                ratio = [1.0, 0.9, 1.1][(L + mode) % 3]
                cases.append(mk(rng, L, eps, ratio) + (mode, bw))
    q, t = S.synth_pair(0, 100000)
    cases.append((q, t, 0, 256))
    q, t = S.synth_pair(1, 20000)
    cases.append((q, t, 0, 64))      # band narrower than the indel drift: band-limited result
    cases.append((q, t, 1, 0))
    for q, t, mode, bw in cases:
        res, cig, n = S.ref_edit(q, t, mode, bw)
        out["q_%d" % kept] = q
        out["t_%d" % kept] = t
        out["meta_%d" % kept] = np.array([mode, bw], dtype=np.int32)
        out["res_%d" % kept] = res
        out["cig_%d" % kept] = cig
        kept += 1
    out["n"] = np.array([kept])
    return out


def main():
    assert S.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    rng = np.random.default_rng(20240611)
    a = gen_align(rng)
    np.savez_compressed(os.path.join(HERE, "align8.npz"), **a)
    e = gen_edit(rng)
    np.savez_compressed(os.path.join(HERE, "edit.npz"), **e)
    print("align8 cases:", int(a["n"][0]), "edit cases:", int(e["n"][0]))


if __name__ == "__main__":
    main()
