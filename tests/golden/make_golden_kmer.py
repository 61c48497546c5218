"""Generate tests/golden/kmer_edit.npz: inputs and the REAL reference's kmer_striped_seqedit_pairwise results
(oracle/_ref/libbsref.so, built from /root/reference by oracle/Makefile).  Run in the build container:
    python tests/golden/make_golden_kmer.py
tests/test_kmer_cpu.py checks the host chaining/stitching (with the oracle's edit DP between the anchors) against it,
tests/test_kmer_gpu.py the device path."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import support as S  # noqa: E402
import kmer_support as K  # noqa: E402


def make_cases(rng):
    cases = []
    for it in range(48):
        L = int(rng.integers(60, 5000)) if it < 40 else int(rng.integers(8000, 12000))
        T = rng.integers(0, 4, L).astype(np.uint8)
        eps = float(rng.choice([0.0, 0.02, 0.05, 0.1, 0.15, 0.3]))
        Q = S.mutate(rng, T, eps)
        kind = it % 8
        if kind == 1 and len(Q) > 300:      # long deletion
            a = int(rng.integers(50, len(Q) - 200))
            Q = np.concatenate([Q[:a], Q[a + int(rng.integers(60, 150)):]])
        elif kind == 2 and len(Q) > 300:    # long insertion
            a = int(rng.integers(50, len(Q) - 100))
            Q = np.concatenate([Q[:a], rng.integers(0, 4, int(rng.integers(60, 400))).astype(np.uint8), Q[a:]])
        elif kind == 3 and len(Q) > 600:    # a block moved: anchors off the main diagonal
            a = int(rng.integers(50, len(Q) // 2))
            b = a + int(rng.integers(50, 200))
            c = int(rng.integers(b, len(Q) - 10))
            Q = np.concatenate([Q[:a], Q[b:c], Q[a:b], Q[c:]])
        elif kind == 4:                     # tandem repeat
            unit = rng.integers(0, 4, int(rng.integers(2, 30))).astype(np.uint8)
            a = int(rng.integers(0, len(T)))
            T = np.concatenate([T[:a], np.tile(unit, int(rng.integers(3, 30))), T[a:]])
            Q = S.mutate(rng, T, eps)
        elif kind == 5:                     # ragged ends
            Q = Q[int(rng.integers(0, min(50, len(Q) // 2 + 1))):]
            T = T[:len(T) - int(rng.integers(0, min(50, len(T) // 2 + 1)))]
        elif kind == 6 and it % 16 == 6:    # no shared k-mers at all: whole-pair global alignment
            Q = rng.integers(0, 4, len(Q) + 1).astype(np.uint8)
        elif kind == 7 and it % 16 == 7:    # homopolymer query
            Q = np.zeros(max(len(Q), 1), dtype=np.uint8)
        if len(Q) == 0:
            Q = T[:1].copy()
        ksz = int(rng.choice([3, 5, 8, 11, 13, 15, 16]))
        cases.append((ksz, Q, T))
    return cases


def main():
    rng = np.random.default_rng(20240613)
    cases = make_cases(rng)
    seqs, meta, res, cigs, coff, anchors = [], [], [], [], [0], []
    off = 0
    for ksz, Q, T in cases:
        r, c = K.ref_kmer_edit(ksz, Q, T)
        meta.append((ksz, off, len(Q), off + len(Q), len(T)))
        seqs += [Q, T]
        off += len(Q) + len(T)
        res.append(r)
        cigs.append(c)
        coff.append(coff[-1] + len(c))
        anchors.append(len(K.kmer_chain(ksz, Q, T)))
    np.savez_compressed(os.path.join(HERE, "kmer_edit.npz"), seqs=np.concatenate(seqs), meta=np.array(meta, dtype=np.int64),
                        res=np.array(res, dtype=np.int32), cigar=np.concatenate(cigs).astype(np.uint32), cigar_off=np.array(coff, dtype=np.int64))
    print(len(cases), "pairs,", sum(1 for a in anchors if a == 0), "without a chain,", off, "bases")


if __name__ == "__main__":
    main()
