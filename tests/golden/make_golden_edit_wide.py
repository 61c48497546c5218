"""Generate tests/golden/edit_wide.npz: striped_seqedit_pairwise (bsalign.h:1046) of the REAL reference (oracle/_ref) on
bands wider than 1024 columns -- overlap / extend mode and bandwidth 0 on long queries, at the borders of the device's
launch classes (64 / 128 / 256 band words), a moving band above 1024 columns, and ragged length ratios.
Run in the build container:  python tests/golden/make_golden_edit_wide.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import support as S  # noqa: E402


def main():
    assert S.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    rng = np.random.default_rng(20240615)
    cases = []
    for L in (1030, 1500, 4090, 4100, 6000, 8190, 8200, 12000, 16390):
        for mode in (0, 1, 2):
            T = rng.integers(0, 4, size=int(L * float(rng.choice([0.6, 1.0, 1.15])))).astype(np.uint8)
            Q = rng.integers(0, 4, size=L).astype(np.uint8)
            M = S.mutate(rng, T, float(rng.choice([0.03, 0.1, 0.2])))[:L]
            Q[:len(M)] = M
            cases.append((Q, T, mode, 0))
    T = rng.integers(0, 4, size=7000).astype(np.uint8)
    Q = S.mutate(rng, T, 0.12)
    cases.append((Q, T, 0, 2048))        # moving band above the register kernels
    cases.append((Q, T, 0, 1088))
    out = {}
    for k, (q, t, mode, bw) in enumerate(cases):
        res, cig, n = S.ref_edit(q, t, mode, bw)
        assert n >= 0
        out["q_%d" % k] = q
        out["t_%d" % k] = t
        out["meta_%d" % k] = np.array([mode, bw], dtype=np.int32)
        out["res_%d" % k] = res
        out["cig_%d" % k] = cig
    out["n"] = np.array([len(cases)])
    np.savez_compressed(os.path.join(HERE, "edit_wide.npz"), **out)
    print("edit_wide cases:", len(cases))


if __name__ == "__main__":
    main()
