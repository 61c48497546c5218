"""GPU: a C program written against the reference's function names / types (include/bsalign_compat.h) gives the
oracle's results -- the drop-in boundary for single-pair callers."""
import os
import subprocess

import numpy as np
import pytest

import support as S

pytestmark = pytest.mark.gpu


def test_c_caller_with_reference_names(tmp_path):
    root = S.ROOT
    exe = str(tmp_path / "compat_main")
    subprocess.run(["gcc", "-O1", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c", "compat_main.c"), "-o", exe,
                    "-L", os.path.join(root, "bsalign_amd"), "-lbsalign_compat", "-lbsalign_hip",
                    "-Wl,-rpath," + os.path.join(root, "bsalign_amd")], check=True)
    rng = np.random.default_rng(4)
    T = rng.integers(0, 4, size=300).astype(np.uint8)
    Q = S.mutate(rng, T, 0.1)
    s = lambda a: "".join("ACGT"[int(c)] for c in a)
    for mode in (0, 1):
        out = subprocess.run([exe, str(mode), "64", s(Q), s(T)], check=True, capture_output=True, text=True).stdout.strip().split("\n")
        got = [int(x) for x in out[0].split()[1:]]
        res, cig, _ = S.oracle_align(Q, T, mode, 64, 2, -6, -3, -2, 0, 0)
        assert got[:10] == res.tolist() and got[10:] == cig.tolist()
        assert len(out[1]) == res[9] == len(out[2]) == len(out[3])           # alignment strings have rs.aln columns
        assert out[1].replace("-", "") == s(Q[res[1]:res[2]]) and out[3].replace("-", "") == s(T[res[3]:res[4]])
        egot = [int(x) for x in out[4].split()[1:]]
        eres, ecig, _ = S.oracle_edit(Q, T, mode, 0)
        assert egot[:10] == eres.tolist() and egot[10:] == ecig.tolist()
