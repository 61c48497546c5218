"""The reference-side binding of the MSA refinement's DP (include/bsalign_poa_diagdp.h + patches/bspoa_device_diagdp.diff,
built into oracle/_ref/libbsref_patched.so): the real end_bspoa with remsa_pedits filling its DP matrices through the
binding -- here with the oracle's orc_diagdp_fill behind it instead of the device -- must give the consensus, qualities,
alternative bases and MSA of the untouched run.  (The GPU test runs the same with bsa_diagdp_batch behind it.)"""
import ctypes as C
import os

import numpy as np
import pytest

import diag_support as D
import poa_support as P
import support as S

PATCHED = os.path.join(S.ROOT, "oracle", "_ref", "libbsref_patched.so")
pytestmark = pytest.mark.skipif(not os.path.exists(PATCHED), reason="oracle/_ref/libbsref_patched.so not built")

PROB = np.dtype([("seq0", np.uint64), ("seq1", np.uint64), ("mats0", np.uint64, (4,)), ("mats1", np.uint64, (4,)),
                 ("out0", np.uint64), ("out1", np.uint64), ("mlen", np.uint32), ("mbeg", np.uint32), ("mend", np.uint32), ("W", np.uint32)])
CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t)
CALLS = {"n": 0, "reads": 0}


@CB
def _oracle_backend(ctx, planes, planes_bytes, probs, n, matrix, matrix_bytes):
    orc = D._libs()
    pr = np.ctypeslib.as_array(C.cast(probs, C.POINTER(C.c_uint8)), shape=(n * PROB.itemsize,)).view(PROB)
    for p in pr:
        a0 = (C.c_void_p * 4)(*[planes + int(o) for o in p["mats0"]])
        a1 = (C.c_void_p * 4)(*[planes + int(o) for o in p["mats1"]])
        orc.orc_diagdp_fill(planes + int(p["seq0"]), planes + int(p["seq1"]), a0, a1, int(p["mlen"]), int(p["mbeg"]), int(p["mend"]), int(p["W"]),
                            matrix + int(p["out0"]), matrix + int(p["out1"]))
    CALLS["n"] += 1
    CALLS["reads"] += int(n)
    return 0


def load(backend_addr):
    L = C.CDLL(PATCHED)
    L.refp_create.restype = C.c_void_p
    L.refp_create.argtypes = [C.c_int] * 16
    L.refp_destroy.argtypes = [C.c_void_p]
    L.refp_push.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.refp_end.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.refp_cns_len.argtypes = [C.c_void_p]
    L.refp_cns_len.restype = C.c_uint32
    L.refp_cns.argtypes = [C.c_void_p] * 4
    L.refp_msa_hash.argtypes = [C.c_void_p] * 3
    L.refp_msa_hash.restype = C.c_uint64
    L.refp_attach_diagdp.argtypes = [C.c_void_p]
    L.refp_diagdp_stats.argtypes = [C.c_void_p] * 3
    L.refp_attach_diagdp(backend_addr)
    return L


def run(L, windows, p, how):
    hs = []
    for reads in windows:
        h = L.refp_create(*[int(p[k]) for k in P.PAR_ORDER])
        lens = np.array([len(x) for x in reads], dtype=np.uint32)
        offs = np.zeros(len(reads), dtype=np.uint64)
        offs[1:] = np.cumsum(lens)[:-1]
        blob = np.concatenate(reads).astype(np.uint8)
        L.refp_push(h, blob.ctypes.data, offs.ctypes.data, lens.ctypes.data, len(reads))
        hs.append(h)
    arr = (C.c_void_p * len(hs))(*hs)
    assert L.refp_end(arr, len(hs), how) == 0
    out = []
    for h in hs:
        n = L.refp_cns_len(h)
        cns, qlt, alt = (np.zeros(n, np.uint8) for _ in range(3))
        L.refp_cns(h, cns.ctypes.data, qlt.ctypes.data, alt.ctypes.data)
        nc, nr = C.c_uint32(), C.c_uint32()
        mh = L.refp_msa_hash(h, C.byref(nc), C.byref(nr))
        out.append((cns, qlt, alt, (mh, nc.value, nr.value)))
        L.refp_destroy(h)
    return out


def same(a, b):
    for w, (x, y) in enumerate(zip(a, b)):
        assert all(np.array_equal(x[k], y[k]) for k in range(3)) and x[3] == y[3], w


def test_end_bspoa_with_the_refinement_dp_through_the_binding():
    L = load(C.cast(_oracle_backend, C.c_void_p))
    rng = np.random.default_rng(11)
    windows = [P.synth_reads(500 + w, int(rng.integers(200, 900)), int(rng.integers(3, 14)), eps=(0.1,)) for w in range(10)]
    for p in (P.par(), P.par(realn=1), P.par(bandwidth=64, alnmode=0)):
        CALLS["n"] = CALLS["reads"] = 0
        ref = run(L, windows, p, 0)
        got = run(L, windows, p, 3)
        same(ref, got)
        assert CALLS["n"] >= len(windows) and CALLS["reads"] >= sum(len(w) for w in windows)      # the binding really ran, every read
