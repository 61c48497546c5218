"""The reference-side binding's window-count policy without a GPU (include/bsalign_poa_batch.h: below BSA_POA_MIN_WINDOWS windows in flight
bsa_poa_end_many / bsa_poa_end_one run the reference's own end_bspoa on host threads, no device attached): the patched header gives the untouched
end_bspoa's consensus, qualities and MSA for a handful of windows, with no device library attached at all."""
import ctypes as C
import os

import numpy as np
import pytest

import poa_support as P
import support as S

PATCHED = os.path.join(S.ROOT, "oracle", "_ref", "libbsref_patched.so")
pytestmark = pytest.mark.skipif(not os.path.exists(PATCHED), reason="oracle/_ref/libbsref_patched.so not built")


def _run(L, windows, p, how):
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


def test_few_windows_run_the_reference_on_host_threads(monkeypatch):
    monkeypatch.delenv("BSA_POA_MIN_WINDOWS", raising=False)
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
    p = P.par()
    rng = np.random.default_rng(11)
    windows = [P.synth_reads(1300 + w, int(rng.integers(300, 900)), int(rng.integers(4, 10)), eps=(0.1,)) for w in range(9)]
    ref = _run(L, windows, p, 0)
    many = _run(L, windows, p, 2)          # bsa_poa_end_many: 9 windows < 64 -> host threads, nothing of libbsalign_hip is called (none is attached)
    one = _run(L, windows[:2], p, 1)       # bsa_poa_end_one
    for a, b in list(zip(ref, many)) + list(zip(ref, one)):
        assert all(np.array_equal(a[k], b[k]) for k in range(3)) and a[3] == b[3]
