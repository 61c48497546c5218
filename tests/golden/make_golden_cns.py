"""Generates tests/golden/cns.npz (build container, needs oracle/_ref): finished windows of the real reference POA -- the MSA columns
with the three consensus bytes blanked, the inputs of cns_bspoa besides them, and what the REAL cns_bspoa (bspoa.h:3457-3733)
produced: consensus, both quality strings, the consensus bytes of every column, its return value."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("BSA_NO_TORCH_PRELOAD", "1")
import msa_support as MS
import poa_support as P

out = {}
CASES = [(11, 300, 6, (0.05, 0.1)), (14, 500, 25, (0.1, 0.2)), (15, 300, 70, (0.1,))]
for k, (seed, L, n, eps) in enumerate(CASES):
    w = MS.RefWindow(P.synth_reads(seed, L, n, eps=eps))
    r = w.r
    r.ref_poa_cns_call.argtypes = [C.c_void_p]; r.ref_poa_cns_call.restype = C.c_double
    r.ref_poa_cns_inputs.argtypes = [C.c_void_p] * 5; r.ref_poa_cns_inputs.restype = None
    nmsa, nrds, nall = C.c_uint32(), C.c_uint32(), C.c_uint32()
    par7 = np.zeros(7, np.float32)
    r.ref_poa_cns_inputs(w.h, C.byref(nmsa), C.byref(nrds), C.byref(nall), par7.ctypes.data)
    score = r.ref_poa_cns_call(w.h)
    cols = w.cols.reshape(-1, w.mrow)[w.idxs.astype(np.int64)].copy()           # columns in MSA order
    out["cols_%d" % k] = cols
    out["dims_%d" % k] = np.array([nmsa.value, nrds.value, nall.value, w.mlen], np.uint32)
    out["par_%d" % k] = par7
    out["cns_%d" % k], out["qlt_%d" % k], out["alt_%d" % k] = w.cns, w.qlt, w.alt
    out["score_%d" % k] = np.array([score], np.float64)
    w.close()
out["n"] = np.array([len(CASES)])
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cns.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path))
