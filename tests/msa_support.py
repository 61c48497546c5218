"""Helpers for the MSA-format tests: a finished window of the REAL reference POA (oracle/_ref) with its MSA arrays and
the bytes its own writers produce (dump_binary_msa_bspoa bspoa.h:1555-1586, print_msa_bspoa :1491-1553)."""
import ctypes as C

import numpy as np

import poa_support as P
import support as S


def _ref():
    r = P.ref_poa()
    if getattr(r, "_msa_ready", False):
        return r
    vp = C.c_void_p
    r.ref_poa_msa_dims.argtypes = [vp] * 6
    r.ref_poa_msa_dims.restype = None
    r.ref_poa_msa_cols.argtypes = [vp] * 4
    r.ref_poa_msa_cols.restype = None
    r.ref_poa_msa_binary.argtypes = [vp, C.c_char_p, C.c_uint32, vp, C.c_uint64]
    r.ref_poa_msa_binary.restype = C.c_uint64
    r.ref_poa_msa_text.argtypes = [vp, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_uint64]
    r.ref_poa_msa_text.restype = C.c_uint64
    r.ref_msa_load_binary.argtypes = [vp, C.c_uint64, vp, vp, vp, C.c_uint64, vp, vp, vp, vp, vp, C.c_uint32, vp]
    r.ref_msa_load_binary.restype = C.c_int
    r._msa_ready = True
    return r


class RefWindow:
    """end_bspoa (untouched) on `reads`; keeps the reference's BSPOA alive until close()"""

    def __init__(self, reads, p=None):
        r = self.r = _ref()
        p = p or P.par()
        self.h = r.ref_poa_create(*[int(p[k]) for k in P.PAR_ORDER])
        lens = np.array([len(x) for x in reads], dtype=np.uint32)
        offs = np.zeros(len(reads), dtype=np.uint64)
        offs[1:] = np.cumsum(lens)[:-1]
        blob = np.concatenate(reads).astype(np.uint8)
        r.ref_poa_run(self.h, blob.ctypes.data, offs.ctypes.data, lens.ctypes.data, len(reads), 0, None, 0)
        mlen, mrow, nrds, nvar, cb = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64()
        r.ref_poa_msa_dims(self.h, C.byref(mlen), C.byref(mrow), C.byref(nrds), C.byref(cb), C.byref(nvar))
        self.mlen, self.mrow, self.nseq = mlen.value, mrow.value, nrds.value
        assert self.mrow == self.nseq + 3
        self.cols = np.zeros(cb.value, dtype=np.uint8)
        self.idxs = np.zeros(self.mlen, dtype=np.uint32)
        self.var = np.zeros(nvar.value, dtype=np.uint32)
        r.ref_poa_msa_cols(self.h, self.cols.ctypes.data, self.idxs.ctypes.data, self.var.ctypes.data if nvar.value else None)
        n = r.ref_poa_cns_len(self.h)
        self.cns, self.qlt, self.alt = (np.zeros(n, np.uint8) for _ in range(3))
        r.ref_poa_cns(self.h, self.cns.ctypes.data, self.qlt.ctypes.data, self.alt.ctypes.data)

    def binary(self, meta=b""):
        n = self.r.ref_poa_msa_binary(self.h, meta or None, len(meta), None, 0)
        out = np.zeros(n, dtype=np.uint8)
        self.r.ref_poa_msa_binary(self.h, meta or None, len(meta), out.ctypes.data, n)
        return out.tobytes()

    def text(self, label, mbeg=0, mend=0, linewidth=0):
        n = self.r.ref_poa_msa_text(self.h, label.encode(), mbeg, mend, linewidth, None, 0)
        out = np.zeros(max(n, 1), dtype=np.uint8)
        self.r.ref_poa_msa_text(self.h, label.encode(), mbeg, mend, linewidth, out.ctypes.data, n)
        return out[:n].tobytes()

    def close(self):
        if self.h:
            self.r.ref_poa_destroy(self.h)
            self.h = None


def ref_load_binary(blob):
    """the reference's load_binary_msa_bspoa on a container -> (rc, nseq, mlen, cols, cns, qlt, alt, meta)"""
    r = _ref()
    buf = np.frombuffer(blob, dtype=np.uint8).copy()
    nseq, mlen, clen, ml = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    cols = np.zeros(len(blob) * 2 + 64, dtype=np.uint8)
    cns, qlt, alt = (np.zeros(len(blob) + 1, dtype=np.uint8) for _ in range(3))
    meta = np.zeros(len(blob) + 1, dtype=np.uint8)
    rc = r.ref_msa_load_binary(buf.ctypes.data, buf.size, C.byref(nseq), C.byref(mlen), cols.ctypes.data, cols.size,
                               cns.ctypes.data, qlt.ctypes.data, alt.ctypes.data, C.byref(clen), meta.ctypes.data, meta.size, C.byref(ml))
    n = clen.value
    return rc, nseq.value, mlen.value, cols[:mlen.value * (nseq.value + 3)].reshape(mlen.value, nseq.value + 3), cns[:n], qlt[:n], alt[:n], meta[:ml.value].tobytes()


CASES = [  # (seed, read length, reads, eps choices)
    (11, 300, 6, (0.05, 0.1)),
    (12, 1500, 12, (0.05, 0.1, 0.15)),
    (13, 90, 3, (0.0, 0.02)),
    (14, 800, 25, (0.1, 0.2)),
]
TEXT_ARGS = [(0, 0, 0), (0, 0, 100), (0, 0, 60), (37, 0, 80), (10, 215, 50), (5, 64, 0)]     # (mbeg, mend, linewidth)
