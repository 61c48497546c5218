"""ctypes mirror of include/bsalign_msa.h (the POA's binary / text MSA formats on plain arrays; host-only code of
libbsalign_hip.so).  `cols` is the reference's msacols (columns of nseq + 3 bytes), `idxs` its msaidxs."""
import ctypes as C

import numpy as np

from . import LIB_PATH, BsaError

_lib = None
E_NOMEM = -3


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.bsa_msa_binary_write.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
        L.bsa_msa_binary_read.argtypes = [vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                          vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_uint32)]
        L.bsa_msa_consensus.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp, C.POINTER(C.c_uint32), vp, vp]
        L.bsa_msa_text.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp, C.c_uint32, C.c_char_p,
                                   C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _two_pass(call):
    """size query, then the real call"""
    need = C.c_size_t(0)
    rc = call(None, 0, need)
    if rc not in (0, E_NOMEM):
        raise BsaError(rc)
    out = np.zeros(max(need.value, 1), dtype=np.uint8)
    rc = call(_p(out), out.size, need)
    if rc != 0:
        raise BsaError(rc)
    return out[:need.value].tobytes()


def binary_write(cols, idxs, nseq, mlen, meta=b""):
    cols = np.ascontiguousarray(cols, dtype=np.uint8)
    idxs = None if idxs is None else np.ascontiguousarray(idxs, dtype=np.uint32)
    return _two_pass(lambda o, cap, need: lib().bsa_msa_binary_write(_p(cols), _p(idxs), nseq, mlen, meta or None, len(meta), o, cap, C.byref(need)))


def binary_read(blob):
    """-> (nseq, mlen, cols[mlen, nseq + 3], meta bytes, consumed)"""
    buf = np.frombuffer(blob, dtype=np.uint8)
    used, nseq, mlen, ml = C.c_size_t(0), C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    rc = lib().bsa_msa_binary_read(_p(buf), buf.size, C.byref(used), C.byref(nseq), C.byref(mlen), None, 0, None, 0, C.byref(ml))
    if rc != 0:
        raise BsaError(rc)
    cols = np.zeros((mlen.value, nseq.value + 3), dtype=np.uint8)
    meta = np.zeros(max(ml.value, 1), dtype=np.uint8)
    rc = lib().bsa_msa_binary_read(_p(buf), buf.size, C.byref(used), C.byref(nseq), C.byref(mlen), _p(cols), cols.size, _p(meta), meta.size, C.byref(ml))
    if rc != 0:
        raise BsaError(rc)
    return nseq.value, mlen.value, cols, meta[:ml.value].tobytes(), used.value


def consensus(cols, idxs, nseq, mlen, reads=False):
    cols = np.ascontiguousarray(cols, dtype=np.uint8)
    idxs = None if idxs is None else np.ascontiguousarray(idxs, dtype=np.uint32)
    cns, qlt, alt = (np.zeros(max(mlen, 1), dtype=np.uint8) for _ in range(3))
    clen = C.c_uint32(0)
    rdseqs = np.zeros(max(mlen * nseq, 1), dtype=np.uint8) if reads else None
    rdoffs = np.zeros(nseq + 1, dtype=np.uint64) if reads else None
    rc = lib().bsa_msa_consensus(_p(cols), _p(idxs), nseq, mlen, _p(cns), _p(qlt), _p(alt), C.byref(clen), _p(rdseqs), _p(rdoffs))
    if rc != 0:
        raise BsaError(rc)
    n = clen.value
    if reads:
        return cns[:n], qlt[:n], alt[:n], rdseqs[:int(rdoffs[nseq])], rdoffs
    return cns[:n], qlt[:n], alt[:n]


def text(cols, idxs, nseq, mlen, cns, qlt, alt, label, mbeg=0, mend=0, linewidth=0, var_mpos=None):
    cols = np.ascontiguousarray(cols, dtype=np.uint8)
    idxs = None if idxs is None else np.ascontiguousarray(idxs, dtype=np.uint32)
    cns, qlt, alt = (np.ascontiguousarray(a, dtype=np.uint8) for a in (cns, qlt, alt))
    var = None if var_mpos is None or len(var_mpos) == 0 else np.ascontiguousarray(var_mpos, dtype=np.uint32)
    nvar = 0 if var is None else var.size
    return _two_pass(lambda o, cap, need: lib().bsa_msa_text(_p(cols), _p(idxs), nseq, mlen, _p(cns), _p(qlt), _p(alt), _p(var), nvar,
                                                            label.encode(), mbeg, mend, linewidth, o, cap, C.byref(need)))


CNS_WINDOW_DTYPE = np.dtype([("cols_off", np.uint64), ("idxs_off", np.uint64), ("out_off", np.uint64),
                             ("nall", np.uint32), ("nseq", np.uint32), ("nmax", np.uint32), ("mlen", np.uint32)])


def call_consensus(cols, idxs, nall, nseq, nmax, mlen, par7):
    """host form (bsa_msa_call_consensus = cns_bspoa): -> (cns, qlt, alt, score); the three consensus bytes of `cols` are overwritten"""
    L = lib()
    L.bsa_msa_call_consensus.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    cns, qlt, alt = (np.zeros(mlen + 1, np.uint8) for _ in range(3))
    clen, score = C.c_uint32(), C.c_double()
    par7 = np.ascontiguousarray(par7, np.float32)
    rc = L.bsa_msa_call_consensus(_p(cols), _p(idxs), nall, nseq, nmax, mlen, _p(par7), _p(cns), _p(qlt), _p(alt), C.byref(clen), C.byref(score))
    if rc != 0:
        raise BsaError(rc)
    return cns[:clen.value], qlt[:clen.value], alt[:clen.value], score.value


def call_consensus_batch(ctx, windows, par7):
    """device form (bsa_msa_call_consensus_batch): windows = [(cols uint8 [ncolumns * (nall + 3)], idxs uint32 [mlen] or None, nall, nseq, nmax, mlen)];
    -> [(cns, qlt, alt, score, cols with the three consensus bytes of every column written)]"""
    L = lib()
    L.bsa_msa_call_consensus_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    n = len(windows)
    win = np.zeros(n, CNS_WINDOW_DTYPE)
    cparts, iparts = [], []
    cacc = iacc = oacc = 0
    for k, (cols, idxs, nall, nseq, nmax, mlen) in enumerate(windows):
        cols = np.ascontiguousarray(cols, np.uint8).reshape(-1)
        win[k] = (cacc, np.uint64(0xFFFFFFFFFFFFFFFF) if idxs is None else iacc, oacc, nall, nseq, nmax, mlen)
        cparts.append(cols); cacc += cols.size
        if idxs is not None:
            iparts.append(np.ascontiguousarray(idxs, np.uint32)); iacc += len(idxs)
        oacc += mlen
    blob = np.concatenate(cparts) if cparts else np.zeros(0, np.uint8)
    iblob = np.concatenate(iparts) if iparts else np.zeros(0, np.uint32)
    cns, qlt, alt = (np.zeros(max(oacc, 1), np.uint8) for _ in range(3))
    clen = np.zeros(max(n, 1), np.uint32); score = np.zeros(max(n, 1), np.float64)
    par7 = np.ascontiguousarray(par7, np.float32)
    rc = L.bsa_msa_call_consensus_batch(ctx.h, _p(blob), blob.size, _p(iblob) if iblob.size else None, iblob.size, _p(win), n, _p(par7),
                                        _p(cns), _p(qlt), _p(alt), oacc, _p(clen), _p(score))
    if rc != 0:
        raise BsaError(rc)
    out = []
    for k, (cols, idxs, nall, nseq, nmax, mlen) in enumerate(windows):
        o = int(win[k]["out_off"]); m = int(clen[k]); c0 = int(win[k]["cols_off"])
        out.append((cns[o:o + m].copy(), qlt[o:o + m].copy(), alt[o:o + m].copy(), float(score[k]), blob[c0:c0 + np.asarray(cols).size].copy()))
    return out
