"""Helpers for the k-mer anchored edit alignment tests (test code only).

kmer_host() drives the product's HOST pieces (bsa_kmer_chain / bsa_kmer_segments / bsa_kmer_assemble, no GPU needed)
with a caller-supplied segment aligner, so the CPU suite can check chaining and stitching with the oracle's edit DP as
the segment aligner, and the GPU suite checks bsa_kmer_edit_batch as a whole.
"""
import ctypes as C
import os

import numpy as np

from support import ROOT, ptr, u8p, u32p, u64p, i32p, ref, MODE_GLOBAL, MODE_EXTEND

SEG_DTYPE = np.dtype([("qb", "<u4"), ("qe", "<u4"), ("tb", "<u4"), ("te", "<u4"), ("mode", "<u4"), ("ml", "<u4")])
SEG_REVERSED = 0x100

_LIB = None


def hostlib():
    global _LIB
    if _LIB is None:
        lib = C.CDLL(os.path.join(ROOT, "bsalign_amd", "libbsalign_hip.so"))
        lib.bsa_kmer_chain.restype = C.c_uint32
        lib.bsa_kmer_chain.argtypes = [C.c_uint32, u8p, C.c_uint32, u8p, C.c_uint32, u64p, C.c_uint32]
        lib.bsa_kmer_segments.restype = C.c_uint32
        lib.bsa_kmer_segments.argtypes = [C.c_uint32, u64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        lib.bsa_kmer_assemble.restype = C.c_int
        lib.bsa_kmer_assemble.argtypes = [C.c_void_p, C.c_uint32, i32p, u32p, u64p, i32p, u32p, C.c_uint64, u64p]
        _LIB = lib
    return _LIB


def kmer_chain(ksz, q, t):
    lib = hostlib()
    q = np.ascontiguousarray(q, dtype=np.uint8)
    t = np.ascontiguousarray(t, dtype=np.uint8)
    cap = max(min(len(q), len(t)), 1)
    maps = np.zeros(cap, dtype=np.uint64)
    n = lib.bsa_kmer_chain(ksz, ptr(q, u8p), len(q), ptr(t, u8p), len(t), ptr(maps, u64p), cap)
    assert n != 0xFFFFFFFF
    return maps[:n].copy()


def kmer_segments(ksz, maps, qlen, tlen):
    lib = hostlib()
    segs = np.zeros(len(maps) + 1, dtype=SEG_DTYPE)
    m = np.ascontiguousarray(maps, dtype=np.uint64) if len(maps) else np.zeros(1, dtype=np.uint64)
    n = lib.bsa_kmer_segments(ksz, ptr(m, u64p), len(maps), qlen, tlen, segs.ctypes.data)
    return segs[:n].copy()


def kmer_host(ksz, q, t, edit):
    """edit(qseg, tseg, mode) -> (res[10] int32, cigar words): the whole alignment through the host pieces"""
    lib = hostlib()
    q = np.ascontiguousarray(q, dtype=np.uint8)
    t = np.ascontiguousarray(t, dtype=np.uint8)
    maps = kmer_chain(ksz, q, t)
    segs = kmer_segments(ksz, maps, len(q), len(t))
    rs = np.zeros((len(segs), 10), dtype=np.int32)
    cigs, off = [], [0]
    for k, s in enumerate(segs):
        if s["mode"] & SEG_REVERSED:
            qs, ts = q[:s["qe"]][::-1], t[:s["te"]][::-1]
        else:
            qs, ts = q[s["qb"]:s["qe"]], t[s["tb"]:s["te"]]
        if len(qs) and len(ts):
            r, c = edit(np.ascontiguousarray(qs), np.ascontiguousarray(ts), int(s["mode"]) & 3)
            rs[k] = r
            cigs.append(np.asarray(c, dtype=np.uint32))
        else:
            cigs.append(np.zeros(0, dtype=np.uint32))
        off.append(off[-1] + len(cigs[-1]))
    seg_cig = np.concatenate(cigs + [np.zeros(1, dtype=np.uint32)])
    seg_off = np.array(off, dtype=np.uint64)
    cap = int(off[-1]) + len(segs) + 8
    out = np.zeros(10, dtype=np.int32)
    cig = np.zeros(cap, dtype=np.uint32)
    n = np.zeros(1, dtype=np.uint64)
    rc = lib.bsa_kmer_assemble(segs.ctypes.data, len(segs), ptr(rs, i32p), ptr(seg_cig, u32p), ptr(seg_off, u64p),
                               ptr(out, i32p), ptr(cig, u32p), cap, ptr(n, u64p))
    assert rc == 0
    return out, cig[:int(n[0])].copy(), maps


def ref_kmer_edit(ksz, q, t):
    lib = ref()
    if not hasattr(lib, "_kmer_bound"):
        lib.ref_kmer_edit_pairwise.restype = C.c_long
        lib.ref_kmer_edit_pairwise.argtypes = [C.c_void_p, C.c_int, u8p, C.c_uint32, u8p, C.c_uint32, i32p, u32p, C.c_long]
        lib._kmer_bound = True
    q = np.ascontiguousarray(q, dtype=np.uint8)
    t = np.ascontiguousarray(t, dtype=np.uint8)
    res = np.zeros(10, dtype=np.int32)
    cap = len(q) + len(t) + 8
    cig = np.zeros(cap, dtype=np.uint32)
    n = lib.ref_kmer_edit_pairwise(lib._ctx, ksz, ptr(q, u8p), len(q), ptr(t, u8p), len(t), ptr(res, i32p), ptr(cig, u32p), cap)
    assert n >= 0
    return res, cig[:n].copy()
