"""Shared helpers for the test-suite (ctypes bindings for the checkers + synthetic inputs).

Only test code imports this.  It binds
  * oracle/liboracle.so        -- own scalar C restatement (the oracle)
  * oracle/_ref/libbsref.so    -- the real reference, when it was built in this container
and provides the numpy form of the synthetic read-pair generator whose C and HIP
forms live in the product library.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

MODE_GLOBAL, MODE_OVERLAP, MODE_EXTEND = 0, 1, 2
ERR_INPUT = -(1 << 40)
ORC_ERR_TRACE = -(1 << 41)

u8p = C.POINTER(C.c_uint8)
i8p = C.POINTER(C.c_int8)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)


def ptr(a, t):
    return a.ctypes.data_as(t)


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "all"], check=True, stdout=subprocess.DEVNULL)


_ORC = None
_REF = None


def oracle():
    global _ORC
    if _ORC is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build_oracle()
        lib = C.CDLL(path)
        lib.orc_align_pairwise.restype = C.c_long
        lib.orc_align_pairwise.argtypes = [u8p, C.c_uint32, u8p, C.c_uint32, C.c_int, C.c_uint32, i8p,
                                           C.c_int, C.c_int, C.c_int, C.c_int, i32p, u32p, C.c_long]
        lib.orc_align_pairwise_trace.restype = C.c_long
        lib.orc_align_pairwise_trace.argtypes = lib.orc_align_pairwise.argtypes + [i32p]
        lib.orc_edit_pairwise.restype = C.c_long
        lib.orc_edit_pairwise.argtypes = [u8p, C.c_uint32, u8p, C.c_uint32, C.c_int, C.c_uint32, i32p, u32p, C.c_long]
        lib.orc_align_batch_time.restype = C.c_double
        lib.orc_align_batch_time.argtypes = [u8p, u64p, u32p, u64p, u32p, C.c_long, C.c_int, C.c_uint32, i8p,
                                             C.c_int, C.c_int, C.c_int, C.c_int, i64p]
        lib.orc_edit_batch_time.restype = C.c_double
        lib.orc_edit_batch_time.argtypes = [u8p, u64p, u32p, u64p, u32p, C.c_long, C.c_int, C.c_uint32, i64p]
        lib.orc_get_piecewise.restype = C.c_int
        lib.orc_get_piecewise.argtypes = [C.c_int] * 5
        _ORC = lib
    return _ORC


def have_ref():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libbsref.so"))


def ref():
    global _REF
    if _REF is None:
        lib = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libbsref.so"))
        lib.ref_ctx_create.restype = C.c_void_p
        lib.ref_ctx_destroy.argtypes = [C.c_void_p]
        lib.ref_align_pairwise.restype = C.c_long
        lib.ref_align_pairwise.argtypes = [C.c_void_p, u8p, C.c_uint32, u8p, C.c_uint32, C.c_int, C.c_uint32,
                                           C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i32p, u32p, C.c_long]
        lib.ref_align_pairwise_mtx.restype = C.c_long
        lib.ref_align_pairwise_mtx.argtypes = [C.c_void_p, u8p, C.c_uint32, u8p, C.c_uint32, C.c_int, C.c_uint32,
                                               i8p, C.c_int, C.c_int, C.c_int, C.c_int, i32p, u32p, C.c_long]
        lib.ref_edit_pairwise.restype = C.c_long
        lib.ref_edit_pairwise.argtypes = [C.c_void_p, u8p, C.c_uint32, u8p, C.c_uint32, C.c_int, C.c_uint32,
                                          i32p, u32p, C.c_long]
        lib.ref_align_batch_time.restype = C.c_double
        lib.ref_align_batch_time.argtypes = [C.c_void_p, u8p, u64p, u32p, u64p, u32p, C.c_long, C.c_int, C.c_uint32,
                                             C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i64p]
        lib.ref_edit_batch_time.restype = C.c_double
        lib.ref_edit_batch_time.argtypes = [C.c_void_p, u8p, u64p, u32p, u64p, u32p, C.c_long, C.c_int, C.c_uint32, i64p]
        lib._ctx = lib.ref_ctx_create()
        _REF = lib
    return _REF


def score_matrix(mat, mis):
    m = np.empty(16, dtype=np.int8)
    for i in range(16):
        m[i] = mat if (i >> 2) == (i & 3) else mis
    return m


def oracle_align(q, t, mode, bw, M, X, O, E, Q, P, want_begs=False, mtx=None):
    q = np.ascontiguousarray(q, dtype=np.uint8)
    t = np.ascontiguousarray(t, dtype=np.uint8)
    m = score_matrix(M, X) if mtx is None else np.ascontiguousarray(mtx, dtype=np.int8)
    res = np.zeros(10, dtype=np.int32)
    cap = len(q) + len(t) + 8
    cig = np.zeros(cap, dtype=np.uint32)
    lib = oracle()
    if want_begs:
        begs = np.zeros(max(len(t), 1), dtype=np.int32)
        n = lib.orc_align_pairwise_trace(ptr(q, u8p), len(q), ptr(t, u8p), len(t), mode, bw, ptr(m, i8p), O, E, Q, P,
                                         ptr(res, i32p), ptr(cig, u32p), cap, ptr(begs, i32p))
        return res, cig[:max(n, 0)].copy(), n, begs
    n = lib.orc_align_pairwise(ptr(q, u8p), len(q), ptr(t, u8p), len(t), mode, bw, ptr(m, i8p), O, E, Q, P,
                               ptr(res, i32p), ptr(cig, u32p), cap)
    return res, cig[:max(n, 0)].copy(), n


def ref_align(q, t, mode, bw, M, X, O, E, Q, P, mtx=None):
    q = np.ascontiguousarray(q, dtype=np.uint8)
    t = np.ascontiguousarray(t, dtype=np.uint8)
    res = np.zeros(10, dtype=np.int32)
    cap = len(q) + len(t) + 8
    cig = np.zeros(cap, dtype=np.uint32)
    lib = ref()
    if mtx is None:
        n = lib.ref_align_pairwise(lib._ctx, ptr(q, u8p), len(q), ptr(t, u8p), len(t), mode, bw, M, X, O, E, Q, P,
                                   ptr(res, i32p), ptr(cig, u32p), cap)
    else:
        m = np.ascontiguousarray(mtx, dtype=np.int8)
        n = lib.ref_align_pairwise_mtx(lib._ctx, ptr(q, u8p), len(q), ptr(t, u8p), len(t), mode, bw, ptr(m, i8p),
                                       O, E, Q, P, ptr(res, i32p), ptr(cig, u32p), cap)
    return res, cig[:max(n, 0)].copy(), n


def oracle_edit(q, t, mode, bw):
    q = np.ascontiguousarray(q, dtype=np.uint8)
    t = np.ascontiguousarray(t, dtype=np.uint8)
    res = np.zeros(10, dtype=np.int32)
    cap = len(q) + len(t) + 8
    cig = np.zeros(cap, dtype=np.uint32)
    n = oracle().orc_edit_pairwise(ptr(q, u8p), len(q), ptr(t, u8p), len(t), mode, bw, ptr(res, i32p), ptr(cig, u32p), cap)
    return res, cig[:max(n, 0)].copy(), n


def ref_edit(q, t, mode, bw):
    q = np.ascontiguousarray(q, dtype=np.uint8)
    t = np.ascontiguousarray(t, dtype=np.uint8)
    res = np.zeros(10, dtype=np.int32)
    cap = len(q) + len(t) + 8
    cig = np.zeros(cap, dtype=np.uint32)
    lib = ref()
    n = lib.ref_edit_pairwise(lib._ctx, ptr(q, u8p), len(q), ptr(t, u8p), len(t), mode, bw, ptr(res, i32p), ptr(cig, u32p), cap)
    return res, cig[:max(n, 0)].copy(), n


# ---------------------------------------------------------------------------
# Synthetic read pairs (numpy form of the generator)
# ---------------------------------------------------------------------------
GOLDEN = np.uint64(0x9E3779B97F4A7C15)
SEED = 20240611


def _mix(z):
    z = z.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _draws(s0, n):
    """outputs 1..n of splitmix64 seeded with s0 (random access: state only ever adds GOLDEN)"""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        return _mix(np.uint64(s0) + idx * GOLDEN)


def synth_pair(k, L, err_q32=None, eps=0.10, seed=SEED, qcap=None):
    """pair k: target = iid uniform ACGT of length L; query = target with errors at total rate eps split
    sub:ins:del = 23:31:46.  Returns (query, target) as uint8 arrays of codes 0..3."""
    if err_q32 is None:
        err_q32 = int(eps * 4294967296.0)
    with np.errstate(over="ignore"):
        sT = np.uint64(seed) ^ (np.uint64(k) * GOLDEN)
        sQ = ~sT
    nw = (L + 31) // 32
    zt = _draws(sT, nw)
    i = np.arange(L, dtype=np.uint64)
    T = ((zt[(i >> np.uint64(5)).astype(np.int64)] >> (np.uint64(2) * (i & np.uint64(31)))) & np.uint64(3)).astype(np.uint8)
    z = _draws(sQ, L)
    r = z >> np.uint64(32)
    err = r < np.uint64(err_q32)
    kind = ((z & np.uint64(0xFFFF)) % np.uint64(100)).astype(np.int64)
    aux = ((z >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64)
    sub = err & (kind < 23)
    ins = err & (kind >= 23) & (kind < 54)
    dele = err & (kind >= 54)
    cnt = np.ones(L, dtype=np.int64)
    cnt[ins] = 2
    cnt[dele] = 0
    off = np.concatenate([[0], np.cumsum(cnt)])
    qlen = int(off[-1])
    Q = np.zeros(qlen, dtype=np.uint8)
    keep = ~err
    Q[off[:-1][keep]] = T[keep]
    Q[off[:-1][sub]] = ((T[sub].astype(np.int64) + 1 + aux[sub] % 3) & 3).astype(np.uint8)
    Q[off[:-1][ins]] = (aux[ins] & 3).astype(np.uint8)
    Q[off[:-1][ins] + 1] = T[ins]
    if qcap is not None and qlen > qcap:
        Q = Q[:qcap]
    return Q, T


def mutate(rng, T, eps, ratio=(23, 31, 46)):
    """free-form mutator for property tests (numpy Generator based)"""
    out = []
    s, i_, d = ratio
    tot = s + i_ + d
    for b in T:
        if rng.random() < eps:
            k = rng.integers(tot)
            if k < s:
                out.append((int(b) + 1 + int(rng.integers(3))) & 3)
            elif k < s + i_:
                out.append(int(rng.integers(4)))
                out.append(int(b))
        else:
            out.append(int(b))
    return np.array(out, dtype=np.uint8)


def cigar_str(cig):
    return "".join("%d%s" % (int(c) >> 4, "MIDNSHP=X*"[int(c) & 0xf]) for c in cig)


def cigar_spans(cig):
    """(query bases consumed, target bases consumed)"""
    qn = tn = 0
    for c in cig:
        op, ln = int(c) & 0xf, int(c) >> 4
        if op == 0:
            qn += ln
            tn += ln
        elif op == 1:
            qn += ln
        elif op == 2:
            tn += ln
    return qn, tn
