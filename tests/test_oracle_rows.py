"""CPU: the oracle's row-level functions (row_init / row_movx incl. jumps >= W / row_cal with the POA's
homopolymer-bonus profiles / row_merge / row_max / band_mov) against the real reference, when it is built here.
These are the functions the POA seq->graph DP drives directly (bspoa.h:2232-2272)."""
import ctypes as C

import numpy as np
import pytest

import support as S

pytestmark = pytest.mark.skipif(not S.have_ref(), reason="reference build (oracle/_ref) not present")

i8p, i32p, u8p = S.i8p, S.i32p, S.u8p


def _al(n, dt):
    """16-byte aligned numpy array (the reference uses aligned SSE loads)"""
    raw = np.zeros(n * np.dtype(dt).itemsize + 16, dtype=np.uint8)
    off = (-raw.ctypes.data) % 16
    return raw[off:off + n * np.dtype(dt).itemsize].view(dt)


def _libs():
    o, r = S.oracle(), S.ref()
    o.orc_row_movx.restype = None
    o.orc_row_cal.restype = C.c_int
    o.orc_row_merge.restype = None
    o.orc_row_max.restype = C.c_uint32
    o.orc_getscore.restype = C.c_int
    o.orc_band_mov.restype = C.c_int
    r.ref_row_movx.restype = None
    r.ref_row_cal.restype = C.c_int
    r.ref_row_merge.restype = None
    r.ref_row_max.restype = C.c_uint32
    r.ref_band_mov.restype = C.c_int
    r.ref_qprof_size.restype = C.c_uint64
    return o, r


class Query(C.Structure):
    _fields_ = [("seq", u8p), ("len", C.c_uint32), ("mtx", i8p), ("hpc", C.c_int), ("bonus", C.c_int)]


def _row(rng, bw, pw, mode, gaps, lib_ref, W):
    """a plausible row: run the reference's row_init, then a few row_cal steps on a random query"""
    us, es, qs, ub = _al(bw, np.int8), _al(bw, np.int8), _al(bw, np.int8), _al(20, np.int32)
    lib_ref.ref_row_init(S.ptr(us, i8p), S.ptr(es, i8p), S.ptr(qs, i8p), S.ptr(ub, i32p), mode, bw, 3, -6, *gaps)
    return us, es, qs, ub


@pytest.mark.parametrize("gaps", [(-3, -2, 0, 0), (0, -3, 0, 0), (-3, -2, -8, -1)])
@pytest.mark.parametrize("bw", [16, 64, 128])
def test_row_functions_match_reference(gaps, bw):
    o, r = _libs()
    rng = np.random.default_rng(bw + abs(gaps[0]) * 7 + abs(gaps[2]))
    W = bw // 16
    pw = o.orc_get_piecewise(*gaps, bw)
    qlen = 400
    q = rng.integers(0, 4, size=qlen).astype(np.uint8)
    mtxs = [S.score_matrix(2, -6), S.score_matrix(3, -6)]
    for trial in range(60):
        mode = int(rng.choice([0, 1]))
        use_hpc = int(rng.integers(2))
        mtx = mtxs[int(rng.integers(2))]
        # reference profile for this (matrix, hpc) choice (bspoa.h:2199-2215)
        qprof = _al(int(r.ref_qprof_size(qlen, bw)) + 64, np.int8)
        if use_hpc:
            r.ref_set_query_prof_hpc(S.ptr(q, u8p), qlen, S.ptr(qprof, i8p), bw, S.ptr(mtx, i8p), 1)
        else:
            r.ref_set_query_prof(S.ptr(q, u8p), qlen, S.ptr(qprof, i8p), bw, S.ptr(mtx, i8p))
        qy = Query(S.ptr(q, u8p), qlen, S.ptr(mtx, i8p), use_hpc, 1)
        us, es, qs, ub = _row(rng, bw, pw, mode, gaps, r, W)
        rbeg = 0
        for step in range(12):
            movx = int(rng.choice([0, 1, 1, 2, 3, W, W + 1, 2 * W + 3, bw - 1, bw, bw + 5])) if step else 0
            if rbeg + movx + bw > qlen:      # the reference only guarantees profile rows up to qlen (bsalign.h:2147)
                movx = 0
            base = int(rng.integers(4))
            # ---- movx: reference vs oracle
            ru, re, rq, rb = _al(bw, np.int8), _al(bw, np.int8), _al(bw, np.int8), _al(20, np.int32)
            ou, oe, oq, ob = _al(bw, np.int8), _al(bw, np.int8), _al(bw, np.int8), _al(20, np.int32)
            r.ref_row_movx(S.ptr(ru, i8p), S.ptr(re, i8p), S.ptr(rq, i8p), S.ptr(rb, i32p), S.ptr(us, i8p), S.ptr(es, i8p), S.ptr(qs, i8p), S.ptr(ub, i32p),
                           W, movx, pw, 3, -6, *gaps)
            o.orc_row_movx(S.ptr(ou, i8p), S.ptr(oe, i8p), S.ptr(oq, i8p), S.ptr(ob, i32p), S.ptr(us, i8p), S.ptr(es, i8p), S.ptr(qs, i8p), S.ptr(ub, i32p),
                           W, movx, pw, 3, -6, *gaps)
            assert np.array_equal(ru, ou) and np.array_equal(rb[:17], ob[:17]), ("movx u/ubegs", gaps, bw, movx, step)
            if pw >= 1:
                assert np.array_equal(re, oe), ("movx e", movx)
            if pw == 2:
                assert np.array_equal(rq, oq), ("movx q", movx)
            rbeg += movx
            # ---- row_cal on the moved row (rh as dpalign_row_update_bspoa computes it, bspoa.h:2243-2255)
            if movx == 0:
                rh = -(0x7FFFFFFF >> 2) if rbeg else (0 if (mode == 1 or step == 0) else gaps[0] + gaps[1] * step)
            elif movx <= bw:
                rh = int(rb[0])
            else:
                rh = -(0x7FFFFFFF >> 2)
            r2u, r2e, r2q, r2b = _al(bw, np.int8), _al(bw, np.int8), _al(bw, np.int8), _al(20, np.int32)
            o2u, o2e, o2q, o2b = _al(bw, np.int8), _al(bw, np.int8), _al(bw, np.int8), _al(20, np.int32)
            r.ref_row_cal(rbeg, base, S.ptr(ru, i8p), S.ptr(re, i8p), S.ptr(rq, i8p), S.ptr(rb, i32p),
                          S.ptr(r2u, i8p), S.ptr(r2e, i8p), S.ptr(r2q, i8p), S.ptr(r2b, i32p), S.ptr(qprof, i8p), *gaps, W, movx, rh, pw)
            o.orc_row_cal(rbeg, base, S.ptr(ou, i8p), S.ptr(oe, i8p), S.ptr(oq, i8p), S.ptr(ob, i32p),
                          S.ptr(o2u, i8p), S.ptr(o2e, i8p), S.ptr(o2q, i8p), S.ptr(o2b, i32p), C.byref(qy), *gaps, W, rh, pw)
            assert np.array_equal(r2u, o2u) and np.array_equal(r2b[:17], o2b[:17]), ("row_cal u/ubegs", gaps, bw, movx, step, use_hpc)
            if pw >= 1:
                assert np.array_equal(r2e, o2e)
            if pw == 2:
                assert np.array_equal(r2q, o2q)
            # ---- row_max / band_mov on the new row
            ms_r, ms_o = C.c_int32(), C.c_int32()
            xr = r.ref_row_max(S.ptr(r2u, i8p), S.ptr(r2b, i32p), W, C.byref(ms_r))
            xo = o.orc_row_max(S.ptr(o2u, i8p), S.ptr(o2b, i32p), W, C.byref(ms_o))
            assert (xr, ms_r.value) == (xo, ms_o.value)
            assert r.ref_band_mov(S.ptr(r2u, i8p), S.ptr(r2b, i32p), W, 50 + step, rbeg, qlen) == o.orc_band_mov(S.ptr(o2b, i32p), W, 50 + step, rbeg, qlen)
            # ---- row_merge of the moved row and the new row (two progenitors of one graph node, bspoa.h:2263-2272)
            m_r = [_al(bw, np.int8) for _ in range(3)] + [_al(20, np.int32)]
            m_o = [_al(bw, np.int8) for _ in range(3)] + [_al(20, np.int32)]
            r.ref_row_merge(S.ptr(ru, i8p), S.ptr(re, i8p), S.ptr(rq, i8p), S.ptr(rb, i32p), S.ptr(r2u, i8p), S.ptr(r2e, i8p), S.ptr(r2q, i8p), S.ptr(r2b, i32p),
                            S.ptr(m_r[0], i8p), S.ptr(m_r[1], i8p), S.ptr(m_r[2], i8p), S.ptr(m_r[3], i32p), W, pw)
            o.orc_row_merge(S.ptr(ou, i8p), S.ptr(oe, i8p), S.ptr(oq, i8p), S.ptr(ob, i32p), S.ptr(o2u, i8p), S.ptr(o2e, i8p), S.ptr(o2q, i8p), S.ptr(o2b, i32p),
                            S.ptr(m_o[0], i8p), S.ptr(m_o[1], i8p), S.ptr(m_o[2], i8p), S.ptr(m_o[3], i32p), W, pw)
            assert np.array_equal(m_r[0], m_o[0]) and np.array_equal(m_r[3][:17], m_o[3][:17]), ("merge", gaps, bw, step)
            if pw >= 1:
                assert np.array_equal(m_r[1], m_o[1])
            if pw == 2:
                assert np.array_equal(m_r[2], m_o[2])
            us, es, qs, ub = r2u, r2e, r2q, r2b
