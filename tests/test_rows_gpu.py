"""GPU parity: the POA row kernels (bsa_rows_run: row_init / row_movx + row_cal with the 4 POA profiles / row_merge)
against the oracle's row functions (which tests/test_oracle_rows.py pins against the real reference)."""
import ctypes as C

import numpy as np
import pytest
import torch

import support as S

pytestmark = pytest.mark.gpu
i8p, i32p, u8p = S.i8p, S.i32p, S.u8p


class Query(C.Structure):
    _fields_ = [("seq", u8p), ("len", C.c_uint32), ("mtx", i8p), ("hpc", C.c_int), ("bonus", C.c_int)]


def _views(block, bw, pw):
    us = block[:bw].view(np.int8)
    es = block[bw:2 * bw].view(np.int8) if pw >= 1 else np.zeros(bw, np.int8)
    qs = block[2 * bw:3 * bw].view(np.int8) if pw == 2 else np.zeros(bw, np.int8)
    ub = block[(pw + 1) * bw:(pw + 1) * bw + 68].view(np.int32)
    return us, es, qs, ub


@pytest.mark.parametrize("gaps", [(-3, -2, 0, 0), (0, -3, 0, 0), (-3, -2, -8, -1)])
@pytest.mark.parametrize("bw", [16, 64, 128, 256])
def test_row_tasks_match_oracle(ctx, gaps, bw):
    import bsalign_amd as B
    lib, o = B.lib(), S.oracle()
    o.orc_row_cal.restype = C.c_int
    rng = np.random.default_rng(bw * 3 + abs(gaps[2]) + abs(gaps[0]))
    W = bw // 16
    pw = o.orc_get_piecewise(*gaps, bw)
    M, X, RB, mode = 2, -6, 1, S.MODE_OVERLAP
    blk = lib.bsa_rows_block_bytes(bw, *gaps)
    nq, qlen_max = 6, 700
    queries = [rng.integers(0, 4, size=int(rng.integers(bw + 40, qlen_max))).astype(np.uint8) for _ in range(nq)]
    qoff = np.zeros(nq, dtype=np.uint64)
    qlen = np.array([len(q) for q in queries], dtype=np.uint32)
    acc = 0
    for k, q in enumerate(queries):
        qoff[k] = acc
        acc += len(q) + 8
    qblob = np.zeros(acc, dtype=np.uint8)
    for k, q in enumerate(queries):
        qblob[int(qoff[k]):int(qoff[k]) + len(q)] = q
    nchain, depth = 24, 14           # independent chains of row updates, one level per launch
    nrows = nchain * (depth + 2)
    rows = np.zeros(nrows * blk, dtype=np.uint8)          # host mirror maintained with the oracle
    d_rows = torch.zeros(nrows * blk, dtype=torch.uint8, device="cuda:0")
    d_q = torch.from_numpy(qblob).cuda()
    d_qoff = torch.from_numpy(qoff.view(np.int64)).cuda()
    d_qlen = torch.from_numpy(qlen.view(np.int32)).cuda()
    par = B.RowsParams(mode, bw, M, X, RB, *gaps)
    mtxs = [S.score_matrix(M, X), S.score_matrix(M + RB, X)]

    def run(tasks):
        t = np.array(tasks, dtype=B.ROW_TASK_DTYPE)
        d_t = torch.from_numpy(t.view(np.uint8)).cuda()
        rc = lib.bsa_rows_run(ctx.h, C.c_void_p(d_rows.data_ptr()), C.c_void_p(d_t.data_ptr()), len(t), C.c_void_p(d_q.data_ptr()),
                              C.c_void_p(d_qoff.data_ptr()), C.c_void_p(d_qlen.data_ptr()), C.byref(par))
        assert rc == 0
        ctx.sync()
        torch.cuda.synchronize()

    def blockv(idx):
        return rows[idx * blk:(idx + 1) * blk]

    # level 0: init every chain's first row (device) and with the oracle (host)
    tasks = []
    for c in range(nchain):
        dst = c * (depth + 2)
        tasks.append((B.lib() and 2, 0, dst, 0, 0, 0, c % nq, 0, 0, 0))
        us, es, qs, ub = _views(blockv(dst), bw, pw)
        ubt = np.zeros(17, dtype=np.int32)
        uu, ee, qq = np.zeros(bw, np.int8), np.zeros(bw, np.int8), np.zeros(bw, np.int8)
        o.orc_row_init(S.ptr(uu, i8p), S.ptr(ee, i8p), S.ptr(qq, i8p), S.ptr(ubt, i32p), mode, bw, M + RB + 1, X, *gaps)
        us[:] = uu
        if pw >= 1:
            es[:] = ee
        if pw == 2:
            qs[:] = qq
        ub[:] = ubt
    run(tasks)
    state = [(0, c % nq) for c in range(nchain)]       # (band offset, query) per chain
    for lev in range(depth):
        tasks, expect = [], []
        for c in range(nchain):
            src = c * (depth + 2) + lev
            dst = src + 1
            qoff_src, qi = state[c]
            ql = int(qlen[qi])
            movx = int(rng.choice([0, 1, 1, 2, 3, W, W + 1, 2 * W + 1, bw - 1, bw, bw + 3]))
            if qoff_src + movx + bw > ql:
                movx = 0
            qoff_dst = qoff_src + movx
            base, prof = int(rng.integers(4)), int(rng.integers(4))
            tasks.append((0, src, dst, qoff_src, qoff_dst, lev + 1, qi, base, prof, 0))
            # oracle: movx then row_cal with the profile's matrix / hpc flag
            us, es, qs, ub = _views(blockv(src), bw, pw)
            mu, me, mq, mb = np.zeros(bw, np.int8), np.zeros(bw, np.int8), np.zeros(bw, np.int8), np.zeros(17, np.int32)
            o.orc_row_movx(S.ptr(mu, i8p), S.ptr(me, i8p), S.ptr(mq, i8p), S.ptr(mb, i32p),
                           S.ptr(np.ascontiguousarray(us), i8p), S.ptr(np.ascontiguousarray(es), i8p), S.ptr(np.ascontiguousarray(qs), i8p), S.ptr(np.ascontiguousarray(ub), i32p),
                           W, movx, pw, M + RB + 1, X, *gaps)
            if movx == 0:
                rh = -(0x7FFFFFFF >> 2) if qoff_src else (0 if mode == S.MODE_OVERLAP else gaps[0] + gaps[1] * (lev + 1))
            elif movx <= bw:
                rh = int(mb[0])
            else:
                rh = -(0x7FFFFFFF >> 2)
            q = queries[qi]
            mtx = mtxs[prof & 1]
            qy = Query(S.ptr(q, u8p), len(q), S.ptr(mtx, i8p), 0 if (prof & 2) else 1, 1)
            nu, ne, nqv, nb = np.zeros(bw, np.int8), np.zeros(bw, np.int8), np.zeros(bw, np.int8), np.zeros(17, np.int32)
            o.orc_row_cal(qoff_dst, base, S.ptr(mu, i8p), S.ptr(me, i8p), S.ptr(mq, i8p), S.ptr(mb, i32p),
                          S.ptr(nu, i8p), S.ptr(ne, i8p), S.ptr(nqv, i8p), S.ptr(nb, i32p), C.byref(qy), *gaps, W, rh, pw)
            dus, des, dqs, dub = _views(blockv(dst), bw, pw)
            dus[:] = nu
            if pw >= 1:
                des[:] = ne
            if pw == 2:
                dqs[:] = nqv
            dub[:] = nb
            state[c] = (qoff_dst, qi)
        run(tasks)
        got = d_rows.cpu().numpy()
        for c in range(nchain):
            dst = c * (depth + 2) + lev + 1
            used = (pw + 1) * bw + 68
            assert np.array_equal(got[dst * blk:dst * blk + used], rows[dst * blk:dst * blk + used]), ("update", gaps, bw, lev, c, tasks[c])
    # merges: rows[dst] = max(rows[src], rows[dst]) for pairs of chains
    tasks = []
    for c in range(0, nchain - 1, 2):
        src = c * (depth + 2) + depth
        dst = (c + 1) * (depth + 2) + depth
        tasks.append((1, src, dst, 0, 0, 0, 0, 0, 0, 0))
        a0 = [np.ascontiguousarray(x) for x in _views(blockv(src), bw, pw)]
        a1 = [np.ascontiguousarray(x) for x in _views(blockv(dst), bw, pw)]
        mu, me, mq, mb = np.zeros(bw, np.int8), np.zeros(bw, np.int8), np.zeros(bw, np.int8), np.zeros(17, np.int32)
        o.orc_row_merge(S.ptr(a0[0], i8p), S.ptr(a0[1], i8p), S.ptr(a0[2], i8p), S.ptr(a0[3], i32p),
                        S.ptr(a1[0], i8p), S.ptr(a1[1], i8p), S.ptr(a1[2], i8p), S.ptr(a1[3], i32p),
                        S.ptr(mu, i8p), S.ptr(me, i8p), S.ptr(mq, i8p), S.ptr(mb, i32p), W, pw)
        dus, des, dqs, dub = _views(blockv(dst), bw, pw)
        dus[:] = mu
        if pw >= 1:
            des[:] = me
        if pw == 2:
            dqs[:] = mq
        dub[:] = mb
    run(tasks)
    got = d_rows.cpu().numpy()
    for c in range(0, nchain - 1, 2):
        dst = (c + 1) * (depth + 2) + depth
        used = (pw + 1) * bw + 68
        assert np.array_equal(got[dst * blk:dst * blk + used], rows[dst * blk:dst * blk + used]), ("merge", gaps, bw, c)
