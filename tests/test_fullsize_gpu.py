"""GPU: BASELINE configuration C2 at its full size (100 000 synthetic 10 kbp pairs, global, bandwidth 128) through the
two-phase C-ABI on device-resident data, checked by properties that do not need the oracle on every pair: every pair
unflagged and aligned end to end, aln = mat + mis + ins + del, every CIGAR consumes exactly its query and target, the
run is idempotent, and a sample of pairs spread over the batch equals the oracle bit for bit."""
import ctypes as C

import numpy as np
import pytest

import support as S

pytestmark = pytest.mark.gpu


def _align8_properties(n, L, bw, sample):
    import torch
    import bsalign_amd as B
    sc = (2, -6, -3, -2, 0, 0)
    dev = torch.device("cuda", 0)
    ctx = B.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    lib = B.lib()
    stride = lib.bsa_synth_stride(L)
    d_seqs = torch.empty(2 * n * stride, dtype=torch.uint8, device=dev)
    d_qlen = torch.empty(n, dtype=torch.int32, device=dev)
    assert lib.bsa_synth_pairs_dev(ctx.h, S.SEED, 0, n, L, int(0.10 * 4294967296.0), C.c_void_p(d_seqs.data_ptr()), C.c_void_p(d_qlen.data_ptr())) == 0
    torch.cuda.synchronize()
    qlen = d_qlen.cpu().numpy().astype(np.uint32)
    tlen = np.full(n, L, dtype=np.uint32)
    toff = np.arange(n, dtype=np.uint64) * np.uint64(stride)
    qoff = (np.arange(n, dtype=np.uint64) + np.uint64(n)) * np.uint64(stride)
    plan = B.AlignPlan(ctx, qoff, qlen, toff, tlen, B.make_params(B.MODE_GLOBAL, bw, *sc))
    cap = n * (L // 4)
    d_out = torch.zeros(n * 10, dtype=torch.int32, device=dev)
    d_cig = torch.empty(cap, dtype=torch.int32, device=dev)
    d_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    sums = []
    for _ in range(2):
        plan.run(d_seqs, d_out, d_cig, d_off, d_st)
        torch.cuda.synchronize()
        sums.append((int(d_out.to(torch.int64).sum().item()), int(d_off[n].item()),
                     int((d_cig[: int(d_off[n].item())].to(torch.int64) * 2654435761 % 1000003).sum().item())))
    assert sums[0] == sums[1], "the same run twice gives different results"
    out = d_out.cpu().numpy().reshape(n, 10)
    off = d_off.cpu().numpy()
    assert not d_st.cpu().numpy().any(), "flagged pairs"
    score, qb, qe, tb, te, mat, mis, ins, dele, aln = (out[:, k] for k in range(10))
    assert (qb == 0).all() and (tb == 0).all() and (qe == qlen.astype(np.int32)).all() and (te == L).all()
    assert (aln == mat + mis + ins + dele).all()
    assert (score <= 2 * np.minimum(qlen, L).astype(np.int64)).all() and (score > 0).all()
    # every CIGAR consumes exactly its query and target: per-pair sums of the run lengths by operation
    cig = d_cig[: int(off[n])].cpu().numpy().view(np.uint32)
    op, ln = cig & 0xF, (cig >> 4).astype(np.int64)
    assert np.isin(op, (0, 1, 2)).all()
    starts = off[:-1].astype(np.int64)
    assert (np.diff(off) > 0).all()
    qsum = np.add.reduceat(np.where(op != 2, ln, 0), starts)
    tsum = np.add.reduceat(np.where(op != 1, ln, 0), starts)
    asum = np.add.reduceat(ln, starts)
    assert (qsum == qlen).all() and (tsum == L).all() and (asum == aln).all()
    assert (np.add.reduceat(np.where(op == 1, ln, 0), starts) == ins).all() and (np.add.reduceat(np.where(op == 2, ln, 0), starts) == dele).all()
    # consecutive CIGAR words of a pair never repeat an operation (runs are merged)
    same = op[1:] == op[:-1]
    same[(starts[1:] - 1)] = False
    assert not same.any()
    # a sample spread over the batch (first, last and the tail round of the launch) against the oracle
    for k in sample:
        q, t = S.synth_pair(k, L)
        res, cg, _ = S.oracle_align(q, t, S.MODE_GLOBAL, bw, *sc)
        assert np.array_equal(out[k], res) and np.array_equal(cig[int(off[k]):int(off[k + 1])], cg), k
    plan.close()
    ctx.close()


def test_c2_full_size_properties():
    """configuration C2: 100 000 x 10 kbp, global, bandwidth 128"""
    _align8_properties(100000, 10000, 128, (0, 1, 4999, 33333, 65535, 65536, 98303, 98304, 99998, 99999))


def test_c5_per_gpu_shape_properties():
    """configuration C5 is 10 M pairs x 15 kbp over 8 GPUs; what one GPU sees of it is a batch of 15 kbp pairs larger than
    its workspace, run in chunks: 200 000 x 15 kbp here (the 8-GPU run itself needs the hardware)"""
    _align8_properties(200000, 15000, 128, (0, 1, 65535, 65536, 100000, 131071, 131072, 199998, 199999))


def test_c3_full_size_properties():
    """configuration C3: 16384 synthetic 100 kbp pairs, edit path, global, bandwidth 256"""
    import torch
    import bsalign_amd as B
    n, L, bw = 16384, 100000, 256
    dev = torch.device("cuda", 0)
    ctx = B.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    lib = B.lib()
    stride = lib.bsa_synth_stride(L)
    d_seqs = torch.empty(2 * n * stride, dtype=torch.uint8, device=dev)
    d_qlen = torch.empty(n, dtype=torch.int32, device=dev)
    assert lib.bsa_synth_pairs_dev(ctx.h, S.SEED, 0, n, L, int(0.10 * 4294967296.0), C.c_void_p(d_seqs.data_ptr()), C.c_void_p(d_qlen.data_ptr())) == 0
    torch.cuda.synchronize()
    qlen = d_qlen.cpu().numpy().astype(np.uint32)
    tlen = np.full(n, L, dtype=np.uint32)
    toff = np.arange(n, dtype=np.uint64) * np.uint64(stride)
    qoff = (np.arange(n, dtype=np.uint64) + np.uint64(n)) * np.uint64(stride)
    plan = B.EditPlan(ctx, qoff, qlen, toff, tlen, B.MODE_GLOBAL, bw)
    cap = n * (L // 4)
    d_out = torch.zeros(n * 10, dtype=torch.int32, device=dev)
    d_cig = torch.empty(cap, dtype=torch.int32, device=dev)
    d_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    sums = []
    for _ in range(2):
        plan.run(d_seqs, d_out, d_cig, d_off, d_st)
        torch.cuda.synchronize()
        sums.append((int(d_out.to(torch.int64).sum().item()), int(d_off[n].item()),
                     int((d_cig[: int(d_off[n].item())].to(torch.int64) * 2654435761 % 1000003).sum().item())))
    assert sums[0] == sums[1], "the same run twice gives different results"
    out = d_out.cpu().numpy().reshape(n, 10)
    off = d_off.cpu().numpy()
    assert not d_st.cpu().numpy().any(), "flagged pairs"
    score, qb, qe, tb, te, mat, mis, ins, dele, aln = (out[:, k] for k in range(10))
    assert (qb == 0).all() and (tb == 0).all() and (qe == qlen.astype(np.int32)).all() and (te == L).all()
    assert (aln == mat + mis + ins + dele).all()
    # (no relation between the score and mis + ins + del is checked: the band's edge rules make the DP value neither a lower
    # nor an upper bound of the cost of the path the reference's backtrace reports, bsalign.h:985-1015)
    assert (score > 0).all()
    cig = d_cig[: int(off[n])].cpu().numpy().view(np.uint32)
    op, ln = cig & 0xF, (cig >> 4).astype(np.int64)
    starts = off[:-1].astype(np.int64)
    assert np.isin(op, (0, 1, 2)).all() and (np.diff(off) > 0).all()
    assert (np.add.reduceat(np.where(op != 2, ln, 0), starts) == qlen).all() and (np.add.reduceat(np.where(op != 1, ln, 0), starts) == L).all()
    assert (np.add.reduceat(np.where(op == 1, ln, 0), starts) == ins).all() and (np.add.reduceat(np.where(op == 2, ln, 0), starts) == dele).all()
    for k in (0, 4095, 8192, 16383):
        q, t = S.synth_pair(k, L)
        res, cg, _ = S.oracle_edit(q, t, S.MODE_GLOBAL, bw)
        assert np.array_equal(out[k], res) and np.array_equal(cig[int(off[k]):int(off[k + 1])], cg), k
    plan.close()
    ctx.close()
