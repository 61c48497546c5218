"""GPU: BASELINE configuration C2 at its full size (100 000 synthetic 10 kbp pairs, global, bandwidth 128) through the
two-phase C-ABI on device-resident data, checked by properties that do not need the oracle on every pair: every pair
unflagged and aligned end to end, aln = mat + mis + ins + del, every CIGAR consumes exactly its query and target, the
run is idempotent, and a sample of pairs spread over the batch equals the oracle bit for bit."""
import ctypes as C

import numpy as np
import pytest

import support as S

pytestmark = pytest.mark.gpu


def _align8_properties(n, L, bw, sample, nref=1024):
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
    # ... and `nref` pairs, evenly spread, against the REFERENCE ITSELF (oracle/_ref/libbsref.so: the reference's own SSE code compiled from
    # /root/reference, 2.4 ms a pair; the oracle restatement where that build is absent): result struct and every CIGAR word
    align = S.ref_align if S.have_ref() else (lambda *a: S.oracle_align(*a)[:2])
    ks = np.unique(np.linspace(0, n - 1, nref).astype(np.int64))
    for k in ks:
        q, t = S.synth_pair(int(k), L)
        res, cg = align(q, t, S.MODE_GLOBAL, bw, *sc)[:2]
        assert np.array_equal(out[k], res) and np.array_equal(cig[int(off[k]):int(off[k + 1])], cg), int(k)
    print("\n[%d x %d bp] %d pairs spread over the batch identical to %s" % (n, L, len(ks), "the reference (libbsref.so)" if S.have_ref() else "the oracle"))
    plan.close()
    ctx.close()


def test_c2_full_size_properties():
    """configuration C2: 100 000 x 10 kbp, global, bandwidth 128"""
    _align8_properties(100000, 10000, 128, (0, 1, 4999, 33333, 65535, 65536, 98303, 98304, 99998, 99999))


def test_c5_per_gpu_shape_properties():
    """configuration C5 is 10 M pairs x 15 kbp over 8 GPUs; what one GPU sees of it is a batch of 15 kbp pairs larger than
    its workspace, run in chunks: 200 000 x 15 kbp here (the 8-GPU run itself needs the hardware)"""
    _align8_properties(200000, 15000, 128, (0, 1, 65535, 65536, 100000, 131071, 131072, 199998, 199999))


def _checksums(d_out, d_off, d_cig, d_st, n):
    import torch
    nw = int(d_off[n].item())
    w = d_cig[:nw].to(torch.int64) & 0xFFFFFFFF
    i = torch.arange(nw, device=w.device, dtype=torch.int64)
    return (int((d_out.to(torch.int64) * (torch.arange(d_out.numel(), device=d_out.device, dtype=torch.int64) % 1000003 + 1)).sum().item()), nw,
            int(((w * 2654435761 + i) % 1000000007).sum().item()), int(d_st.to(torch.int64).sum().item()), int((d_off.to(torch.int64) % 1000003).sum().item()))


def test_c2_whole_batch_row_segments_equal_whole_pairs(monkeypatch):
    """The headline kernel hands a pair's band state from wave to wave through memory (k_align8_fwd_xq: write-through stores + vmcnt(0) + flag; reader:
    poll + agent acquire).  A hand-over ordering fault would be a valid-looking but different alignment, invisible to "two runs agree": here the WHOLE
    C2 batch runs five times in row segments and once as whole pairs (BSA_ALIGN8_XQ=0: no hand-over at all) and results, status words, CIGAR offsets and
    every CIGAR word (position-sensitive checksums) must be the same."""
    import torch
    import bsalign_amd as B
    n, L, bw, sc = 100000, 10000, 128, (2, -6, -3, -2, 0, 0)
    dev = torch.device("cuda", 0)
    lib = B.lib()
    stride = lib.bsa_synth_stride(L)
    d_seqs = torch.empty(2 * n * stride, dtype=torch.uint8, device=dev)
    d_qlen = torch.empty(n, dtype=torch.int32, device=dev)
    got = {}
    for xq, reps in (("0", 1), ("1", 5)):
        monkeypatch.setenv("BSA_ALIGN8_XQ", xq)
        ctx = B.Context(0)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        if not got:
            assert lib.bsa_synth_pairs_dev(ctx.h, S.SEED, 0, n, L, int(0.10 * 4294967296.0), C.c_void_p(d_seqs.data_ptr()), C.c_void_p(d_qlen.data_ptr())) == 0
            torch.cuda.synchronize()
            qlen = d_qlen.cpu().numpy().astype(np.uint32)
            tlen = np.full(n, L, dtype=np.uint32)
            toff = np.arange(n, dtype=np.uint64) * np.uint64(stride)
            qoff = (np.arange(n, dtype=np.uint64) + np.uint64(n)) * np.uint64(stride)
        plan = B.AlignPlan(ctx, qoff, qlen, toff, tlen, B.make_params(B.MODE_GLOBAL, bw, *sc))
        d_out = torch.zeros(n * 10, dtype=torch.int32, device=dev)
        d_cig = torch.empty(n * (L // 4), dtype=torch.int32, device=dev)
        d_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        d_st = torch.zeros(n, dtype=torch.int32, device=dev)
        for r in range(reps):
            d_out.zero_(); d_off.zero_(); d_st.zero_()
            plan.run(d_seqs, d_out, d_cig, d_off, d_st)
            torch.cuda.synchronize()
            got.setdefault(xq, []).append(_checksums(d_out, d_off, d_cig, d_st, n))
        name = ctx.last_kernel_names()[0]
        assert ("fwd_xq" in name) == (xq == "1"), name
        plan.close()
        ctx.close()
        del d_out, d_cig, d_off, d_st
    assert got["0"][0][3] == 0, "flagged pairs"
    assert all(c == got["0"][0] for c in got["1"]), (got["0"], got["1"])


def test_a_hand_over_that_never_comes_is_flagged_not_hung(monkeypatch):
    """k_align8_fwd_xq's wait for the previous segment is bounded: with the bound forced to one turn (BSA_ALIGN8_XQ_SPIN_CAP=1) on a launch whose later segments
    start while the first still runs, waves give up -- their pairs come back flagged BSA_ST_DEVICE with zeroed records, every other pair is right, nothing hangs,
    and the host-pointer entry reports BSA_E_HIP"""
    import bsalign_amd as B
    monkeypatch.setenv("BSA_ALIGN8_XQ", "1")
    monkeypatch.setenv("BSA_ALIGN8_XQ_SEG", "64")
    monkeypatch.setenv("BSA_ALIGN8_XQ_SPIN_CAP", "1")
    pairs = [S.synth_pair(k, 1500) for k in range(256)]
    ctx = B.Context(0)
    try:
        with pytest.raises(B.BsaError) as ei:
            ctx.align_batch(pairs, B.make_params(B.MODE_GLOBAL, 128, 2, -6, -3, -2, 0, 0))
        assert ei.value.code == -4
        monkeypatch.setenv("BSA_ALIGN8_XQ_SPIN_CAP", "4194304")
        res, cigs, st = ctx.align_batch(pairs, B.make_params(B.MODE_GLOBAL, 128, 2, -6, -3, -2, 0, 0))
        assert not np.asarray(st).any()
        for k in (0, 100, 255):
            r, cg, _ = S.oracle_align(pairs[k][0], pairs[k][1], S.MODE_GLOBAL, 128, 2, -6, -3, -2, 0, 0)
            assert np.array_equal(np.array([res[k][f] for f in res.dtype.names], dtype=np.int32), r) and np.array_equal(cigs[k], cg)
    finally:
        ctx.close()


def test_c5_real_per_gpu_share_in_chunks():
    """configuration C5: 10 M pairs x 15 kbp over 8 GPUs = 1.25 M pairs a GPU (18.75 Gbases a side: 37.5 GB of input, 2.4 * 10^12 band cells).  One GPU's real
    share, generated on the device in five batches of 250 000 pairs (the pair indices of rank 3's range of the global stream), each batch through the plan in
    workspace chunks: every pair unflagged and end to end, aln = mat + mis + ins + del, CIGARs consume exactly their sequences, and 1024 pairs spread over the
    share equal the reference itself"""
    import torch
    import time
    import bsalign_amd as B
    L, bw, sc = 15000, 128, (2, -6, -3, -2, 0, 0)
    share, nb = 1250000, 250000
    first = 3 * share
    dev = torch.device("cuda", 0)
    ctx = B.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    lib = B.lib()
    stride = lib.bsa_synth_stride(L)
    d_seqs = torch.empty(2 * nb * stride, dtype=torch.uint8, device=dev)
    d_qlen = torch.empty(nb, dtype=torch.int32, device=dev)
    d_out = torch.zeros(nb * 10, dtype=torch.int32, device=dev)
    d_cig = torch.empty(nb * (L // 4), dtype=torch.int32, device=dev)
    d_off = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
    d_st = torch.zeros(nb, dtype=torch.int32, device=dev)
    align = S.ref_align if S.have_ref() else (lambda *a: S.oracle_align(*a)[:2])
    cells, secs, checked = 0.0, 0.0, 0
    for b in range(share // nb):
        k0 = first + b * nb
        assert lib.bsa_synth_pairs_dev(ctx.h, S.SEED, k0, nb, L, int(0.10 * 4294967296.0), C.c_void_p(d_seqs.data_ptr()), C.c_void_p(d_qlen.data_ptr())) == 0
        torch.cuda.synchronize()
        qlen = d_qlen.cpu().numpy().astype(np.uint32)
        tlen = np.full(nb, L, dtype=np.uint32)
        toff = np.arange(nb, dtype=np.uint64) * np.uint64(stride)
        qoff = (np.arange(nb, dtype=np.uint64) + np.uint64(nb)) * np.uint64(stride)
        plan = B.AlignPlan(ctx, qoff, qlen, toff, tlen, B.make_params(B.MODE_GLOBAL, bw, *sc))
        t0 = time.perf_counter()
        plan.run(d_seqs, d_out, d_cig, d_off, d_st)
        torch.cuda.synchronize()
        secs += time.perf_counter() - t0
        cells += plan.cells()
        plan.close()
        out = d_out.cpu().numpy().reshape(nb, 10)
        off = d_off.cpu().numpy()
        assert not d_st.cpu().numpy().any(), "flagged pairs"
        score, qb, qe, tb, te, mat, mis, ins, dele, aln = (out[:, k] for k in range(10))
        assert (qb == 0).all() and (tb == 0).all() and (qe == qlen.astype(np.int32)).all() and (te == L).all() and (aln == mat + mis + ins + dele).all() and (score > 0).all()
        cig = d_cig[: int(off[nb])].cpu().numpy().view(np.uint32)
        op, ln = cig & 0xF, (cig >> 4).astype(np.int64)
        starts = off[:-1].astype(np.int64)
        assert np.isin(op, (0, 1, 2)).all() and (np.diff(off) > 0).all()
        assert (np.add.reduceat(np.where(op != 2, ln, 0), starts) == qlen).all() and (np.add.reduceat(np.where(op != 1, ln, 0), starts) == L).all()
        for k in np.unique(np.linspace(0, nb - 1, 205).astype(np.int64)):
            q, t = S.synth_pair(int(k0 + k), L)
            res, cg = align(q, t, S.MODE_GLOBAL, bw, *sc)[:2]
            assert np.array_equal(out[k], res) and np.array_equal(cig[int(off[k]):int(off[k + 1])], cg), int(k0 + k)
            checked += 1
    print("\n[C5 per-GPU share: %d x %d bp] %.0f GCUPS over the five batches (run + synchronise, chunked workspace), %d pairs identical to %s"
          % (share, L, cells / secs / 1e9, checked, "the reference (libbsref.so)" if S.have_ref() else "the oracle"))
    ctx.close()


@pytest.mark.parametrize("n", [16384, 32768])
def test_c3_full_size_properties(n):
    """configuration C3: synthetic 100 kbp pairs, edit path, global, bandwidth 256 -- 16384 pairs (the batch of the earlier rounds) and 32768 (bench.py's
    batch since round 5: 210 GB of row planes)"""
    import torch
    import bsalign_amd as B
    L, bw = 100000, 256
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
    for k in (0, n // 4 - 1, n // 2, n - 1):
        q, t = S.synth_pair(k, L)
        res, cg, _ = S.oracle_edit(q, t, S.MODE_GLOBAL, bw)
        assert np.array_equal(out[k], res) and np.array_equal(cig[int(off[k]):int(off[k + 1])], cg), k
    # 64 pairs spread over the batch against the reference itself (libbsref.so; the oracle where that build is absent)
    edit = S.ref_edit if S.have_ref() else (lambda *a: S.oracle_edit(*a)[:2])
    for k in np.unique(np.linspace(0, n - 1, 64).astype(np.int64)):
        q, t = S.synth_pair(int(k), L)
        res, cg = edit(q, t, S.MODE_GLOBAL, bw)[:2]
        assert np.array_equal(out[k], res) and np.array_equal(cig[int(off[k]):int(off[k + 1])], cg), int(k)
    plan.close()
    ctx.close()
