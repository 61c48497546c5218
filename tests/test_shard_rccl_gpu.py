"""GPU: the C-level shard exchange (bsa_shard_scatter / bsa_shard_gather, bsalign_amd/csrc/bsa_shard_rccl.hip) on the one GPU of the
test box: a communicator of one rank takes every code path that does not need a peer -- lengths, ranges, packing, the device blob, the
gather's size exchange and reassembly -- and the aligned shard gives the results of bsa_align_batch.  (The N-rank messages are RCCL
ncclSend / ncclRecv groups; their rank arithmetic is what tests/test_shard_cpu.py covers for the Python twin.)"""
import ctypes as C

import numpy as np
import pytest

import support as S

pytestmark = pytest.mark.gpu


def test_one_rank_scatter_align_gather(ctx):
    import bsalign_amd as B
    L = B.lib()
    vp = C.c_void_p
    L.bsa_shard_comm_create.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.bsa_shard_comm_destroy.argtypes = [vp]
    L.bsa_shard_comm_destroy.restype = None
    L.bsa_shard_scatter.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, C.c_size_t, C.c_uint32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(vp),
                                    C.POINTER(C.c_size_t), vp, vp, vp, vp, C.c_size_t]
    L.bsa_shard_gather.argtypes = [vp, C.c_int, vp, vp, vp, C.c_size_t, vp, vp, C.c_size_t, vp, C.c_size_t]
    rng = np.random.default_rng(31)
    pairs = []
    for _ in range(300):
        T = rng.integers(0, 4, size=int(rng.integers(50, 900))).astype(np.uint8)
        pairs.append((S.mutate(rng, T, 0.1), T))
    n = len(pairs)
    seqs = np.concatenate([np.concatenate([q, t]) for q, t in pairs])
    qlen = np.array([len(q) for q, _ in pairs], np.uint32); tlen = np.array([len(t) for _, t in pairs], np.uint32)
    off = np.concatenate([[0], np.cumsum(qlen.astype(np.uint64) + tlen)[:-1]]).astype(np.uint64)
    qoff = off; toff = off + qlen
    comm = vp()
    assert L.bsa_shard_comm_create(ctx.h, 0, 1, None, C.byref(comm)) == 0
    try:
        first, count, nbytes, dptr = C.c_size_t(), C.c_size_t(), C.c_size_t(), vp()
        lq, lt = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        lqo, lto = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        rc = L.bsa_shard_scatter(comm, 0, seqs.ctypes.data, qoff.ctypes.data, qlen.ctypes.data, toff.ctypes.data, tlen.ctypes.data, n, 128,
                                 C.byref(first), C.byref(count), C.byref(dptr), C.byref(nbytes), lq.ctypes.data, lt.ctypes.data, lqo.ctypes.data, lto.ctypes.data, n)
        assert rc == 0 and (first.value, count.value) == (0, n) and np.array_equal(lq, qlen) and np.array_equal(lt, tlen)
        par = B.make_params(S.MODE_GLOBAL, 128, 2, -6, -3, -2, 0, 0)
        # align the shard where it lies (device pointers), then gather
        import torch
        plan = B.AlignPlan(ctx, lqo, lq, lto, lt, par)
        dev = torch.device("cuda", 0)
        d_out = torch.zeros(n * 10, dtype=torch.int32, device=dev); d_cig = torch.zeros(n * 400, dtype=torch.int32, device=dev)
        d_off = torch.zeros(n + 1, dtype=torch.int64, device=dev); d_st = torch.zeros(n, dtype=torch.int32, device=dev)

        class _Blob:                                   # the scatter's device blob as the tensor-like the plan's run() wants
            def data_ptr(self_inner):
                return dptr.value
        plan.run(_Blob(), d_out, d_cig, d_off, d_st)
        torch.cuda.synchronize()
        hoff = d_off.cpu().numpy().astype(np.uint64)
        out = np.zeros((n, 10), np.int32); cig = np.zeros(int(hoff[n]) + 8, np.uint32); ooff = np.zeros(n + 1, np.uint64)
        rc = L.bsa_shard_gather(comm, 0, d_out.data_ptr(), d_cig.data_ptr(), hoff.ctypes.data, n, out.ctypes.data, cig.ctypes.data, cig.size, ooff.ctypes.data, n)
        assert rc == 0 and np.array_equal(ooff, hoff)
        plan.close()
        want, wcig, _ = ctx.align_batch(pairs, par)
        for k in range(n):
            assert np.array_equal(out[k], np.array([want[k][f] for f in want.dtype.names], np.int32))
            assert np.array_equal(cig[int(ooff[k]):int(ooff[k + 1])], wcig[k])
    finally:
        L.bsa_shard_comm_destroy(comm)
