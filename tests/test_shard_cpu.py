"""CPU, world_size 2 over gloo: the N>1 path of the benchmark -- pairs sharded across ranks with no data-path
collective, results gathered once -- gives exactly the single-process result."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import support as S


def _worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, S.ROOT)
    from bsalign_amd import shard
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    rng = np.random.default_rng(9)
    lens = rng.integers(50, 400, size=24)
    pairs = [S.synth_pair(k, int(L)) for k, L in enumerate(lens)]
    tl = [len(t) for _, t in pairs]
    bounds = shard.partition_pairs(tl, 64, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    mine = []
    for k in range(lo, hi):
        res, cig, _ = S.oracle_align(pairs[k][0], pairs[k][1], 0, 64, 2, -6, -3, -2, 0, 0)
        mine.append((k, res.tolist(), cig.tolist()))
    allr = shard.gather_results(mine)
    if rank == 0:
        q.put((bounds, allr))
    dist.barrier()
    dist.destroy_process_group()


def test_partition_is_contiguous_and_balanced():
    sys.path.insert(0, S.ROOT)
    from bsalign_amd import shard
    tl = [100] * 10 + [1000] * 10
    b = shard.partition_pairs(tl, 128, 4)
    assert b[0] == 0 and b[-1] == 20 and all(b[i] <= b[i + 1] for i in range(4))
    w = np.array(tl, dtype=np.float64)
    loads = [w[b[i]:b[i + 1]].sum() for i in range(4)]
    assert max(loads) <= 1.5 * (w.sum() / 4)
    assert shard.partition_pairs([5, 5], 16, 8)[-1] == 2
    assert shard.synthetic_first_pair(3, 100000) == 300000


def test_two_rank_sharding_equals_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    bounds, allr = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [k for k, _, _ in allr] == list(range(24))        # rank order == pair order (contiguous ranges)
    rng = np.random.default_rng(9)
    lens = rng.integers(50, 400, size=24)
    for k, L in enumerate(lens):
        qq, tt = S.synth_pair(k, int(L))
        res, cig, _ = S.oracle_align(qq, tt, 0, 64, 2, -6, -3, -2, 0, 0)
        assert allr[k][1] == res.tolist() and allr[k][2] == cig.tolist()


def _worker_exchange(rank, world, port, q):
    """scatter_batch / gather_batch over gloo: rank 0 owns the batch, every rank aligns its shard (oracle stands in for
    the device here), rank 0 receives everything back in pair order"""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, S.ROOT)
    from bsalign_amd import shard
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    batch = None
    if rank == 0:
        rng = np.random.default_rng(17)
        lens = rng.integers(30, 500, size=21)
        pairs = [S.synth_pair(k, int(L)) for k, L in enumerate(lens)]
        blob, qoff, qlen, toff, tlen = [], [], [], [], []
        acc = 0
        for qq, tt in pairs:
            toff.append(acc); blob.append(tt); acc += len(tt); tlen.append(len(tt))
            qoff.append(acc); blob.append(qq); acc += len(qq); qlen.append(len(qq))
        batch = dict(seqs=np.concatenate(blob), qoff=qoff, qlen=qlen, toff=toff, tlen=tlen)
    sh = shard.scatter_batch(batch, 64, src=0)
    seqs = sh["seqs"].numpy()
    res, words, off = [], [], [0]
    for k in range(len(sh["qlen"])):
        qq = seqs[int(sh["qoff"][k]):int(sh["qoff"][k]) + int(sh["qlen"][k])]
        tt = seqs[int(sh["toff"][k]):int(sh["toff"][k]) + int(sh["tlen"][k])]
        r, cig, _ = S.oracle_align(qq, tt, 0, 64, 2, -6, -3, -2, 0, 0)
        res.append(r); words.append(cig.astype(np.int64)); off.append(off[-1] + len(cig))
    rt = torch.from_numpy(np.array(res, dtype=np.int32).reshape(-1, 10))
    ct = torch.from_numpy(np.concatenate(words).astype(np.int32)) if words else torch.zeros(0, dtype=torch.int32)
    got = shard.gather_batch(rt, ct, torch.tensor(off, dtype=torch.int64), dst=0)
    if rank == 0:
        q.put((sh["bounds"], got[0].numpy(), got[1].numpy().view(np.uint32), got[2].numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_scatter_and_gather_exchange_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_exchange, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    bounds, res, words, off = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert bounds[0] == 0 and bounds[-1] == 21 and 0 < bounds[1] < 21
    rng = np.random.default_rng(17)
    lens = rng.integers(30, 500, size=21)
    assert res.shape == (21, 10) and len(off) == 22
    for k, L in enumerate(lens):
        qq, tt = S.synth_pair(k, int(L))
        r, cig, _ = S.oracle_align(qq, tt, 0, 64, 2, -6, -3, -2, 0, 0)
        assert np.array_equal(res[k], r) and np.array_equal(words[int(off[k]):int(off[k + 1])], cig), k


def _worker_big(rank, world, port, q):
    """100 k pairs through scatter_batch over gloo: the scatter must not do per-pair work in Python (C2's pair count)"""
    import time
    import torch.distributed as dist
    sys.path.insert(0, S.ROOT)
    from bsalign_amd import shard
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    n, batch = 100000, None
    rng = np.random.default_rng(23)
    tlen = rng.integers(40, 90, size=n).astype(np.uint32)
    qlen = rng.integers(40, 90, size=n).astype(np.uint32)
    toff = np.zeros(n, np.uint64); qoff = np.zeros(n, np.uint64)
    both = tlen.astype(np.uint64) + qlen.astype(np.uint64)
    toff[1:] = np.cumsum(both)[:-1]
    qoff[:] = toff + tlen
    if rank == 0:
        seqs = rng.integers(0, 4, size=int(both.sum())).astype(np.uint8)
        batch = dict(seqs=seqs, qoff=qoff, qlen=qlen, toff=toff, tlen=tlen)
    t0 = time.time()
    sh = shard.scatter_batch(batch, 128, src=0)
    secs = time.time() - t0
    lo = sh["first"]
    # spot check: a few pairs of this rank's shard are the right bytes (every rank can regenerate the blob)
    seqs = rng.integers(0, 4, size=int(both.sum())).astype(np.uint8) if rank else batch["seqs"]
    mine = sh["seqs"].numpy()
    ok = True
    for k in (0, 1, len(sh["qlen"]) // 2, len(sh["qlen"]) - 1):
        g = lo + k
        ok &= bool(np.array_equal(mine[int(sh["toff"][k]):int(sh["toff"][k]) + int(tlen[g])], seqs[int(toff[g]):int(toff[g]) + int(tlen[g])]))
        ok &= bool(np.array_equal(mine[int(sh["qoff"][k]):int(sh["qoff"][k]) + int(qlen[g])], seqs[int(qoff[g]):int(qoff[g]) + int(qlen[g])]))
    q.put((rank, secs, ok, sh["bounds"]))
    dist.barrier()
    dist.destroy_process_group()


def test_scatter_of_100k_pairs_is_vectorised():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_big, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, secs, ok, bounds in got:
        assert ok and secs < 5.0, (rank, secs)
        assert bounds[0] == 0 and bounds[-1] == 100000 and 40000 < bounds[1] < 60000


# ---- the C-level exchange (bsa_shard_scatter / bsa_shard_gather, bsalign_amd/csrc/bsa_shard_rccl.hip) at world size 2 and 3 --------------
# The product wire is RCCL over xGMI with device buffers; BSA_SHARD_TRANSPORT=shm swaps in shared memory between processes and host
# buffers (bsa_shard_shm.cpp) under the SAME exchange code, so its rank arithmetic -- ranges, blob offsets, grouped sends / receives, the
# agreement on errors -- runs here, without a GPU.
def _c_api():
    import ctypes as C
    import bsalign_amd as B
    L = B.lib()
    vp = C.c_void_p
    L.bsa_shard_unique_id.argtypes = [vp]
    L.bsa_shard_comm_create.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.bsa_shard_comm_destroy.argtypes = [vp]
    L.bsa_shard_comm_destroy.restype = None
    L.bsa_shard_scatter.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, C.c_size_t, C.c_uint32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(vp),
                                    C.POINTER(C.c_size_t), vp, vp, vp, vp, C.c_size_t]
    L.bsa_shard_gather.argtypes = [vp, C.c_int, vp, vp, vp, C.c_size_t, vp, vp, C.c_size_t, vp, C.c_size_t]
    return L


def _c_batch(n=257, seed=5):
    rng = np.random.default_rng(seed)
    qlen = rng.integers(1, 400, size=n).astype(np.uint32)
    tlen = rng.integers(1, 500, size=n).astype(np.uint32)
    tlen[n // 2:] += 900                                    # uneven cells: the cut is not at n / 2
    off = np.concatenate([[0], np.cumsum(qlen.astype(np.uint64) + tlen)[:-1]]).astype(np.uint64)
    seqs = rng.integers(0, 4, size=int((qlen.astype(np.uint64) + tlen).sum())).astype(np.uint8)
    return seqs, off, qlen, off + qlen, tlen


def _c_batch_large(with_seqs, huge, n=100000, seed=6):
    """C2's pair count with offsets beyond 4 GiB (the 64-bit arithmetic of the exchange): the blob is untouched zero pages but for a marker
    in every 499th pair -- its index in its query's first four bytes, one byte at each sequence's end.  huge: lengths of 21-23 kbp, the
    blob itself (and every rank's shard) passes 4 GiB, two minutes of copying through the shared-memory wire; else lengths a tenth of
    that and a 5 GiB hole in the root's blob before the second half of the pairs (64-bit offsets on the root's side only)"""
    rng = np.random.default_rng(seed)
    lo, hi = (21000, 23000) if huge else (2100, 2300)
    qlen = rng.integers(lo, hi, size=n).astype(np.uint32)
    tlen = rng.integers(lo, hi, size=n).astype(np.uint32)
    tot = qlen.astype(np.uint64) + tlen
    off = np.concatenate([[0], np.cumsum(tot)[:-1]]).astype(np.uint64)
    if not huge:
        off[n // 2:] += np.uint64(5 << 30)
        tot = tot.copy(); tot[-1] += np.uint64(5 << 30)          # (the blob's size below: the last pair's end)
    assert int(off[-1]) > (1 << 32)
    seqs = None
    if with_seqs:
        seqs = np.zeros(int(off[-1]) + int(qlen[-1]) + int(tlen[-1]), np.uint8)
        for g in range(0, n, 499):
            o = int(off[g])
            seqs[o:o + 4] = np.frombuffer(np.uint32(g).tobytes(), np.uint8)
            seqs[o + int(qlen[g]) - 1] = 1 + g % 3
            seqs[o + int(qlen[g]) + int(tlen[g]) - 1] = 1 + g % 2
    return seqs, off, qlen, off + qlen, tlen


def _fake_results(first, count, qlen, tlen):
    """what a rank would hand to the gather: a result record and a few CIGAR words per pair, functions of the pair's global index"""
    out = np.zeros((count, 10), np.int32)
    words, offs = [], [0]
    for i in range(count):
        g = first + i
        out[i] = [g, int(qlen[g]), int(tlen[g]), 0, 0, g % 7, 1, 2, 3, 4]
        k = 1 + g % 5
        words.extend([(g << 4) | j for j in range(k)])
        offs.append(len(words))
    return out, np.array(words, np.uint32), np.array(offs, np.uint64)


def _worker_c_exchange(rank, world, idfile, q, scenario):
    try:
        import ctypes as C
        import time
        os.environ["BSA_SHARD_TRANSPORT"] = "shm"
        L = _c_api()
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            assert L.bsa_shard_unique_id(ident) == 0
            with open(idfile + ".tmp", "wb") as f:
                f.write(bytes(ident))
            os.replace(idfile + ".tmp", idfile)
        else:
            for _ in range(6000):
                if os.path.exists(idfile):
                    break
                time.sleep(0.01)
            ident = (C.c_uint8 * 128).from_buffer_copy(open(idfile, "rb").read())
        comm = C.c_void_p()
        assert L.bsa_shard_comm_create(None, rank, world, ident, C.byref(comm)) == 0
        root = 1 if scenario == "root1" else 0
        large = scenario in ("large", "huge")
        seqs, qoff, qlen, toff, tlen = _c_batch_large(rank == root, scenario == "huge") if large else _c_batch()
        n = len(qlen)
        first, count, nbytes, blob = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_void_p()
        cap = 3 if (scenario == "small_cap" and rank == 1) else n
        lq, lt, lqo, lto = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        src = (seqs, qoff, qlen, toff, tlen) if rank == root else (None,) * 5
        ptr = lambda a: a.ctypes.data if a is not None else None
        rc = L.bsa_shard_scatter(comm, root, ptr(src[0]), ptr(src[1]), ptr(src[2]), ptr(src[3]), ptr(src[4]), n if rank == root else 0, 128,
                                 C.byref(first), C.byref(count), C.byref(blob), C.byref(nbytes), lq.ctypes.data, lt.ctypes.data, lqo.ctypes.data, lto.ctypes.data, cap)
        if scenario == "small_cap":
            q.put((rank, "scatter_rc", rc))                 # every rank must come back with the error of the rank that cannot take its shard
            L.bsa_shard_comm_destroy(comm)
            return
        assert rc == 0, rc
        f0, cn = first.value, count.value
        # the shard blob: every pair's bytes at the offsets the scatter reported
        raw = (C.c_uint8 * nbytes.value).from_address(blob.value)
        got = np.frombuffer(raw, np.uint8)
        ok = bool(np.array_equal(lq[:cn], qlen[f0:f0 + cn]) and np.array_equal(lt[:cn], tlen[f0:f0 + cn]))
        if large:
            # the marked pairs at the offsets the scatter reported, the blob's size, nothing but zeros in between (a sum over the whole shard)
            marks = 0
            for g in range(((f0 + 498) // 499) * 499, f0 + cn, 499):
                i = g - f0
                qo, to = int(lqo[i]), int(lto[i])
                ok &= bool(int(np.frombuffer(got[qo:qo + 4].tobytes(), np.uint32)[0]) == g and got[qo + int(qlen[g]) - 1] == 1 + g % 3 and got[to + int(tlen[g]) - 1] == 1 + g % 2)
                marks += sum(int(b) for b in np.frombuffer(np.uint32(g).tobytes(), np.uint8)) + (1 + g % 3 if int(qlen[g]) > 4 else 0) + 1 + g % 2
            ok &= bool(int(got.sum(dtype=np.uint64)) == marks)
            ok &= bool(nbytes.value >= int((qlen[f0:f0 + cn].astype(np.uint64) + tlen[f0:f0 + cn]).sum()))
            ok &= bool(int(lqo[cn - 1]) + int(qlen[f0 + cn - 1]) <= nbytes.value and int(lto[cn - 1]) + int(tlen[f0 + cn - 1]) <= nbytes.value)
        for i in range(0 if not large else cn, cn):
            g = f0 + i
            ok &= bool(np.array_equal(got[int(lqo[i]):int(lqo[i]) + int(qlen[g])], seqs[int(qoff[g]):int(qoff[g]) + int(qlen[g])]))
            ok &= bool(np.array_equal(got[int(lto[i]):int(lto[i]) + int(tlen[g])], seqs[int(toff[g]):int(toff[g]) + int(tlen[g])]))
        res, words, offs = _fake_results(f0, cn, qlen, tlen)
        out = np.zeros((n, 10), np.int32); cig = np.zeros(8 * n, np.uint32); ooff = np.zeros(n + 1, np.uint64)
        capw = 4 if (scenario == "small_arena" and rank == root) else cig.size
        rc = L.bsa_shard_gather(comm, root, res.ctypes.data, words.ctypes.data, offs.ctypes.data, cn, out.ctypes.data, cig.ctypes.data, capw, ooff.ctypes.data, n)
        if scenario == "small_arena":
            q.put((rank, "gather_rc", rc))
            L.bsa_shard_comm_destroy(comm)
            return
        assert rc == 0, rc
        if rank == root:
            wres, wwords, woffs = _fake_results(0, n, qlen, tlen)
            ok &= bool(np.array_equal(out, wres) and np.array_equal(ooff, woffs) and np.array_equal(cig[:len(wwords)], wwords))
        L.bsa_shard_comm_destroy(comm)
        q.put((rank, "ok", (ok, f0, cn)))
    except Exception as ex:          # pragma: no cover
        import traceback
        q.put((rank, "error", traceback.format_exc() + str(ex)))


def _run_c_exchange(world, scenario, tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    idfile = str(tmp_path / ("id_%s_%d" % (scenario, world)))
    procs = [ctx.Process(target=_worker_c_exchange, args=(r, world, idfile, q, scenario)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    return sorted(got)


@pytest.mark.parametrize("world,scenario", [(2, "plain"), (2, "root1"), (3, "plain")])
def test_c_level_exchange_over_shared_memory(world, scenario, tmp_path):
    got = _run_c_exchange(world, scenario, tmp_path)
    assert all(kind == "ok" and val[0] for _, kind, val in got), got
    ranges = sorted((val[1], val[2]) for _, _, val in got)
    assert ranges[0][0] == 0 and all(ranges[i][0] + ranges[i][1] == ranges[i + 1][0] for i in range(world - 1)) and ranges[-1][0] + ranges[-1][1] == 257
    assert all(c > 0 for _, c in ranges)                     # (balanced by cells: nobody is empty)


def test_c_level_exchange_agrees_on_errors(tmp_path):
    """a rank whose arrays cannot take its shard, a root whose arena is too small: EVERY rank returns the error, nobody is left waiting for a
    message (the calls used to return on one side before the matching send / receive was posted)"""
    got = _run_c_exchange(2, "small_cap", tmp_path)                     # BSA_E_NOMEM = -3, BSA_E_CIGAR_CAP = -5 (include/bsalign_hip.h)
    assert [(k, v) for _, k, v in got] == [("scatter_rc", -3)] * 2, got
    got = _run_c_exchange(2, "small_arena", tmp_path)
    assert [(k, v) for _, k, v in got] == [("gather_rc", -5)] * 2, got
