"""CPU, world_size 2 over gloo: the N>1 path of the benchmark -- pairs sharded across ranks with no data-path
collective, results gathered once -- gives exactly the single-process result."""
import os
import sys

import numpy as np
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
