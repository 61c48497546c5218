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
