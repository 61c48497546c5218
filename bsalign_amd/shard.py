"""Sharding of independent read pairs across the GPUs of one node.

Pairs are independent (bsalign.h:3854-4050 touches only its arguments), so the batch is cut into contiguous
ranges balanced by band cells (tlen * bw_eff); every rank aligns its own range and the only exchange is the
gather of fixed-size result records (plus CIGAR blobs) at the end.  No collective sits inside the DP.
"""
import numpy as np


def partition_pairs(tlen, bw, world):
    """contiguous ranges [s_r, s_{r+1}) balanced by sum(tlen * bw): returns world+1 boundaries"""
    tlen = np.asarray(tlen, dtype=np.float64)
    w = tlen * float(bw)
    tot = float(w.sum())
    cum = np.concatenate([[0.0], np.cumsum(w)])
    targets = tot * np.arange(1, world, dtype=np.float64) / world
    ks = np.minimum(np.searchsorted(cum, targets, side="left"), len(tlen))
    bounds = [0] + [int(k) for k in np.maximum.accumulate(ks)]
    bounds.append(len(tlen))
    return bounds


def synthetic_first_pair(rank, pairs_per_rank):
    """weak-scaling benchmark: rank r generates pairs [r*n, (r+1)*n) of the global synthetic stream"""
    return rank * pairs_per_rank


def gather_results(local_results, group=None):
    """all ranks contribute their list of per-pair records (any picklable); returns the concatenation in rank order"""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    bucket = [None] * world
    dist.all_gather_object(bucket, local_results, group=group)
    out = []
    for part in bucket:
        out.extend(part)
    return out


# ---- the one exchange each way of SURVEY.md 8(e): scatter the inputs, gather the results ----------------------
# Point-to-point messages of unequal size, posted together (torch.distributed.batch_isend_irecv == one
# ncclGroupStart / ncclGroupEnd around N-1 sends on the source rank: on MI355X that is one message per xGMI link,
# all links busy at once).  Tensors may live on the device (backend "nccl" = RCCL) or on the host (gloo, CPU tests).

def _dev(device):
    import torch
    return torch.device(device) if device is not None else torch.device("cpu")


def _layout(qlen, tlen, a, b):
    """offsets of pairs [a, b) inside their shard blob (target k, then query k, each padded to 16 bytes) and its size"""
    tp = (tlen[a:b].astype(np.uint64) + np.uint64(15)) & ~np.uint64(15)
    qp = (qlen[a:b].astype(np.uint64) + np.uint64(15)) & ~np.uint64(15)
    both = tp + qp
    to = np.zeros(b - a, np.uint64)
    if b - a > 1:
        to[1:] = np.cumsum(both)[:-1]
    return to, to + tp, int(both.sum())


def scatter_batch(batch, bw, src=0, device=None, group=None):
    """rank `src` passes batch = dict(seqs uint8 blob, qoff, qlen, toff, tlen as numpy); the others pass None.
    Every rank gets its contiguous shard back as a dict of the same keys (offsets re-based to the shard's own blob,
    `seqs` a uint8 tensor on `device`) plus `first` (global index of its first pair) and `bounds`.
    Lengths travel as two tensor broadcasts, the shards as one grouped set of point-to-point messages; the source packs
    each shard with the library's C helper (bsa_shard_pack), nothing is done per pair in Python."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from . import lib
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = _dev(device)
    nt = torch.zeros(1, dtype=torch.int64, device=dev)
    if rank == src:
        qoff, qlen = np.ascontiguousarray(batch["qoff"], np.uint64), np.ascontiguousarray(batch["qlen"], np.uint32)
        toff, tlen = np.ascontiguousarray(batch["toff"], np.uint64), np.ascontiguousarray(batch["tlen"], np.uint32)
        nt[0] = len(qlen)
    dist.broadcast(nt, src=src, group=group)
    n = int(nt.item())
    lens = torch.zeros(2, max(n, 1), dtype=torch.int64, device=dev)          # lengths only: 16 bytes per pair
    if rank == src and n:
        lens[0, :n] = torch.from_numpy(qlen.astype(np.int64)).to(dev)
        lens[1, :n] = torch.from_numpy(tlen.astype(np.int64)).to(dev)
    dist.broadcast(lens, src=src, group=group)
    if rank != src:
        lh = lens.cpu().numpy()
        qlen, tlen = lh[0, :n].astype(np.uint32), lh[1, :n].astype(np.uint32)
    bounds = partition_pairs(tlen, bw, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    my_toff, my_qoff, my_bytes = _layout(qlen, tlen, lo, hi)
    mine = torch.zeros(max(my_bytes, 1), dtype=torch.uint8, device=dev)
    ops, keep = [], []
    if rank == src:
        seqs = np.ascontiguousarray(batch["seqs"], np.uint8)
        L = lib()
        L.bsa_shard_pack.argtypes = [C.c_void_p] * 5 + [C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint]
        for r in range(world):
            a, b = bounds[r], bounds[r + 1]
            to, qo, nbytes = _layout(qlen, tlen, a, b)
            blob = np.empty(max(nbytes, 1), np.uint8)
            oq, ot = np.zeros(max(b - a, 1), np.uint64), np.zeros(max(b - a, 1), np.uint64)
            rc = L.bsa_shard_pack(seqs.ctypes.data, qoff.ctypes.data, qlen.ctypes.data, toff.ctypes.data, tlen.ctypes.data, a, b - a,
                                  blob.ctypes.data, nbytes, oq.ctypes.data, ot.ctypes.data, 0)
            assert rc == 0 and (b == a or (np.array_equal(oq[:b - a], qo) and np.array_equal(ot[:b - a], to)))
            t = torch.from_numpy(blob).to(dev)
            if r == src:
                mine.copy_(t[:mine.numel()])
            elif nbytes:
                keep.append(t)
                ops.append(dist.P2POp(dist.isend, t, r, group))
    elif my_bytes:
        ops.append(dist.P2POp(dist.irecv, mine, src, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return dict(seqs=mine, qoff=my_qoff, qlen=qlen[lo:hi].copy(), toff=my_toff, tlen=tlen[lo:hi].copy(), first=lo, bounds=bounds)


def gather_batch(results, cigar, cigar_off, dst=0, group=None):
    """results: [n_r, 10] int32 tensor, cigar: uint32/int32 word tensor, cigar_off: [n_r + 1] int64 offsets into it
    (all of this rank's pairs, in pair order).  Rank `dst` returns (results [n, 10], cigar words, offsets [n + 1]) of the
    whole batch in pair order; the other ranks return None.  Sizes first (one all_gather of two integers per rank),
    then the payloads as grouped point-to-point messages."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = results.device
    n_r = int(results.shape[0])
    nw_r = int(cigar_off[n_r]) if n_r else 0
    sizes = torch.zeros(world, 2, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, torch.tensor([[n_r, nw_r]], dtype=torch.int64, device=dev), group=group)
    sizes = sizes.cpu().numpy()
    counts = torch.from_numpy((np.asarray(cigar_off.cpu().numpy()[1:n_r + 1]) - np.asarray(cigar_off.cpu().numpy()[:n_r])).astype(np.int64)).to(dev) \
        if n_r else torch.zeros(0, dtype=torch.int64, device=dev)
    res32 = results.to(torch.int32).contiguous()
    cig32 = cigar[:nw_r].view(torch.int32).contiguous() if nw_r else torch.zeros(0, dtype=torch.int32, device=dev)
    if rank != dst:
        ops = []
        if n_r:
            ops += [dist.P2POp(dist.isend, res32, dst, group), dist.P2POp(dist.isend, counts, dst, group)]
        if nw_r:
            ops.append(dist.P2POp(dist.isend, cig32, dst, group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return None
    parts_r, parts_c, parts_w, ops = [], [], [], []
    for r in range(world):
        nr, nw = int(sizes[r, 0]), int(sizes[r, 1])
        if r == dst:
            parts_r.append(res32); parts_c.append(counts); parts_w.append(cig32)
            continue
        pr = torch.zeros(nr, 10, dtype=torch.int32, device=dev)
        pc = torch.zeros(nr, dtype=torch.int64, device=dev)
        pw = torch.zeros(nw, dtype=torch.int32, device=dev)
        parts_r.append(pr); parts_c.append(pc); parts_w.append(pw)
        if nr:
            ops += [dist.P2POp(dist.irecv, pr, r, group), dist.P2POp(dist.irecv, pc, r, group)]
        if nw:
            ops.append(dist.P2POp(dist.irecv, pw, r, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    allr = torch.cat(parts_r, dim=0)
    allc = torch.cat(parts_c)
    off = torch.zeros(allc.numel() + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(allc, dim=0)
    return allr, torch.cat(parts_w), off
