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
    bounds = [0]
    for r in range(1, world):
        target = tot * r / world
        k = int(np.searchsorted(cum, target, side="left"))
        k = min(max(k, bounds[-1]), len(tlen))
        bounds.append(k)
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
