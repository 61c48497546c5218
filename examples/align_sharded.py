"""Real-input multi-GPU flow of SURVEY.md 8(e): rank 0 owns a batch of read pairs, scatters contiguous shards over
RCCL (one grouped set of point-to-point messages), every rank aligns its shard on its own MI355X through the C-ABI
(device pointers, no host round trip), and the result records + CIGAR arenas are gathered back on rank 0.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29533 \
        examples/align_sharded.py [--pairs 2000] [--length 3000]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bsalign_amd as B  # noqa: E402
from bsalign_amd import shard  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=2000)
    ap.add_argument("--length", type=int, default=3000)
    ap.add_argument("--bw", type=int, default=128)
    args = ap.parse_args()
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    batch = None
    if rank == 0:
        import support as S
        pairs = [S.synth_pair(k, args.length) for k in range(args.pairs)]
        blob, qoff, qlen, toff, tlen, acc = [], [], [], [], [], 0
        for q, t in pairs:
            toff.append(acc); blob.append(t); acc += len(t); tlen.append(len(t))
            qoff.append(acc); blob.append(q); acc += len(q); qlen.append(len(q))
        batch = dict(seqs=np.concatenate(blob), qoff=qoff, qlen=qlen, toff=toff, tlen=tlen)
    sh = shard.scatter_batch(batch, args.bw, src=0, device=dev)
    n = len(sh["qlen"])
    ctx = B.Context(local)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    par = B.make_params(B.MODE_GLOBAL, args.bw, 2, -6, -3, -2, 0, 0)
    d_out = torch.zeros(max(n, 1) * 10, dtype=torch.int32, device=dev)
    d_cig = torch.zeros(max(n, 1) * max(args.length // 2, 64), dtype=torch.int32, device=dev)
    d_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    d_st = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    if n:
        plan = B.AlignPlan(ctx, sh["qoff"], sh["qlen"], sh["toff"], sh["tlen"], par)
        plan.run(sh["seqs"], d_out, d_cig, d_off, d_st)
        ctx.sync()
        plan.close()
    got = shard.gather_batch(d_out.view(-1, 10)[:n], d_cig, d_off, dst=0)
    flagged = torch.tensor([int((d_st[:n] != 0).sum())], dtype=torch.int64, device=dev)
    dist.all_reduce(flagged)
    if rank == 0:
        import support as S
        res, words, off = got[0].cpu().numpy(), got[1].cpu().numpy().view(np.uint32), got[2].cpu().numpy()
        ok = res.shape[0] == args.pairs
        for k in range(0, args.pairs, max(1, args.pairs // 16)):
            r, cig, _ = S.oracle_align(pairs[k][0], pairs[k][1], 0, args.bw, 2, -6, -3, -2, 0, 0)
            ok &= bool(np.array_equal(res[k], r)) and bool(np.array_equal(words[int(off[k]):int(off[k + 1])], cig))
        print("sharded alignment: %d pairs over %d rank(s), bounds %s, flagged %d, sampled pairs identical to the oracle: %s"
              % (args.pairs, world, sh["bounds"], int(flagged.item()), ok), flush=True)
        if not ok:
            sys.exit(1)
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
