"""debug tool (not a test): run a few pairs through the compact path with both forward kernels (BSA_ALIGN8_FWD=pk, the
saturating packed kernel, and the exact-arithmetic one) and report the first band offset / code row that differs.
usage: python tools/debug_codes.py [mode bw L npairs [M X O E Q P]]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import bsalign_amd as B
import support as S


def run(pairs, par, which, W):
    os.environ["BSA_ALIGN8_FWD"] = which
    ctx = B.Context(0)
    n = len(pairs)
    seqs, qoff, qlen, toff, tlen = B.pack_pairs(pairs)
    plan = B.AlignPlan(ctx, qoff, qlen, toff, tlen, par)
    dev = torch.device("cuda:0")
    d_seqs = torch.from_numpy(seqs).to(dev)
    d_out = torch.zeros(n * 10, dtype=torch.int32, device=dev)
    d_cig = torch.zeros(int(qlen.sum() + tlen.sum()) + 64, dtype=torch.int32, device=dev)
    d_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    plan.run(d_seqs, d_out, d_cig, d_off, d_st)
    ctx.sync()
    CW = max(W // 8, 1)
    slots = []
    for k in range(n):
        tl = int(tlen[k])
        nb = (((tl + 2) * 4 + 255) & ~255) + ((tl + 3) // 4 * 4) * 64 * CW
        slots.append(plan.debug_slot(k, nb).copy())
    return d_out.cpu().numpy().reshape(n, 10), d_st.cpu().numpy(), slots


def planes(words, W):
    if W == 8:
        w = int(words[0])
        return [w & 0xff, (w >> 8) & 0xff, (w >> 16) & 0xff, w >> 24]
    if W == 4:
        w = int(words[0])
        return [w & 0xf, (w >> 4) & 0xf, (w >> 8) & 0xf, (w >> 12) & 0xf]
    w0, w1 = int(words[0]), int(words[1])
    return [w0 & 0xffff, w0 >> 16, w1 & 0xffff, w1 >> 16]


def main():
    mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    bw = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    npairs = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    sc = tuple(int(x) for x in sys.argv[5:11]) if len(sys.argv) > 10 else (2, -6, -3, -2, 0, 0)
    W = bw // 16
    CW = max(W // 8, 1)
    pairs = [S.synth_pair(k, L) for k in range(npairs)]
    par = B.make_params(mode, bw, *sc)
    o1, s1, a = run(pairs, par, "pk", W)
    o2, s2, b = run(pairs, par, "x", W)
    for k, (q, t) in enumerate(pairs):
        tl = len(t)
        print("pair", k, "qlen", len(q), "tlen", tl, "pk", o1[k], s1[k], "x", o2[k], s2[k])
        ga, gb = a[k][:(tl + 2) * 4].view(np.int32), b[k][:(tl + 2) * 4].view(np.int32)
        if not np.array_equal(ga, gb):
            r = int(np.nonzero(ga != gb)[0][0])
            print("  begs first differ at index", r, "(row", r - 1, ") pk", ga[r:r + 4], "x", gb[r:r + 4], "last (score) pk", ga[tl + 1], "x", gb[tl + 1])
        bb = ((tl + 2) * 4 + 255) & ~255
        # tiled rows (bsa_common.h): groups of four rows, inside a group [block][row % 4][CW dwords]
        def untile(x):
            g = x[bb:].view(np.uint32).reshape(-1, 16, 4, CW)
            return g.transpose(0, 2, 1, 3).reshape(-1, 16, CW)[:tl]
        ra, rb = untile(a[k]), untile(b[k])
        bad = np.nonzero((ra != rb).any(axis=(1, 2)))[0]
        if len(bad) == 0:
            print("  all code rows equal")
            continue
        r = int(bad[0])
        print("  %d rows differ, first row %d (band offset %d, previous %d)" % (len(bad), r, ga[r + 1], ga[r]))
        for y in range(16):
            if not np.array_equal(ra[r, y], rb[r, y]):
                pa, pb = planes(ra[r, y], W), planes(rb[r, y], W)
                print("    block %2d  pk M %s D %s R %s O %s" % ((y,) + tuple(format(v, "0%db" % W) for v in pa)))
                print("              x  M %s D %s R %s O %s" % tuple(format(v, "0%db" % W) for v in pb))


if __name__ == "__main__":
    main()
