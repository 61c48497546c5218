"""debug tool (not a test): run a few pairs on the GPU, fetch the stored DP rows and report the first
row / field that differs from the oracle's rows.  usage: python tools/debug_rows.py [mode bw L npairs]"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import bsalign_amd as B
import support as S


def main():
    mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    bw = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    npairs = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    sc = tuple(int(x) for x in sys.argv[5:11]) if len(sys.argv) > 10 else (2, -6, -3, -2, 0, 0)
    pairs = [S.synth_pair(k, L) for k in range(npairs)]
    ctx = B.Context(0)
    par = B.make_params(mode, bw, *sc)
    seqs, qoff, qlen, toff, tlen = B.pack_pairs(pairs)
    plan = B.AlignPlan(ctx, qoff, qlen, toff, tlen, par)
    dev = torch.device("cuda:0")
    d_seqs = torch.from_numpy(seqs).to(dev)
    d_out = torch.zeros(npairs * 10, dtype=torch.int32, device=dev)
    d_cig = torch.zeros(int(qlen.sum() + tlen.sum()) + 64, dtype=torch.int32, device=dev)
    d_off = torch.zeros(npairs + 1, dtype=torch.int64, device=dev)
    d_st = torch.zeros(npairs, dtype=torch.int32, device=dev)
    plan.run(d_seqs, d_out, d_cig, d_off, d_st)
    ctx.sync()
    out = d_out.cpu().numpy().reshape(npairs, 10)
    print("status", d_st.cpu().numpy())
    lib = S.oracle()
    lib.orc_align_pairwise_rows.restype = C.c_long
    for k, (q, t) in enumerate(pairs):
        tl = len(t)
        pw = lib.orc_get_piecewise(sc[2], sc[3], sc[4], sc[5], bw)
        W = bw // 16
        cells = ((pw + 1) * W + 3) & ~3
        blk = cells + 4
        tg = max(64 // blk, 1)
        tileb = (tg * blk + 15) & ~15
        begs_bytes = ((tl + 2) * 4 + 255) & ~255
        ngroups = (tl + 1 + tg - 1) // tg + 1
        nbytes = begs_bytes + ngroups * 16 * tileb
        slot = plan.debug_slot(k, nbytes)
        oslot = np.zeros(nbytes, dtype=np.uint8)
        res = np.zeros(10, dtype=np.int32)
        m = S.score_matrix(sc[0], sc[1])
        lib.orc_align_pairwise_rows(S.ptr(q, S.u8p), len(q), S.ptr(t, S.u8p), tl, mode, bw, S.ptr(m, S.i8p),
                                    sc[2], sc[3], sc[4], sc[5], S.ptr(res, S.i32p), oslot.ctypes.data_as(C.c_void_p), 0)
        print("pair", k, "qlen", len(q), "tlen", tl, "gpu", out[k], "orc", res)
        gb, ob = slot[:(tl + 1) * 4].view(np.int32), oslot[:(tl + 1) * 4].view(np.int32)
        if not np.array_equal(gb, ob):
            r = int(np.nonzero(gb != ob)[0][0])
            print("  band offsets first differ at row", r - 1, "gpu", gb[r:r + 4], "orc", ob[r:r + 4])
        for r in range(tl + 1):
            diff = False
            for y in range(16):
                o0 = begs_bytes + ((r // tg) * 16 + y) * tileb + (r % tg) * blk
                gy, oy = slot[o0:o0 + blk], oslot[o0:o0 + blk]
                if not np.array_equal(gy, oy):
                    if not diff:
                        print("  first differing row:", r - 1)
                    diff = True
                    print("    block", y, "gpu u", gy[:W].view(np.int8), "e", gy[W:2 * W].view(np.int8) if pw else "", "ub", gy[cells:cells + 4].view(np.int32))
                    print("    block", y, "orc u", oy[:W].view(np.int8), "e", oy[W:2 * W].view(np.int8) if pw else "", "ub", oy[cells:cells + 4].view(np.int32))
            if diff:
                break
        else:
            print("  all rows equal")


if __name__ == "__main__":
    main()
