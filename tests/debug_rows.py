"""debug tool (not a test): run a few pairs on the GPU, fetch the stored DP rows and report the first
row / field that differs from the oracle's rows.  usage: python tests/debug_rows.py [mode bw L npairs]"""
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
        rows = plan.debug_rows(k)
        rowb = rows.shape[1]
        orows = np.zeros(rows.shape, dtype=np.uint8)
        res = np.zeros(10, dtype=np.int32)
        m = S.score_matrix(sc[0], sc[1])
        lib.orc_align_pairwise_rows(S.ptr(q, S.u8p), len(q), S.ptr(t, S.u8p), len(t), mode, bw, S.ptr(m, S.i8p),
                                    sc[2], sc[3], sc[4], sc[5], S.ptr(res, S.i32p), orows.ctypes.data_as(C.c_void_p), rowb)
        pw = lib.orc_get_piecewise(sc[2], sc[3], sc[4], sc[5], bw)
        nb = (pw + 1) * bw
        used = nb + 72
        print("pair", k, "qlen", len(q), "tlen", len(t), "gpu", out[k], "orc", res)
        for r in range(rows.shape[0]):
            if not np.array_equal(rows[r, :used], orows[r, :used]):
                g, o = rows[r], orows[r]
                print("  first differing row:", r - 1)
                for name, lo, hi in (("u", 0, bw), ("e", bw, 2 * bw if pw >= 1 else bw), ("q", 2 * bw, 3 * bw if pw == 2 else 2 * bw)):
                    if hi > lo and not np.array_equal(g[lo:hi], o[lo:hi]):
                        idx = np.nonzero(g[lo:hi] != o[lo:hi])[0]
                        print("   ", name, "differs at band pos", idx[:16], "gpu", g[lo:hi].view(np.int8)[idx[:16]], "orc", o[lo:hi].view(np.int8)[idx[:16]])
                gu, ou = g[nb:nb + 72].view(np.int32), o[nb:nb + 72].view(np.int32)
                if not np.array_equal(gu, ou):
                    print("    ubegs/rbeg gpu", gu)
                    print("    ubegs/rbeg orc", ou)
                break
        else:
            print("  all rows equal")


if __name__ == "__main__":
    main()
