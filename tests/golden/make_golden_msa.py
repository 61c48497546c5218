"""Generates tests/golden/msa_formats.npz from the REAL reference (oracle/_ref): finished POA windows with their MSA
arrays and the bytes dump_binary_msa_bspoa / print_msa_bspoa produce for them.  Run in the build container:
    python tests/golden/make_golden_msa.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import msa_support as M  # noqa: E402
import poa_support as P  # noqa: E402


def main():
    out = {}
    meta = []
    for ci, (seed, L, n, eps) in enumerate(M.CASES[:3]):
        w = M.RefWindow(P.synth_reads(seed, L, n, eps))
        out["cols%d" % ci], out["idxs%d" % ci], out["var%d" % ci] = w.cols, w.idxs, w.var
        out["cns%d" % ci], out["qlt%d" % ci], out["alt%d" % ci] = w.cns, w.qlt, w.alt
        out["bin%d" % ci] = np.frombuffer(w.binary(b"window %d" % ci), dtype=np.uint8)
        for ti, (mb, me, lw) in enumerate(M.TEXT_ARGS):
            if mb >= w.mlen:
                continue
            out["txt%d_%d" % (ci, ti)] = np.frombuffer(w.text("W%d" % ci, mb, me, lw), dtype=np.uint8)
        meta.append((w.nseq, w.mlen))
        w.close()
    out["meta"] = np.array(meta, dtype=np.uint32)
    np.savez_compressed(os.path.join(HERE, "msa_formats.npz"), **out)
    print("wrote msa_formats.npz:", meta)


if __name__ == "__main__":
    main()
