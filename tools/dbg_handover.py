"""hand-over counts of the checked whole-query kernel per (scoring, mode) on the pairs of tests/test_align8_gpu.py (debug aid)"""
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import support as S, bsalign_amd as B
from test_align8_gpu import _mk_pairs, BIG_SCORINGS
ctx = B.Context(0)
for bw in (0, 1008):
    rng = np.random.default_rng(9100 + bw)
    top = bw if bw else 3000
    lens = [l for l in (257, 272, 273, 300, 320, 500, 511, 512, 513, 700, 1000, 1008, 1500, 2047, 2048, 2049, 3000) if l <= top]
    pairs = [(q[:top] if len(q) > top else q, t) for q, t in _mk_pairs(rng, 60, lens, eps_list=(0.0, 0.05, 0.2, 0.4), ratios=(1.0, 1.0, 0.5, 0.9, 1.1))]
    pairs = [(q, t) for q, t in pairs if len(q) > 256 or bw]
    row = []
    for scname, sc in BIG_SCORINGS.items():
        for mode in (0, 1, 2):
            out, cigs, st = ctx.align_batch(pairs, B.make_params(mode, bw, *sc))
            row.append(ctx.last_handover())
    print("bw", bw, "pairs", len(pairs), "handover per (scoring, mode):", row, flush=True)
