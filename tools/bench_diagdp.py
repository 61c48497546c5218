"""Anti-diagonal u8 DP of the MSA refinement (bsa_diagdp_batch): device kernel time against the reference's SSE code on one
core, for 1 .. N windows of 64 reads over a 22 k-column MSA (the refinement of C4 windows, band 32).  Run through gpurun:
    python tools/bench_diagdp.py [max_windows]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bsalign_amd as B
import diag_support as D
import support as S

maxw = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(4)
planes1, probs1 = D.make_window(rng, 22000, 64, 2, 0.12, False)
ctx = B.Context(0)
t_ref = None
if S.have_ref():
    nb = D.matrix_layout([dict(p) for p in probs1])
    pr = [dict(p) for p in probs1]
    D.matrix_layout(pr)
    t0 = time.time()
    D.ref_fill(planes1, pr, nb)
    t_ref = time.time() - t0
w = 1
while w <= maxw:
    probs = []
    for k in range(w):
        for p in probs1:
            q = dict(p)
            off = k * planes1.size
            q["seq0"] += off; q["seq1"] += off
            q["mats0"] = [o + off for o in p["mats0"]]; q["mats1"] = [o + off for o in p["mats1"]]
            probs.append(q)
    planes = np.tile(planes1, w)
    nbytes = D.matrix_layout(probs)
    st = D.to_struct(probs)
    ctx.diagdp_batch(planes, st, nbytes)
    t0 = time.time()
    ctx.diagdp_batch(planes, st, nbytes)
    wall = time.time() - t0
    ms = ctx.diagdp_last_ms()
    steps = sum(2 * (p["mend"] - p["mbeg"]) - 1 for p in probs)
    cells = steps * 32
    print(json.dumps({"windows": w, "reads": len(probs), "steps": steps, "device_kernel_ms": round(ms, 3), "call_wall_ms": round(wall * 1e3, 1),
                      "gcups_kernel": round(cells / ms / 1e6, 2), "bytes_written_GBps": round(2 * (32 + 2) / 32 * cells / ms / 1e6, 1),
                      "ref_one_core_ms_per_window": None if t_ref is None else round(t_ref * 1e3, 1),
                      "speedup_vs_one_core": None if t_ref is None else round(t_ref * 1e3 * w / ms, 1)}), flush=True)
    w *= 4
