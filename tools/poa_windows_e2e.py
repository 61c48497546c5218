"""End to end, many POA windows (12 reads x 1.5 kbp each): the reference's end_bspoa on 16 host threads against the batcher paths -- the library's own graph
surface (harness mode 10) and the round-4 binding on the reference's graph (mode 7) -- identical consensus / MSA required.
    python tools/poa_windows_e2e.py [windows]        (GPU box, needs oracle/_ref)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import bsalign_amd as B  # noqa: E402
import poa_support as P  # noqa: E402
from test_poa_batched_gpu import Batcher, _compare  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
p = P.par()
windows = [P.synth_reads(7000 + w, 1500, 12, eps=(0.1,)) for w in range(n)]
P.attach_product(P.ref_poa())
ctx = B.Context(0)
ref, t_ref = P.run_many(windows, 0, p, threads=16)
c_ref = P.run_many.last_cpu_seconds
print("%d windows: reference end_bspoa on 16 host threads %.2f s (%.1f CPU-seconds)" % (n, t_ref, c_ref), flush=True)
for mode in (10, 7, 10):
    bt = Batcher(ctx, n)
    try:
        dev, t = P.run_many(windows, mode, p)
        c = P.run_many.last_cpu_seconds
        st = bt.stats()
    finally:
        bt.close()
    _compare(ref, dev)
    if mode == 10:
        r = P.ref_poa()
        # (the handles are gone after run_many; the seconds were summed there)
        ps = getattr(P.run_many, "last_pog_seconds", None)
        if ps is not None:
            print("        binding, summed over the windows: keeping the library's graph %.2f s, guide alignment + columns %.2f s, inside the library %.2f s "
                  "[select %.2f, place %.2f, program %.2f, waiting for the device %.2f, surgery %.2f], reference-side surgery %.2f s" % (ps[0], ps[1], ps[2], ps[4], ps[5], ps[6], ps[7], ps[8], ps[3]))
    print("    %s: %.2f s (%.1f CPU-seconds; %d batches, %d launches, %.0f MB up, %.0f MB down, device %.2f s), identical results"
          % ("the library's own graph (bsa_pog_*)" if mode == 10 else "round-4 binding on the reference's graph", t, c, st["batches"], st["launches"], st["bytes_up"] / 1e6, st["bytes_down"] / 1e6, st["device_us"] / 1e6), flush=True)
ctx.close()
