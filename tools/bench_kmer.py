"""Timing of the k-mer anchored edit alignment on synthetic read pairs (host chaining + device segments).
    python tools/bench_kmer.py [pairs] [length] [ksz]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
import bsalign_amd as B

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
ksz = int(sys.argv[3]) if len(sys.argv) > 3 else 13
pairs = B.synth_pairs_host(n, L)
ctx = B.Context(0)
ctx.kmer_edit_batch(pairs[:64], ksz=ksz)
for threads in (0, 0, 1):
    t0 = time.time()
    out, cigs, st = ctx.kmer_edit_batch(pairs, ksz=ksz, threads=threads)
    dt = time.time() - t0
    print("kmer edit: %d pairs x %d bp, ksz %d, threads %d: %.3f s  (%.0f pairs/s, %.2f Mbp/s of query)" % (n, L, ksz, threads, dt, n / dt, sum(len(p[0]) for p in pairs) / dt / 1e6))
if len(sys.argv) > 4 and sys.argv[4] == "cpu":
    # the reference's kmer_striped_seqedit_pairwise (oracle/_ref) on one host core, a sample of the same pairs
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    import kmer_support as K
    import support as S
    if S.have_ref():
        m = min(n, 400)
        t0 = time.time()
        same = 0
        for k in range(m):
            r, c = K.ref_kmer_edit(ksz, pairs[k][0], pairs[k][1])
            same += int(np.array_equal(np.array(out[k].tolist(), dtype=np.int32), r) and np.array_equal(cigs[k], c))
        dt = time.time() - t0
        print("reference on one core: %d pairs in %.2f s = %.0f pairs/s; identical to the device results: %d of %d" % (m, dt, m / dt, same, m))
if len(sys.argv) > 4:
    sys.exit(0)
t0 = time.time()
out2, _, _ = ctx.edit_batch(pairs, mode=B.MODE_GLOBAL, bandwidth=0)
dt = time.time() - t0
print("plain global edit of the same pairs: %.3f s; mean score kmer %.1f vs exact %.1f" % (dt, out["score"].mean(), out2["score"].mean()))
