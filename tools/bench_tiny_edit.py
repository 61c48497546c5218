"""Device edit path on very many very short pairs (the gap segments of the k-mer anchored alignment)."""
import sys
import time
import ctypes as C

import numpy as np

sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
import bsalign_amd as B

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500000
rng = np.random.default_rng(1)
ql = rng.integers(1, 12, n).astype(np.uint32)
tl = rng.integers(1, 12, n).astype(np.uint32)
qoff = np.zeros(n, dtype=np.uint64)
toff = np.zeros(n, dtype=np.uint64)
tot = np.cumsum(ql.astype(np.uint64) + tl)
qoff[1:] = tot[:-1]
toff[:] = qoff + ql
seqs = rng.integers(0, 4, int(tot[-1])).astype(np.uint8)
ctx = B.Context(0)
out = np.zeros(n, dtype=B.RESULT_DTYPE)
cap = int(tot[-1]) + 2 * n + 16
cig = np.zeros(cap, dtype=np.uint32)
off = np.zeros(n + 1, dtype=np.uint64)
st = np.zeros(n, dtype=np.uint32)
par = B.EditParams()
par.mode, par.bandwidth = 0, 0
for rep in range(2):
    t0 = time.time()
    rc = B.lib().bsa_edit_batch(ctx.h, B._p(seqs), seqs.size, B._p(qoff), B._p(ql), B._p(toff), B._p(tl), n, C.byref(par),
                                B._p(out), B._p(cig), cap, B._p(off), B._p(st))
    print("bsa_edit_batch rc %d: %d tiny pairs in %.3f s" % (rc, n, time.time() - t0))
