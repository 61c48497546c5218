"""Randomised shadow campaign for the graph-form POA kernel (run on the GPU box: gpurun -- python tools/campaign_poa_shadow_gpu.py SEED N):
inside real end_bspoa runs of the reference (oracle/_ref, harness mode 5), read after read, the device's best end cell and every step of
its walk are compared with what the reference's own align_rd_bspoacore + alignment2graph_bspoa do on the same graph.  Parameter sets are
drawn from modes x gap models x bandwidths x read lengths / error rates.  BSA_POA_FWD=wf runs the wavefront forward pass instead,
BSA_POA_FORCE_GEN=1 sends every read through the generic-width kernel (bsa_poa_gen.hip)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bsalign_amd as B  # noqa: E402
import poa_support as P  # noqa: E402
import test_poa_graph_gpu as T  # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    rng = np.random.default_rng(seed)
    ctx = B.Context(0)
    lib = P.ref_poa_trace()
    T._attach(lib, ctx)
    tot_reads = tot_steps = bad = declined = 0
    for k in range(n):
        kw = dict(alnmode=int(rng.integers(3)), bandwidth=int(rng.choice([32, 64, 96, 128, 128, 128, 192, 256, 0, 320, 512])))      # (0 / above 256: every read through the generic-width kernel, as a window's first read always is)
        gaps = int(rng.integers(3))
        if gaps == 0:
            kw.update(O=0, E=-2, Q=0, P=0)
        elif gaps == 1:
            kw.update(Q=0, P=0)
        L = int(rng.choice([300, 800, 1500, 3000, 6000]))
        nreads = int(rng.integers(6, 20))
        eps = tuple(float(x) for x in rng.choice([0.02, 0.05, 0.1, 0.15, 0.2], size=3))
        p = P.par(**kw)
        reads = P.synth_reads(int(rng.integers(1 << 30)), L, nreads, eps=eps)
        r = P.run_ref_graph(reads, 5, p, record=True, lib=lib, backend="device")
        steps = sum(len(rc["trace"]) for rc in r["recs"] if "trace" in rc)
        tot_reads += len(reads); tot_steps += steps; bad += r["bad"]; declined += len(reads) - r["graph_reads"]
        print("case %d %s gaps %d-piece, %d reads x %d bp eps %s: reads through the graph form %d, walk steps %d, mismatches %d" % (k, kw, gaps, nreads, L, eps, r["graph_reads"], steps, r["bad"]), flush=True)
    print("TOTAL reads %d (declined by the kernel's guard or the first read of a window: %d), walk steps %d, mismatches %d" % (tot_reads, declined, tot_steps, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
