"""Randomised campaign of the library's own POA graph surface (include/bsalign_poa.h) in the shadow of the reference (harness mode 8, CPU: the oracle's
scalar statement of the kernel between the steps): per read the selection list, band placement, auxiliary edges, program bytes, result and the WHOLE graph
after the surgery are compared with the reference's (oracle/ref_poa_harness.c: poa_align_read_shadow_pog).  Parameters are drawn per window: mode,
gap model, bandwidth, nrec, seqcore, shuffle, bwtrigger, read count / length / divergence.

    python tools/campaign_pog_shadow_cpu.py [windows] [seed]        (needs oracle/_ref: build container only)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import poa_support as P  # noqa: E402


def main():
    nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20250930
    rng = np.random.default_rng(seed)
    tot = dict(windows=0, reads=0, declined=0, imports=0, sel=0, prog=0, steps=0, gnodes=0, gedges=0, bad=0)
    t0 = time.time()
    for w in range(nwin):
        gap = int(rng.integers(0, 3))
        kw = dict(alnmode=int(rng.integers(0, 3)), bandwidth=int(rng.choice([32, 64, 128, 128, 256])), nrec=int(rng.choice([0, 1, 2, 5, 20])),
                  seqcore=int(rng.choice([4, 8, 40])), shuffle=int(rng.integers(0, 2)), bwtrigger=int(rng.random() < 0.85))
        if gap == 0:
            kw.update(O=0, E=-int(rng.integers(2, 5)), Q=0, P=0)
        elif gap == 1:
            kw.update(O=-int(rng.integers(1, 5)), E=-int(rng.integers(1, 3)), Q=0, P=0)
        if rng.random() < 0.15:
            kw.update(bandwidth=0, bwtrigger=0)
        p = P.par(**kw)
        nreads = int(rng.integers(3, 14))
        L = int(rng.integers(120, 250)) if p["bandwidth"] == 0 else int(rng.integers(200, 1600))
        eps = tuple(float(x) for x in rng.uniform(0.02, 0.22, 3))
        reads = P.synth_reads(int(rng.integers(1, 1 << 30)), L, nreads, eps=eps)
        # round 6: a third of the windows re-align a stretch of every read (the realn entry, bsa_pog_cut), a quarter run in refmode (read 0 the reference,
        # half of those with SAM CIGARs -- clips, partial reads -- placing the bands)
        realn_pass = int(rng.choice([0, 0, 1, 2])) if (p["bandwidth"] or L < 250) else 0
        refmode = int(rng.random() < 0.25)
        cigs = None
        if refmode:
            import support as S
            import test_poa_pog_cpu as TT
            T = rng.integers(0, 4, size=L).astype(np.uint8)
            reads = [T] + [np.asarray(S.mutate(rng, T, float(rng.choice(eps))), dtype=np.uint8) for _ in range(nreads)]
            if rng.random() < 0.5:
                reads, cigs = TT._sam_cigars(reads, rng)
            if realn_pass == 1 and L > 1000:
                realn_pass = 2                  # (the oracle's scalar kernel takes bands up to 256 columns here; a middle half of 500+ bases is the device's)
        elif realn_pass == 1 and L > 500:
            realn_pass = 2
        tot["realn"] = tot.get("realn", 0) + (1 if realn_pass else 0); tot["refmode"] = tot.get("refmode", 0) + refmode; tot["sam"] = tot.get("sam", 0) + (1 if cigs is not None else 0)
        r = P.run_ref_graph(reads, 8, p, record=False, refmode=refmode, cigars=cigs, realn_pass=realn_pass)
        g = r["pog"]
        tot["windows"] += 1; tot["reads"] += g["reads"]; tot["declined"] += g["declined"]; tot["imports"] += g["imports"]
        tot["sel"] += g["sel_nodes"]; tot["prog"] += g["program_bytes"]; tot["steps"] += g["steps"]; tot["gnodes"] += g["graph_nodes"]; tot["gedges"] += g["graph_edges"]
        if r["bad"]:
            tot["bad"] += 1
            print("MISMATCH window %d: %s reads %d L %d eps %s realn %d refmode %d sam %s -> %s" % (w, kw, nreads, L, eps, realn_pass, refmode, cigs is not None, [(i, rc["mismatch"]) for i, rc in enumerate(r["recs"]) if rc["mismatch"]]), flush=True)
    print("%d windows (seed %d, %.0f s): %d reads through the library's own graph, %d declined by the kernel (whole-read bands above 256 columns), %d re-imports; compared without a difference: "
          "%d selected nodes, %.1f MB of programs, %d walk steps, %d graph nodes and %d edge-list entries after the surgeries; windows with a mismatch: %d"
          % (tot["windows"], seed, time.time() - t0, tot["reads"], tot["declined"], tot["imports"], tot["sel"], tot["prog"] / 1e6, tot["steps"], tot["gnodes"], tot["gedges"], tot["bad"]))
    print("of them %d windows with the realn pass, %d in refmode (%d with SAM CIGARs)" % (tot.get("realn", 0), tot.get("refmode", 0), tot.get("sam", 0)))
    return 1 if tot["bad"] else 0


if __name__ == "__main__":
    sys.exit(main())
