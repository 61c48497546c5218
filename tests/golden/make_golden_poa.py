"""Generate tests/golden/poa_sweep.npz from the REAL reference (oracle/_ref/libbsref.so, built from /root/reference by
oracle/Makefile).  Run in the build container only:  python tests/golden/make_golden_poa.py

Every case runs the reference POA (beg/push/end_bspoa) on seeded synthetic reads with align_rd_bspoacore replaced by the
adapter + oracle sweep (harness mode 2), which re-runs the reference's own sweep after every read; a case is only
written if every read matched and the consensus / MSA equal the untouched end_bspoa (mode 0).  Stored per read: the
flattened program, the query, and the REFERENCE's results (best end cell, seqalign_result_t, FNV-1a of its row blocks).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import poa_support as P  # noqa: E402

CASES = [
    (11, 420, 7, P.par()),                                   # default POA parameters: overlap, bw 128, 2-piece gaps
    (12, 500, 6, P.par(bandwidth=64)),
    (13, 300, 6, P.par(bandwidth=32, Q=0, P=0)),             # 1-piece affine
    (14, 260, 5, P.par(bandwidth=16, O=0, E=-3, Q=0, P=0)),  # linear gaps
    (15, 400, 6, P.par(alnmode=0)),                          # global
    (16, 400, 6, P.par(alnmode=2)),                          # extend
    (17, 90, 6, P.par()),                                    # reads shorter than the band: W = 5..7
    (18, 200, 5, P.par(bandwidth=0)),                        # band = whole read
    (19, 600, 5, P.par(bandwidth=256)),
]


def main():
    out = {"ncases": np.array([len(CASES)], dtype=np.int32)}
    for c, (seed, L, n, p) in enumerate(CASES):
        reads = P.synth_reads(seed, L, n)
        r0 = P.run_ref_poa(reads, 0, p)
        r2 = P.run_ref_poa(reads, 2, p)
        assert r2["bad"] == 0, "case %d: adapter/oracle differs from the reference sweep" % c
        assert all(np.array_equal(r0[k], r2[k]) for k in ("cns", "qlt", "alt")) and r0["msa"] == r2["msa"], "case %d: end-to-end mismatch" % c
        meta = np.zeros((len(r2["recs"]), 22), dtype=np.int64)
        for k, rec in enumerate(r2["recs"]):
            meta[k, :10] = rec["rs"]
            meta[k, 10:20] = [rec["maxscr"], rec["maxidx"], rec["maxoff"], rec["bandwidth"], rec["slen"], rec["qb"], rec["nblocks"],
                              rec["ntasks"], rec["piecewise"], rec["mismatch"]]
            meta[k, 20], meta[k, 21] = rec["task_off"], rec["query_off"]
        out["par_%d" % c] = np.array([p[k] for k in P.PAR_ORDER], dtype=np.int32)
        out["meta_%d" % c] = meta
        out["hash_%d" % c] = np.array([rec["rows_hash"] for rec in r2["recs"]], dtype=np.uint64)
        out["tasks_%d" % c] = r2["tasks"].view(np.uint8)
        out["queries_%d" % c] = r2["queries"]
        out["cns_%d" % c] = r0["cns"]
        print("case", c, "reads", len(reads), "programs", len(r2["recs"]), "tasks", len(r2["tasks"]),
              "bandwidths", sorted(set(rec["bandwidth"] for rec in r2["recs"])))
    np.savez_compressed(P.GOLDEN, **out)
    print("wrote", P.GOLDEN, os.path.getsize(P.GOLDEN), "bytes")


if __name__ == "__main__":
    main()
