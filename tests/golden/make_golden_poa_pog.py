"""Generates tests/golden/poa_pog.npz (run in the build container; needs oracle/_ref/libbsref_trace.so = the reference with the test-only recording
hook of oracle/bspoa_trace_record.diff).  For a few small POA windows under several parameter sets, per aligned read: the reference's WHOLE graph as
flat arrays right before the read (nodes with their rings and columns, every edge list in order), the guide alignment prepare_rd_align_bspoa made,
and what the reference then decided -- its selection list (sel_nodes_bspoa), band width / read interval / auxiliary edges (prepare_rd_align_bspoa),
the program the binding flattened from its graph, its best end cell, every step of its own alignment2graph_bspoa walk, the result of align_rd_bspoa --
plus the graph its last surgery left.  tests/test_poa_pog_fixture.py replays them through bsa_pog_* without any reference build."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import poa_support as P

SETS = [P.par(), P.par(alnmode=0, bandwidth=64), P.par(alnmode=2, Q=0, P=0), P.par(O=0, E=-3, Q=0, P=0, bandwidth=32, nrec=2), P.par(bwtrigger=0, bandwidth=0, seqcore=5)]
out = {"ncases": np.array([len(SETS)])}
for c, p in enumerate(SETS):
    reads = P.synth_reads(9300 + c, 240 if p["bandwidth"] == 0 else 420, 6, eps=(0.06, 0.12, 0.2))
    r = P.run_ref_graph(reads, 5, p, record=5, lib=P.ref_poa_trace())
    assert r["bad"] == 0
    snaps = r["snaps"]
    out["par_%d" % c] = np.array([p[k] for k in P.PAR_ORDER], dtype=np.int32)
    out["nsnap_%d" % c] = np.array([len(snaps)])
    # the records of the aligned reads, in order (a read the graph form declined has no program: its snapshot is still checked for selection and placement)
    recs = r["recs"]
    assert len(recs) == len(snaps) - 1
    for k, sn in enumerate(snaps):
        for key in ("nodes", "ndoff", "rdlen", "out_off", "out_to", "out_cov", "in_off", "in_from", "cigar", "sels", "aux"):
            out["s%d_%d_%s" % (c, k, key)] = sn[key].view(np.uint8) if key == "nodes" else sn[key]
        out["s%d_%d_hdr" % (c, k)] = np.array([sn[h] for h in P.SNAP_HDR], dtype=np.int64)
        if k < len(recs):
            rc = recs[k]
            have = len(rc["nodes"]) > 0
            out["s%d_%d_rs" % (c, k)] = rc["rs"]
            out["s%d_%d_best" % (c, k)] = np.array([rc["maxscr"], rc["maxidx"], rc["maxoff"], rc["fin_gnode"], rc["fin_x"], int(have)], dtype=np.int64)
            out["s%d_%d_pnodes" % (c, k)] = rc["nodes"].view(np.uint8)
            out["s%d_%d_pedges" % (c, k)] = rc["edges"].view(np.uint8)
            out["s%d_%d_pcands" % (c, k)] = rc["cands"].view(np.uint8)
            out["s%d_%d_trace" % (c, k)] = rc["trace"].view(np.uint8)
            out["s%d_%d_query" % (c, k)] = rc["query"]
    print("case %d: %d snapshots, %d nodes in the last, %d programs" % (c, len(snaps), snaps[-1]["nnodes"], sum(1 for rc in recs if len(rc["nodes"]))))
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "poa_pog.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes")
