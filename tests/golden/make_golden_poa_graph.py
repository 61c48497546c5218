"""Generates tests/golden/poa_graph.npz (run in the build container, needs oracle/_ref/libbsref_trace.so = the reference with the
test-only recording hook of oracle/bspoa_trace_record.diff).  Per read of a few small POA windows under several parameter sets:
the graph-form program the binding built (nodes, in-edges, candidates, read) and what the REFERENCE did with it -- its best end
cell and every (node, x, bt) step of its own alignment2graph_bspoa walk with the walk's end."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import poa_support as P

SETS = [P.par(), P.par(alnmode=0), P.par(alnmode=2), P.par(Q=0, P=0), P.par(O=0, E=-3, Q=0, P=0, bandwidth=32), P.par(bandwidth=64), P.par(bandwidth=256, nrec=2)]
out = {"ncases": np.array([len(SETS)])}
for c, p in enumerate(SETS):
    reads = P.synth_reads(9100 + c, 500 if p["bandwidth"] < 256 else 700, 7, eps=(0.06, 0.12, 0.2))
    r = P.run_ref_graph(reads, 5, p, record=True, lib=P.ref_poa_trace())
    assert r["bad"] == 0
    recs = [rc for rc in r["recs"] if "nodes" in rc and len(rc["nodes"])]
    out["par_%d" % c] = np.array([p[k] for k in P.PAR_ORDER], dtype=np.int32)
    meta = []
    n0 = e0 = c0 = t0 = q0 = 0
    for rc in recs:
        # bandwidth, slen, maxscr, maxidx (graph node), maxoff, fin node (graph), fin x, nnodes, nedges, ncands, ntrace, offsets
        meta.append((rc["bandwidth"], rc["slen"], rc["maxscr"], rc["maxidx"], rc["maxoff"], rc["fin_gnode"], rc["fin_x"],
                     len(rc["nodes"]), len(rc["edges"]), len(rc["cands"]), len(rc["trace"]), n0, e0, c0, t0, q0))
        n0 += len(rc["nodes"]); e0 += len(rc["edges"]); c0 += len(rc["cands"]); t0 += len(rc["trace"]); q0 += rc["slen"]
    out["meta_%d" % c] = np.array(meta, dtype=np.int64)
    out["nodes_%d" % c] = np.concatenate([rc["nodes"] for rc in recs]).view(np.uint8)
    out["edges_%d" % c] = np.concatenate([rc["edges"] for rc in recs]).view(np.uint8)
    out["cands_%d" % c] = np.concatenate([rc["cands"] for rc in recs]).view(np.uint8)
    out["trace_%d" % c] = np.concatenate([rc["trace"] for rc in recs]).view(np.uint8)
    out["query_%d" % c] = np.concatenate([rc["query"] for rc in recs])
    print("case %d: %d reads, %d nodes, %d steps" % (c, len(recs), n0, t0))
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "poa_graph.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes")
