"""debugging aid (GPU box): the reads of BASELINE's C4 window through the reference's harness (oracle backend, mode 5: programs and the reference's own
walk recorded), then every read's program on the device -- step-by-step comparison with the reference's walk, wall time and kernel time of the call."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import poa_support as P
import bsalign_amd as B
from test_poa_graph_gpu import _sweep_params
ctx = B.Context(0)
p = P.par()
nreads = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reads = P.synth_reads(20240611 & 0xFFFF, 20000, nreads, eps=(0.1,))
r = P.run_ref_graph(reads, 5, p, record=True, lib=P.ref_poa_trace(), backend="oracle")
print("bad", r["bad"], "graph reads", r["graph_reads"])
for k, rd in enumerate(r["recs"]):
    if "nodes" not in rd or len(rd["nodes"]) < 2 or k < int(os.environ.get("DBG_FROM", "0")):
        continue
    cap = 2 * (rd["slen"] + len(rd["nodes"])) + 64
    pr = np.zeros(1, P.WF_PROG)
    pr[0] = (0, len(rd["nodes"]), 0, len(rd["edges"]), 0, len(rd["cands"]), rd["slen"], cap, 0, 0)
    sp = _sweep_params(dict(p, alnmode=p["alnmode"] | (0x200 if os.environ.get("DBG_FWD_ONLY") else 0)), rd["bandwidth"])
    ctx.poa_graph_host(rd["nodes"], rd["edges"], rd["cands"], pr, rd["query"], sp, cap)
    t0 = time.perf_counter()
    res, ev, _, _ = ctx.poa_graph_host(rd["nodes"], rd["edges"], rd["cands"], pr, rd["query"], sp, cap)
    wall = time.perf_counter() - t0
    kms = ctx.last_kernel_ms()[0]
    rr = res[0]
    mine = ev[:int(rr["nevents"])]
    tr = rd["trace"]
    gn = rd["nodes"]["gnode"][mine["node"]]
    nmin = min(len(mine), len(tr))
    diff = np.nonzero((gn[:nmin] != tr["node"][:nmin]) | (mine["x"][:nmin] != tr["x"][:nmin]) | (mine["bt"][:nmin] != tr["bt"][:nmin]))[0]
    ok = rr["status"] == 0 and len(mine) == len(tr) and len(diff) == 0
    print("read", k, "bw", rd["bandwidth"], "nodes", len(rd["nodes"]), "edges", len(rd["edges"]), "steps", len(tr), "status", int(rr["status"]), "OK" if ok else "DIFF at step %d" % (int(diff[0]) if len(diff) else nmin),
          "call %.1f ms, kernel %.1f ms" % (wall * 1e3, kms))
    if os.environ.get("DBG_DIST"):
        nd = rd["nodes"]; idx = np.arange(len(nd))
        for nm in ("in0", "in1"):
            pres = (nd[nm + "_tk"] & 0x80000000) != 0
            dist = (idx - nd[nm + "_src"].astype(np.int64))[pres]
            print("   ", nm, "present", int(pres.sum()), "distance > 7: %.2f %%, > 15: %.2f %%, > 31: %.2f %%, max %d" % (100.0 * (dist > 7).mean(), 100.0 * (dist > 15).mean(), 100.0 * (dist > 31).mean(), int(dist.max())),
                  "movx > 8: %.3f %%" % (100.0 * (nd[nm + "_movx"][pres] > 8).mean()))
