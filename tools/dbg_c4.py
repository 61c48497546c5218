import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np
import poa_support as P, support as S
import bsalign_amd as B
from test_poa_graph_gpu import _sweep_params
ctx = B.Context(0)
p = P.par()
nreads = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reads = P.synth_reads(20240611 & 0xFFFF, 20000, nreads, eps=(0.1,))
r = P.run_ref_graph(reads, 5, p, record=True, lib=P.ref_poa_trace(), backend="oracle")
print("bad", r["bad"], "graph reads", r["graph_reads"])
for k, rd in enumerate(r["recs"]):
    if "nodes" not in rd or len(rd["nodes"]) < 2 or rd["bandwidth"] > 256 or k < int(os.environ.get("DBG_FROM", "0")): continue
    cap = 2 * (rd["slen"] + len(rd["nodes"])) + 64
    pr = np.zeros(1, P.WF_PROG)
    pr[0] = (0, len(rd["nodes"]), 0, len(rd["edges"]), 0, len(rd["cands"]), rd["slen"], cap, 0, 0)
    res, ev, _, _ = ctx.poa_graph_host(rd["nodes"], rd["edges"], rd["cands"], pr, rd["query"], _sweep_params(p, rd["bandwidth"]), cap)
    rr = res[0]
    mine = ev[:int(rr["nevents"])]
    tr = rd["trace"]
    gn = rd["nodes"]["gnode"][mine["node"]]
    nmin = min(len(mine), len(tr))
    diff = np.nonzero((gn[:nmin] != tr["node"][:nmin]) | (mine["x"][:nmin] != tr["x"][:nmin]) | (mine["bt"][:nmin] != tr["bt"][:nmin]))[0]
    ok = rr["status"] == 0 and len(mine) == len(tr) and len(diff) == 0
    print("read", k, "bw", rd["bandwidth"], "nodes", len(rd["nodes"]), "steps", len(tr), "status", int(rr["status"]), "mine", len(mine), "OK" if ok else "DIFF")
    if not ok:
        i = int(diff[0]) if len(diff) else nmin
        g2l = {int(g): j for j, g in enumerate(rd["nodes"]["gnode"]) if g != 0xFFFFFFFF}
        print("  first difference at step", i)
        for j in range(max(0, i - 3), min(nmin, i + 3)):
            ln = g2l[int(tr["node"][j])]
            print("   step", j, "mine (node %d, x %d, bt %d)" % (mine["node"][j], mine["x"][j], mine["bt"][j]), "ref (node %d, x %d, bt %d)" % (ln, tr["x"][j], tr["bt"][j]),
                  "rpos", int(rd["nodes"]["rpos"][ln]), "nin", int(rd["nodes"]["n_in"][ln]), "first_in", int(rd["nodes"]["first_in"][ln]))
        ln = g2l[int(tr["node"][i])] if i < len(tr) else -1
        if ln >= 0:
            nd = rd["nodes"][ln]
            for e in rd["edges"][int(nd["first_in"]):int(nd["first_in"]) + int(nd["n_in"])]:
                print("    in-edge src", int(e["src"]), "cov", int(e["cov"]), "src_rpos", int(e["src_rpos"]), "delta node", ln - int(e["src"]))
        break
