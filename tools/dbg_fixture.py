"""debugging aid (GPU box): the programs of tests/golden/poa_graph.npz one by one -- the device's walk against the reference's recorded steps, first difference"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import poa_support as P
import bsalign_amd as B
from test_poa_graph_gpu import _sweep_params
ctx = B.Context(0)
for ci, case in enumerate(P.load_golden_graph()):
    p = case["par"]
    for ri, rd in enumerate(case["reads"]):
        if rd["bandwidth"] > 256 or len(rd["nodes"]) < 2: continue
        cap = 2 * (rd["slen"] + len(rd["nodes"])) + 64
        pr = np.zeros(1, P.WF_PROG)
        pr[0] = (0, len(rd["nodes"]), 0, len(rd["edges"]), 0, len(rd["cands"]), rd["slen"], cap, 0, 0)
        res, ev, _, _ = ctx.poa_graph_host(rd["nodes"], rd["edges"], rd["cands"], pr, rd["query"], _sweep_params(p, rd["bandwidth"]), cap)
        rr = res[0]; mine = ev[:int(rr["nevents"])]; tr = rd["trace"]
        gn = rd["nodes"]["gnode"][mine["node"]]
        nmin = min(len(mine), len(tr))
        diff = np.nonzero((gn[:nmin] != tr["node"][:nmin]) | (mine["x"][:nmin] != tr["x"][:nmin]) | (mine["bt"][:nmin] != tr["bt"][:nmin]))[0]
        ok = rr["status"] == 0 and len(mine) == len(tr) and len(diff) == 0
        if not ok:
            i = int(diff[0]) if len(diff) else nmin
            g2l = {int(g): j for j, g in enumerate(rd["nodes"]["gnode"]) if g != 0xFFFFFFFF}
            print("case", ci, "read", ri, "bw", rd["bandwidth"], "nodes", len(rd["nodes"]), "status", int(rr["status"]), "steps", len(mine), "of", len(tr), "first difference at", i)
            for j in range(max(0, i - 2), min(nmin, i + 2)):
                ln = g2l[int(tr["node"][j])]
                print("   step", j, "mine (node %d, x %d, bt %d)" % (mine["node"][j], mine["x"][j], mine["bt"][j]), "ref (node %d, x %d, bt %d)" % (ln, tr["x"][j], tr["bt"][j]))
            if i < len(tr):
                ln = g2l[int(tr["node"][i])]; nd = rd["nodes"][ln]
                for e in rd["edges"][int(nd["first_in"]):int(nd["first_in"]) + int(nd["n_in"])]:
                    print("    in-edge of node", ln, ": src", int(e["src"]), "cov", int(e["cov"]), "src_rpos", int(e["src_rpos"]))
            sys.exit(0)
print("all equal")
