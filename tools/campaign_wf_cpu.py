"""CPU campaign: the absolute-score sweep (oracle/bsalign_oracle_wf.c) against the REAL reference (oracle/_ref): windows of synthetic
reads run through the reference's end_bspoa with every read's program recorded; every program's row blocks (hash) and best
end cell must be the reference's.  usage: python tools/campaign_wf_cpu.py [windows] [reads] [length]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import poa_support as P

nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
L = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
sets = [P.par(), P.par(alnmode=0), P.par(alnmode=2), P.par(Q=0, P=0), P.par(O=0, E=-3, Q=0, P=0), P.par(bandwidth=64), P.par(bandwidth=256),
        P.par(bandwidth=32, alnmode=0), P.par(M=3, X=-5, O=-4, E=-3, Q=-12, P=-1), P.par(nrec=3)]
tot = bad = 0
t0 = time.time()
for si, p in enumerate(sets):
    for w in range(nwin):
        reads = P.synth_reads(1000 * si + w, L, nreads, eps=(0.03, 0.1, 0.2))
        r = P.run_ref_poa(reads, 1, p, record=True)
        for rc in r["recs"]:
            if rc["bandwidth"] > 512:
                continue
            t = r["tasks"][rc["task_off"]:rc["task_off"] + rc["ntasks"]]
            q = r["queries"][rc["query_off"]:rc["query_off"] + rc["slen"]]
            nodes, edges, cands, blocks = P.tasks_to_graph(t)
            rows, u0 = P.oracle_wf_forward(nodes, q, p, rc["bandwidth"])
            mine = P.wf_rows_to_blocks(rows, u0, blocks, rc["nblocks"], rc["bandwidth"], rc["piecewise"])
            best = P.oracle_wf_best(nodes, cands, rc["slen"], p, rc["bandwidth"], rows)
            gidx = int(nodes[int(best["maxidx"])]["gnode"]) if best["maxidx"] >= 0 else -1
            ok = P.hash_node_blocks(mine, rc["nblocks"], rc["bandwidth"], rc["piecewise"], t) == rc["rows_hash"] and \
                (int(best["maxscr"]), gidx, int(best["maxoff"])) == (rc["maxscr"], rc["maxidx"], rc["maxoff"])
            tot += 1; bad += (not ok)
            if not ok:
                print("DIFF set %d window %d read bw %d" % (si, w, rc["bandwidth"]))
    print("set %d done: %d programs, %d differ, %.0f s" % (si, tot, bad, time.time() - t0), flush=True)
print("campaign: %d programs, %d differ" % (tot, bad))
