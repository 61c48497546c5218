"""tests/golden/poa_pog.npz -- the reference's flat graph before every aligned read with what it then decided (make_golden_poa_pog.py) -- and the replay of
one window through the library's own graph surface (bsalign_amd/poa.py -> bsa_pog_*), shared by the CPU test (the reference's recorded walk stands in for
the device) and the GPU test (the MI355X runs the program)."""
import os

import numpy as np

import bsalign_amd as B
import poa_support as P
from bsalign_amd import poa as PG

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "poa_pog.npz")


def load():
    z = np.load(PATH)
    cases = []
    for c in range(int(z["ncases"][0])):
        par = dict(zip(P.PAR_ORDER, (int(x) for x in z["par_%d" % c])))
        snaps = []
        for k in range(int(z["nsnap_%d" % c][0])):
            d = dict(zip(P.SNAP_HDR, (int(x) for x in z["s%d_%d_hdr" % (c, k)])))
            for key in ("ndoff", "rdlen", "out_off", "out_to", "out_cov", "in_off", "in_from", "cigar", "sels", "aux"):
                d[key] = z["s%d_%d_%s" % (c, k, key)]
            d["nodes"] = z["s%d_%d_nodes" % (c, k)].view(PG.POG_NODE_DTYPE)
            if "s%d_%d_rs" % (c, k) in z:
                d["rs"] = z["s%d_%d_rs" % (c, k)]
                d["best"] = [int(x) for x in z["s%d_%d_best" % (c, k)]]
                d["pnodes"] = z["s%d_%d_pnodes" % (c, k)].view(B.POA_NODE_DTYPE)
                d["pedges"] = z["s%d_%d_pedges" % (c, k)].view(B.POA_EDGE_DTYPE)
                d["pcands"] = z["s%d_%d_pcands" % (c, k)].view(B.POA_CAND_DTYPE)
                d["trace"] = z["s%d_%d_trace" % (c, k)].view(B.POA_EVENT_DTYPE)
                d["query"] = z["s%d_%d_query" % (c, k)]
            snaps.append(d)
        cases.append(dict(par=par, snaps=snaps))
    return cases


def same_structure(g, sn):
    """the library's exported graph against a snapshot of the reference's: rings, reads, coverages (kept on the headers), flags, every edge list in order
    (columns are the caller's input before every read -- msa_bspoa recomputes them -- and not part of the structure)"""
    a, b = g["nodes"], sn["nodes"]
    if len(a) != len(b) or g["head"] != sn["head"] or g["tail"] != sn["tail"]:
        return "sizes"
    for f in ("header", "next", "prev", "pos", "rid", "base", "flags"):
        if not np.array_equal(a[f], b[f]):
            return f
    hd = a["header"] == np.arange(len(a))
    if not np.array_equal(a["cov"][hd], b["cov"][hd]):
        return "cov"
    for f in ("ndoff", "rdlen", "out_off", "out_to", "out_cov", "in_off", "in_from"):
        if not np.array_equal(g[f], sn[f]):
            return f
    return None


def replay_window(case, run):
    """the window's reads through ONE library graph: built from the first snapshot, then evolving by its own surgery; before every read it must equal the
    reference's graph of that moment.  run(pog, sn) -> (result, events) executes the program (device, or the recorded walk).  -> counters"""
    par, snaps = case["par"], case["snaps"]
    pog = PG.Pog(**par)
    stats = dict(reads=0, imports=0, sel=0, prog_bytes=0, steps=0, graph_nodes=0)
    try:
        pog.import_graph(snaps[0]); stats["imports"] += 1
        for k, sn in enumerate(snaps[:-1]):
            why = same_structure(pog.export_graph(), sn)
            assert why is None, (k, "graph before the read differs in", why)
            stats["graph_nodes"] += len(sn["nodes"])
            rd, sel = pog.select(sn["rid"], 0, sn["rlen"])
            assert np.array_equal(sel, sn["sels"]), (k, "selection")
            stats["sel"] += len(sel)
            assert pog.needs_guide(sn["reflen"]) == bool(sn["have"]), (k, "guide")
            cpos = sn["nodes"]["cpos"][sel]
            rd = pog.place(sn["reflen"], (sn["gqb"], sn["gqe"], sn["gtb"], sn["gte"]) if sn["have"] else None, sn["cigar"], cpos)
            assert (rd.bandwidth, rd.slen, rd.qb, rd.qe, rd.qlen) == (sn["bandwidth"], sn["slen"], sn["qb"], sn["qe"], sn["rlen"]), (k, "placement")
            assert np.array_equal(pog.aux_edges(), sn["aux"]), (k, "auxiliary edges")
            if not sn["best"][5]:
                # the kernel declined this read in the recording run (a window's first read: whole-read band): the reference's own path took it, the library's
                # graph is re-imported from the next snapshot, as the binding does
                pog.abort()
                pog.import_graph(snaps[k + 1]); stats["imports"] += 1
                continue
            pn, pe, pc, q, sp = pog.program()
            assert pn.tobytes() == sn["pnodes"].tobytes() and pe.tobytes() == sn["pedges"].tobytes() and pc.tobytes() == sn["pcands"].tobytes(), (k, "program")
            assert sp.rows.bandwidth == sn["bandwidth"] and sp.rows.mode == (par["alnmode"] & 3)
            stats["prog_bytes"] += pn.nbytes + pe.nbytes + pc.nbytes
            res, ev = run(pog, sn)
            gn_of = sn["pnodes"]["gnode"]
            assert (int(res["maxscr"]), int(gn_of[int(res["maxidx"])]), int(res["maxoff"])) == tuple(sn["best"][:3]), (k, "best end cell")
            assert len(ev) == len(sn["trace"]) and np.array_equal(gn_of[ev["node"]], sn["trace"]["node"]) and np.array_equal(ev["x"], sn["trace"]["x"]) and np.array_equal(ev["bt"], sn["trace"]["bt"]), (k, "walk")
            rs, gn = pog.apply(len(ev))
            assert np.array_equal(np.array([rs[f] for f in rs.dtype.names], np.int32)[:9], sn["rs"][:9]), (k, "result", rs, sn["rs"])
            assert np.array_equal(gn, sn["trace"]["node"])
            stats["steps"] += len(ev); stats["reads"] += 1
        why = same_structure(pog.export_graph(), snaps[-1])
        assert why is None, ("graph after the last read differs in", why)
    finally:
        pog.close()
    return stats


def recorded_walk(pog, sn):
    """the reference's own recorded walk as the backend: best end cell and steps in the program's local node indices"""
    local = {int(g): i for i, g in enumerate(sn["pnodes"]["gnode"]) if g != 0xFFFFFFFF}
    ev = np.zeros(len(sn["trace"]), B.POA_EVENT_DTYPE)
    ev["node"] = [local[int(g)] for g in sn["trace"]["node"]]
    ev["x"] = sn["trace"]["x"]; ev["bt"] = sn["trace"]["bt"]

    def backend(nodes, edges, cands, query, slen):
        assert np.array_equal(query, sn["query"])
        return dict(maxscr=sn["best"][0], maxidx=local[sn["best"][1]], maxoff=sn["best"][2], status=0, fin_node=local[sn["best"][3]], fin_x=sn["best"][4]), ev
    return pog.run(backend=backend)
