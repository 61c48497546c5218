"""The absolute-score formulation of the POA sweep (oracle/bsalign_oracle_wf.c -- what the device's wavefront kernel computes)
against the lane-exact restatement of the reference's striped int8 rows (orc_sweep_run), on the recorded programs of real
end_bspoa runs (tests/golden/poa_sweep.npz): every row block byte for byte, and the best end cell."""
import numpy as np
import pytest

import poa_support as P


def _check(case, pg):
    p = case["par"]
    nodes, edges, cands, blocks = P.tasks_to_graph(pg["tasks"])
    rows, u0 = P.oracle_wf_forward(nodes, pg["query"], p, pg["bandwidth"])
    bw, pw = pg["bandwidth"], pg["piecewise"]
    mine = P.wf_rows_to_blocks(rows, u0, blocks, pg["nblocks"], bw, pw)
    t = pg["tasks"].copy(); t["query"] = 0
    orows, ores = P.oracle_sweep(t, np.array([(0, len(t), 0, 0)], dtype=P.PROG_DTYPE), pg["query"], np.zeros(1, np.uint64),
                                 np.array([pg["slen"]], np.uint32), p, bw, pg["nblocks"], pw)
    blk, used = P.block_bytes(bw, pw), bw * (pw + 1) + 68
    real = blocks[blocks != 0]
    a = mine.reshape(pg["nblocks"], blk)[real, :used]
    b = orows.reshape(pg["nblocks"], blk)[real, :used]
    bad = np.nonzero((a != b).any(axis=1))[0]
    assert len(bad) == 0, "row of block %d differs: first byte %d" % (real[bad[0]], np.nonzero(a[bad[0]] != b[bad[0]])[0][0])
    best = P.oracle_wf_best(nodes, cands, pg["slen"], p, bw, rows)
    gidx = int(nodes[int(best["maxidx"])]["gnode"]) if best["maxidx"] >= 0 else -1
    assert (int(best["maxscr"]), gidx, int(best["maxoff"])) == (pg["maxscr"], pg["maxidx"], pg["maxoff"])
    assert P.hash_node_blocks(mine, pg["nblocks"], bw, pw, pg["tasks"]) == pg["rows_hash"]


def test_recorded_programs_of_the_reference():
    n = 0
    for case in P.load_golden():
        for pg in case["programs"]:
            if pg["bandwidth"] > 512:       # whole-read bands of a window's first read: two nodes, nothing to compute
                continue
            _check(case, pg)
            n += 1
    assert n >= 30


def test_fixture_of_the_references_own_walks():
    """tests/golden/poa_graph.npz: programs as the binding builds them from the reference's graph (in-edges in erev order with
    their coverage) and what the reference did: the scalar statement of the device kernel finds the same best end cell and takes
    the same steps, one by one"""
    steps = 0
    for case in P.load_golden_graph():
        p = case["par"]
        for rd in case["reads"]:
            rows, u0 = P.oracle_wf_forward(rd["nodes"], rd["query"], p, rd["bandwidth"])
            best = P.oracle_wf_best(rd["nodes"], rd["cands"], rd["slen"], p, rd["bandwidth"], rows)
            assert (int(best["maxscr"]), int(rd["nodes"][int(best["maxidx"])]["gnode"]), int(best["maxoff"])) == (rd["maxscr"], rd["maxidx"], rd["maxoff"])
            n, ev, fin = P.oracle_wf_trace(rd["nodes"], rd["edges"], rd["query"], p, rd["bandwidth"], rows, u0, 0, int(best["maxidx"]), int(best["maxoff"]))
            assert n == len(rd["trace"])
            assert np.array_equal(rd["nodes"]["gnode"][ev["node"]], rd["trace"]["node"]) and np.array_equal(ev["x"], rd["trace"]["x"]) and np.array_equal(ev["bt"], rd["trace"]["bt"])
            assert (int(rd["nodes"][int(fin[0])]["gnode"]), int(fin[1])) == (rd["fin_gnode"], rd["fin_x"])
            steps += n
    assert steps > 20000
