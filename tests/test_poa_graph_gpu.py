"""GPU parity of the wavefront sweep + device traceback (bsa_poa_graph_host; bsalign_amd/csrc/bsa_poa_wf.hip):
* every row of every recorded program of the real end_bspoa (tests/golden/poa_sweep.npz), converted back into the reference's
  row blocks, byte for byte against the lane-exact oracle (orc_sweep_run) and against the reference's own row hash;
* the best end cell against the reference's;
* the traceback steps against the scalar statement of the walk (orc_wf_trace) on the same rows."""
import numpy as np
import pytest

import poa_support as P
import support as S

pytestmark = pytest.mark.gpu


def _sweep_params(p, bandwidth):
    import bsalign_amd as B
    sp = B.SweepParams()
    sp.rows = B.RowsParams(p["alnmode"], bandwidth, p["M"], p["X"], p["refbonus"], p["O"], p["E"], p["Q"], p["P"])
    sp.T = p["T"]
    return sp


def _prog(nodes, edges, cands, slen, first_event=0, base=(0, 0, 0), qoff=0):
    cap = 4 * (slen + len(nodes)) + 64
    pr = np.zeros(1, P.WF_PROG)
    pr[0] = (base[0], len(nodes), base[1], len(edges), base[2], len(cands), slen, cap, qoff, first_event)
    return pr, cap


def _check_program(ctx, p, pg, want_trace=True):
    bw, pw = pg["bandwidth"], pg["piecewise"]
    nodes, edges, cands, blocks = P.tasks_to_graph(pg["tasks"])
    pr, cap = _prog(nodes, edges, cands, pg["slen"])
    res, ev, rows, u0 = ctx.poa_graph_host(nodes, edges, cands, pr, pg["query"], _sweep_params(p, bw), cap, want_rows=True)
    orows, ou0 = P.oracle_wf_forward(nodes, pg["query"], p, bw)
    bad = np.nonzero((rows.view(np.uint64)[:, :] != orows.view(np.uint64)[:, :]).any(axis=1))[0]
    assert len(bad) == 0, ("first differing node", int(bad[0]), "of", len(nodes), "cell", int(np.nonzero(rows[bad[0]] != orows[bad[0]])[0][0]))
    assert np.array_equal(u0, ou0)
    mine = P.wf_rows_to_blocks(rows, u0, blocks, pg["nblocks"], bw, pw)
    assert P.hash_node_blocks(mine, pg["nblocks"], bw, pw, pg["tasks"]) == pg["rows_hash"]
    r = res[0]
    gidx = int(nodes[int(r["maxidx"])]["gnode"]) if r["maxidx"] >= 0 else -1
    assert (int(r["maxscr"]), gidx, int(r["maxoff"])) == (pg["maxscr"], pg["maxidx"], pg["maxoff"])
    if want_trace:
        n, oev, fin = P.oracle_wf_trace(nodes, edges, pg["query"], p, bw, orows, ou0, 0, int(r["maxidx"]), int(r["maxoff"]))
        assert (r["status"] == 0) == (n >= 0), (int(r["status"]), n)
        if n >= 0:
            assert int(r["nevents"]) == n and np.array_equal(ev[:n], oev)
            assert (int(r["fin_node"]), int(r["fin_x"])) == (int(fin[0]), int(fin[1]))


@pytest.mark.parametrize("fwd", ["rows", "wf", "rows-ring2", "rows-ring4"])
def test_golden_programs_rows_best_cell_and_walk(ctx, fwd, monkeypatch):
    """both forward passes of k_poa_wf -- a row at a time (the default) and the anti-diagonal wavefront (BSA_POA_FWD=wf) -- against the
    scalar statement, the reference's row hashes, its best end cell and the walk.  `ring2` / `ring4`: the row-at-a-time pass with a ring of
    two / four rows (BSA_POA_FWD_RING), so that inputs further back are read back from HBM -- the path a deep graph takes now and then"""
    if fwd == "wf":
        monkeypatch.setenv("BSA_POA_FWD", "wf")
    if fwd.startswith("rows-ring"):
        monkeypatch.setenv("BSA_POA_FWD_RING", fwd[len("rows-ring"):])
    n = 0
    for case in P.load_golden():
        for pg in case["programs"]:
            if pg["bandwidth"] > 256:
                continue
            _check_program(ctx, case["par"], pg)
            n += 1
    assert n >= 30


def test_many_programs_in_one_launch(ctx):
    """all programs of a case that share a bandwidth as ONE launch (many windows side by side), three copies of each"""
    import bsalign_amd as B
    for case in P.load_golden()[:3]:
        p = case["par"]
        by_bw = {}
        for pg in case["programs"]:
            if pg["bandwidth"] <= 256:
                by_bw.setdefault(pg["bandwidth"], []).append(pg)
        for bw, pgs in by_bw.items():
            N, E, Cd, PR, Q, want = [], [], [], [], [], []
            n0 = e0 = c0 = q0 = v0 = 0
            for rep in range(3):
                for pg in pgs:
                    nodes, edges, cands, blocks = P.tasks_to_graph(pg["tasks"])
                    cap = 4 * (pg["slen"] + len(nodes)) + 64
                    pr = np.zeros(1, P.WF_PROG)
                    pr[0] = (n0, len(nodes), e0, len(edges), c0, len(cands), pg["slen"], cap, q0, v0)
                    N.append(nodes); E.append(edges); Cd.append(cands); PR.append(pr); Q.append(pg["query"]); want.append((pg, nodes))
                    n0 += len(nodes); e0 += len(edges); c0 += len(cands); q0 += pg["slen"]; v0 += cap
            res, ev, _, _ = ctx.poa_graph_host(np.concatenate(N), np.concatenate(E), np.concatenate(Cd), np.concatenate(PR), np.concatenate(Q),
                                               _sweep_params(p, bw), v0)
            for k, (pg, nodes) in enumerate(want):
                r = res[k]
                assert r["status"] == 0
                assert (int(r["maxscr"]), int(nodes[int(r["maxidx"])]["gnode"]), int(r["maxoff"])) == (pg["maxscr"], pg["maxidx"], pg["maxoff"])


@pytest.mark.parametrize("shift", [0, 40, -40])
def test_fixture_of_the_references_own_walks(ctx, shift, monkeypatch):
    """tests/golden/poa_graph.npz: the graph-form programs the binding built from the reference's graph, and the (node, x, bt) steps
    the REFERENCE's alignment2graph_bspoa took (recorded through oracle/ref_poa_harness.c with the test-only hook): the device's
    best end cell, every step of its walk and the walk's end are the reference's.  All reads of a case in one launch.
    `shift`: the traceback's windows of the rows put that many cells off their place (BSA_POA_WIN_SHIFT), so that the places a tile
    cannot decide, the steps taken one at a time and their reads from HBM carry the walk -- same steps."""
    if shift:
        monkeypatch.setenv("BSA_POA_WIN_SHIFT", str(shift))
    steps = 0
    for case in P.load_golden_graph():
        p = case["par"]
        by_bw = {}
        for rd in case["reads"]:
            by_bw.setdefault(rd["bandwidth"], []).append(rd)
        for bw, rds in by_bw.items():
            PR, n0, e0, c0, q0, v0 = [], 0, 0, 0, 0, 0
            for rd in rds:
                cap = 2 * (rd["slen"] + len(rd["nodes"])) + 64
                pr = np.zeros(1, P.WF_PROG)
                pr[0] = (n0, len(rd["nodes"]), e0, len(rd["edges"]), c0, len(rd["cands"]), rd["slen"], cap, q0, v0)
                PR.append(pr)
                n0 += len(rd["nodes"]); e0 += len(rd["edges"]); c0 += len(rd["cands"]); q0 += rd["slen"]; v0 += cap
            res, ev, _, _ = ctx.poa_graph_host(np.concatenate([r["nodes"] for r in rds]), np.concatenate([r["edges"] for r in rds]),
                                               np.concatenate([r["cands"] for r in rds]), np.concatenate(PR), np.concatenate([r["query"] for r in rds]),
                                               _sweep_params(p, bw), v0)
            for k, rd in enumerate(rds):
                r, pr = res[k], PR[k][0]
                assert r["status"] == 0
                assert (int(r["maxscr"]), int(rd["nodes"][int(r["maxidx"])]["gnode"]), int(r["maxoff"])) == (rd["maxscr"], rd["maxidx"], rd["maxoff"])
                mine = ev[int(pr["first_event"]):int(pr["first_event"]) + int(r["nevents"])]
                assert len(mine) == len(rd["trace"])
                assert np.array_equal(rd["nodes"]["gnode"][mine["node"]], rd["trace"]["node"]) and np.array_equal(mine["x"], rd["trace"]["x"]) and np.array_equal(mine["bt"], rd["trace"]["bt"])
                assert (int(rd["nodes"][int(r["fin_node"])]["gnode"]), int(r["fin_x"])) == (rd["fin_gnode"], rd["fin_x"])
                steps += len(mine)
    assert steps > 20000


def _attach(lib, ctx):
    import ctypes as C
    import bsalign_amd as B
    b = B.lib()
    lib.ref_poa_set_graph_host(C.cast(b.bsa_poa_graph_host, C.c_void_p), ctx.h)
    lib.ref_poa_set_device.argtypes = [C.c_void_p, C.c_void_p]
    lib.ref_poa_set_device.restype = None
    lib.ref_poa_set_device(C.cast(b.bsa_sweep_host, C.c_void_p), ctx.h)


@pytest.mark.skipif(not P.have_ref_trace(), reason="oracle/_ref/libbsref_trace.so not built")
@pytest.mark.parametrize("kw", [dict(), dict(alnmode=0, bandwidth=64), dict(alnmode=2, Q=0, P=0), dict(deep=48)])
def test_device_in_the_shadow_of_the_reference(ctx, kw):
    """harness mode 5 with the MI355X as backend: inside a real end_bspoa, read after read, the device's best end cell and every
    step of its walk against what the reference's own align_rd_bspoacore + alignment2graph_bspoa do on the same graph.  `deep`: a
    window of that many reads -- a graph in which the walk skips more than a node a step and predecessors lie many nodes back
    (tiles whose columns follow the drift, steps further down than the ring reaches)"""
    lib = P.ref_poa_trace()
    _attach(lib, ctx)
    kw = dict(kw)
    deep = kw.pop("deep", 0)
    p = P.par(**kw)
    reads = P.synth_reads(777, 1500, deep, eps=(0.08, 0.15, 0.12)) if deep else P.synth_reads(520 + len(kw), 2500, 14, eps=(0.05, 0.12, 0.2))
    r = P.run_ref_graph(reads, 5, p, record=True, lib=lib, backend="device")
    assert r["bad"] == 0, [(i, rc["mismatch"]) for i, rc in enumerate(r["recs"]) if rc["mismatch"]]
    assert r["graph_reads"] >= (len(reads) * 3 // 4 if deep else len(reads) - 3)          # (a window's first reads run with wider bands than the kernel takes)
    assert sum(len(rc["trace"]) for rc in r["recs"] if "trace" in rc) > (30 * 1500 if deep else 10 * 2500)


@pytest.mark.skipif(not S.have_ref(), reason="oracle/_ref/libbsref.so not built")
def test_end_bspoa_with_sweep_and_walk_on_the_device(ctx):
    """harness mode 6: the product's path -- graph form on the device, the binding applies the steps -- gives the untouched
    end_bspoa's consensus, qualities and MSA"""
    lib = P.ref_poa()
    _attach(lib, ctx)
    for kw in (dict(), dict(bandwidth=64), dict(alnmode=0)):
        p = P.par(**kw)
        reads = P.synth_reads(620 + len(kw), 3000, 16, eps=(0.08, 0.12))
        ref = P.run_ref_poa(reads, 0, p, record=False)
        mine = P.run_ref_graph(reads, 6, p, record=False, lib=lib, backend="device")
        assert mine["graph_reads"] >= len(reads) - 3
        assert np.array_equal(mine["cns"], ref["cns"]) and np.array_equal(mine["qlt"], ref["qlt"]) and np.array_equal(mine["alt"], ref["alt"]) and mine["msa"] == ref["msa"]


@pytest.mark.skipif(not S.have_ref(), reason="oracle/_ref/libbsref.so not built")
def test_c4_full_size_clean_wall_time(ctx, capsys):
    """BASELINE config C4 as stated: one window of 64 ONT-like reads x 20 kbp, default POA parameters.  The product's path alone
    (harness mode 6: no reference sweep or walk runs, nothing is re-checked read by read): the same consensus, qualities and MSA
    as the untouched end_bspoa, and the two wall times."""
    import time
    lib = P.ref_poa()
    _attach(lib, ctx)
    p = P.par()
    reads = P.synth_reads(20240611 & 0xFFFF, 20000, 64, eps=(0.1,))
    t0 = time.time(); ref = P.run_ref_poa(reads, 0, p, record=False); t_ref = time.time() - t0
    one = P.run_ref_poa(reads, 1, p, record=False)
    t0 = time.time(); mine = P.run_ref_graph(reads, 6, p, record=False, lib=lib, backend="device"); t_dev = time.time() - t0
    assert np.array_equal(mine["cns"], ref["cns"]) and np.array_equal(mine["qlt"], ref["qlt"]) and np.array_equal(mine["alt"], ref["alt"]) and mine["msa"] == ref["msa"]
    b = mine["binding_seconds"]
    with capsys.disabled():
        print("\n[C4 full size, clean] end_bspoa 64 x 20 kbp: reference %.2f s (of which align_rd_bspoacore %.2f s for %d row updates + %d merges); with sweep and walk "
              "on the device %.2f s (%d reads through the graph form: building programs %.2f s, device calls incl. transfers %.2f s, applying walks %.2f s)"
              % (t_ref, one["core_seconds"], one["core_updates"], one["core_merges"], t_dev, mine["graph_reads"], b[0], b[1], b[2]))


# ---- bands above 256 columns: the generic-width kernel (bsa_poa_gen.hip) behind the same entry point ----
def _check_program_gen(ctx, p, pg):
    """the generic-width kernel keeps its rows to itself: best end cell against the reference's, every step of the walk against the scalar
    statement on the scalar statement's rows (which are pinned to the reference's row hash)"""
    bw, pw = pg["bandwidth"], pg["piecewise"]
    nodes, edges, cands, blocks = P.tasks_to_graph(pg["tasks"])
    pr, cap = _prog(nodes, edges, cands, pg["slen"])
    res, ev, _, _ = ctx.poa_graph_host(nodes, edges, cands, pr, pg["query"], _sweep_params(p, bw), cap)
    orows, ou0 = P.oracle_wf_forward(nodes, pg["query"], p, bw)
    assert P.hash_node_blocks(P.wf_rows_to_blocks(orows, ou0, blocks, pg["nblocks"], bw, pw), pg["nblocks"], bw, pw, pg["tasks"]) == pg["rows_hash"]
    r = res[0]
    gidx = int(nodes[int(r["maxidx"])]["gnode"]) if r["maxidx"] >= 0 else -1
    assert (int(r["maxscr"]), gidx, int(r["maxoff"])) == (pg["maxscr"], pg["maxidx"], pg["maxoff"])
    n, oev, fin = P.oracle_wf_trace(nodes, edges, pg["query"], p, bw, orows, ou0, 0, int(r["maxidx"]), int(r["maxoff"]))
    assert (r["status"] == 0) == (n >= 0), (int(r["status"]), n)
    if n >= 0:
        assert int(r["nevents"]) == n and np.array_equal(ev[:n], oev)
        assert (int(r["fin_node"]), int(r["fin_x"])) == (int(fin[0]), int(fin[1]))
    return n


def test_generic_width_kernel_on_every_golden_program(ctx, monkeypatch):
    """BSA_POA_FORCE_GEN=1: every recorded program of tests/golden/poa_sweep.npz through the generic-width kernel, the narrow ones as well
    (merges of several in-edges, moved rows with their synthetic cells, dead rows, all three gap models and modes of the recording; one cell
    a thread there) and the ones above 256 columns k_poa_wf declines"""
    import bsalign_amd as B
    monkeypatch.setenv("BSA_POA_FORCE_GEN", "1")
    n = wide = 0
    for case in P.load_golden():
        for pg in case["programs"]:
            if pg["bandwidth"] > 256:
                assert B.lib().bsa_poa_graph_supported(_sweep_params(case["par"], pg["bandwidth"]), int(pg["slen"])) == 0
                wide += 1
            _check_program_gen(ctx, case["par"], pg)
            n += 1
    assert n >= 40 and wide >= 3


def _chain_program(rng, L, eps, mode):
    """what a window's first aligned read meets: the chain of the read before it (one in-edge a node, every band offset 0), the whole read as band"""
    T = rng.integers(0, 4, size=L).astype(np.uint8)
    Q = S.mutate(rng, T, eps)
    if len(Q) == 0:
        Q = T.copy()
    nodes = np.zeros(L + 1, P.WF_NODE); edges = np.zeros(L, P.WF_EDGE)
    nodes[0]["base"] = 4; nodes[0]["gnode"] = 0
    for i in range(1, L + 1):
        same = P.IN_SAME if (i >= 2 and T[i - 1] == T[i - 2]) else 0
        nodes[i] = (0, i, i - 1, 1, T[i - 1], int(rng.integers(2)), i - 1, 0, P.IN_PRESENT | same | i, 0, 0, 0, 0, 0)
        edges[i - 1] = (i - 1, 1 + int(rng.integers(3)), 0, 0)
    cands = np.zeros(L + 1 if mode != 0 else 1, P.WF_CAND)
    if mode != 0:
        cands["node"][:L] = np.arange(1, L + 1); cands["kind"][:L] = 1
    cands[-1] = (L, 0)
    return nodes, edges, cands, Q


@pytest.mark.parametrize("L,eps,mode", [(300, 0.1, 1), (1000, 0.15, 1), (1100, 0.05, 0), (3000, 0.1, 2), (5000, 0.2, 1), (20000, 0.1, 1)])
def test_generic_width_kernel_on_the_chain_a_first_read_meets(ctx, L, eps, mode):
    """VERDICT r05 'missing' 3: a window's first aligned read has the whole read as its band (bspoa.h:2045-2054, 2109-2111) and k_poa_wf stops at
    256 columns.  A read against the chain of the read before it (C4: 20 000 columns, 20 cells a thread): best end cell, every step of the walk
    and where it ends against the scalar statement (oracle/bsalign_oracle_wf.c) -- one to twenty cells a thread, all three modes, the POA's
    default two-piece scoring"""
    rng = np.random.default_rng(L + mode)
    p = dict(P.par()); p["alnmode"] = mode
    nodes, edges, cands, Q = _chain_program(rng, L, eps, mode)
    bw = (len(Q) + 15) // 16 * 16
    pr, cap = _prog(nodes, edges, cands, len(Q))
    res, ev, _, _ = ctx.poa_graph_host(nodes, edges, cands, pr, Q, _sweep_params(p, bw), cap)
    orows, ou0 = P.oracle_wf_forward(nodes, Q, p, bw)
    ob = P.oracle_wf_best(nodes, cands, len(Q), p, bw, orows)
    r = res[0]
    assert (int(r["maxscr"]), int(r["maxidx"]), int(r["maxoff"])) == (int(ob["maxscr"]), int(ob["maxidx"]), int(ob["maxoff"]))
    n, oev, fin = P.oracle_wf_trace(nodes, edges, Q, p, bw, orows, ou0, 0, int(r["maxidx"]), int(r["maxoff"]))
    assert n > L // 2 and r["status"] == 0 and int(r["nevents"]) == n and np.array_equal(ev[:n], oev)
    assert (int(r["fin_node"]), int(r["fin_x"])) == (int(fin[0]), int(fin[1]))
