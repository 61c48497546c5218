"""The library's OWN POA graph surface (include/bsalign_poa.h, bsalign_amd/csrc/bsa_pog.cpp: container, node selection, band placement, program
building, graph surgery -- SURVEY.md section 8(a) rows P0, P2, P3, P6) against the real reference on the CPU.  The entry points are the ones of the
real libbsalign_hip.so (the harness forwards into it; the surface is host code, no GPU needed); the DP + walk between them is the oracle's scalar
statement of the device kernel here, the MI355X in tests/test_poa_pog_gpu.py.
* shadow mode (harness mode 8): on EVERY read the library's selection list equals sel_nodes_bspoa's, its band width / read interval / auxiliary
  edges equal prepare_rd_align_bspoa's, its program equals the binding's flattening of the reference's graph byte for byte (band offsets and
  in-degrees are part of it), its result equals the binding's, and after its own surgery its WHOLE graph -- rings, coverages, flags, every edge
  list in order -- equals the reference's.  The library's graph is built once (bsa_pog_add_read) and never re-imported.
* product mode (mode 9): a patched reference's align_rd_bspoa on this surface -- none of sel_nodes / prepare_rd_align / align_rd_bspoacore /
  alignment2graph runs -- gives the untouched end_bspoa's consensus, qualities and MSA."""
import numpy as np
import pytest

import poa_support as P
import support as S

pytestmark = pytest.mark.skipif(not S.have_ref(), reason="needs oracle/_ref (the reference build, only in the build container)")

SETS = [P.par(), P.par(alnmode=0), P.par(alnmode=2), P.par(Q=0, P=0), P.par(O=0, E=-3, Q=0, P=0), P.par(bandwidth=64), P.par(bandwidth=256, nrec=3),
        P.par(nrec=0), P.par(seqcore=6), P.par(bwtrigger=0, bandwidth=0), P.par(shuffle=0, nrec=2)]


@pytest.mark.parametrize("k", range(len(SETS)))
def test_every_step_in_the_shadow_of_the_reference(k):
    p = SETS[k]
    L = 1200 if p["bandwidth"] else 230          # (bandwidth 0 = whole-read bands: the kernel takes them up to 256 columns)
    reads = P.synth_reads(700 + k, L, 10, eps=(0.05, 0.12, 0.2))
    r = P.run_ref_graph(reads, 8, p, record=False)
    assert r["bad"] == 0, [(i, rc["mismatch"]) for i, rc in enumerate(r["recs"]) if rc["mismatch"]]
    g = r["pog"]
    # the library's graph went through every read but the window's first (aligned against the empty backbone with the whole read as band, which the
    # kernel declines above 256 columns: the one re-import)
    aligned = min(len(reads) + 1, p["seqcore"] or 10 ** 9) - 1          # (beg_bspoa pushes an empty read 0; reads 1 .. nmsa - 1 are aligned in the first stage)
    assert g["declined"] <= 1 and g["imports"] <= g["declined"] and g["reads"] == aligned - g["declined"]
    # something was compared on every read: selections, programs, steps, whole graphs
    assert g["sel_nodes"] > aligned * L // 4 and g["program_bytes"] > 0 and g["steps"] > 0 and g["graph_nodes"] > 0 and g["graph_edges"] > 0
    ref = P.run_ref_poa(reads, 0, p, record=False)
    assert np.array_equal(r["cns"], ref["cns"]) and r["msa"] == ref["msa"]


@pytest.mark.parametrize("k", [0, 1, 4, 5, 7, 8])
def test_a_patched_reference_on_the_librarys_graph(k):
    p = SETS[k]
    reads = P.synth_reads(800 + k, 1500, 12, eps=(0.05, 0.12))
    ref = P.run_ref_poa(reads, 0, p, record=False)
    mine = P.run_ref_graph(reads, 9, p, record=False)
    g = mine["pog"]
    aligned = min(len(reads) + 1, p["seqcore"] or 10 ** 9) - 1
    assert g["declined"] <= 1 and g["reads"] == aligned - g["declined"]
    assert np.array_equal(mine["cns"], ref["cns"]) and np.array_equal(mine["qlt"], ref["qlt"]) and np.array_equal(mine["alt"], ref["alt"])
    assert mine["msa"] == ref["msa"]


# ---- the realn entry of align_rd_bspoa (bspoa.h:2626-2630, cut_rdnode_bspoa :741-795): a stretch of a read that is already in the graph is cut out and
# aligned again.  The reference's only caller (remsa_lsps_bspoa) is compiled out of its main.c, so the harness drives it: after the first stage every
# aligned read has its middle half (1) or all of it (2) re-aligned, on both graphs, and the graphs are compared after the cut and after the surgery.
REALN_SETS = [(P.par(), 1), (P.par(nrec=0, alnmode=0), 1), (P.par(Q=0, P=0, nrec=2), 1), (P.par(bwtrigger=0, bandwidth=0), 2), (P.par(shuffle=0, seqcore=6), 2),
              (P.par(bandwidth=64), 2)]


@pytest.mark.parametrize("k", range(len(REALN_SETS)))
def test_re_aligned_stretches_in_the_shadow_of_the_reference(k):
    p, how = REALN_SETS[k]
    L = 900 if p["bandwidth"] else 230
    if how == 1 and p["bandwidth"]:
        L = 440                                  # (a stretch between two inner nodes gets the whole stretch as its band: 220 columns)
    reads = P.synth_reads(900 + k, L, 9, eps=(0.05, 0.12))
    r = P.run_ref_graph(reads, 8, p, record=False, realn_pass=how)
    aligned = min(len(reads) + 1, p["seqcore"] or 10 ** 9) - 1
    assert len(r["recs"]) == 2 * aligned                     # every read once, then a stretch of every read again
    assert r["bad"] == 0, [(i, rc["mismatch"]) for i, rc in enumerate(r["recs"]) if rc["mismatch"]]
    g = r["pog"]
    assert g["imports"] <= g["declined"] <= 1 and g["reads"] == 2 * aligned - g["declined"]


def _sam_cigars(reads, rng, clip=True):
    """reads[0] = the reference; every other read's CIGAR against it from the oracle's global edit alignment, some with soft / hard clips at the ends
    (the read gets the clipped bases), some covering only a part of the reference (leading / trailing D)"""
    T = reads[0]
    out_reads, cigs = [T], [np.zeros(0, np.uint32)]
    for q in reads[1:]:
        kind = int(rng.integers(0, 4)) if clip else 0
        lo, hi = 0, len(T)
        if kind >= 2:                            # a read over the middle of the reference: margins as D (2) or as H / N (3), the three ops the reference counts alike
            lo, hi = len(T) // 5, len(T) - len(T) // 7
            q = q[len(q) // 5: len(q) - len(q) // 7]
        res, cg, n = S.oracle_edit(q, T[lo:hi], 0, 0)
        assert n > 0
        cg = [int(x) for x in cg]
        if lo:
            cg = [(lo << 4) | (2 if kind == 2 else 5)] + cg + [((len(T) - hi) << 4) | (2 if kind == 2 else 3)]
        if kind == 1:                            # soft clips: junk bases on the read
            a, b = int(rng.integers(1, 30)), int(rng.integers(1, 30))
            q = np.concatenate([rng.integers(0, 4, a).astype(np.uint8), q, rng.integers(0, 4, b).astype(np.uint8)])
            cg = [(a << 4) | 4] + cg + [(b << 4) | 4]
        out_reads.append(np.asarray(q, dtype=np.uint8)); cigs.append(np.array(cg, dtype=np.uint32))
    return out_reads, cigs


@pytest.mark.parametrize("k", range(6))
def test_refmode_in_the_shadow_of_the_reference(k):
    """refmode (bspoa.h:2039-2085): read 0 is a reference sequence; bands are placed by the reads' SAM CIGARs against it (k even) or, without CIGARs,
    by the guide alignment against it (k odd)."""
    p = [P.par(shuffle=0), P.par(shuffle=0), P.par(shuffle=0, nrec=0, alnmode=0), P.par(shuffle=0, bandwidth=64), P.par(shuffle=0, Q=0, P=0), P.par(shuffle=1)][k]
    rng = np.random.default_rng(4100 + k)
    T = rng.integers(0, 4, size=1100).astype(np.uint8)
    reads = [T] + [S.mutate(rng, T, float(rng.choice((0.04, 0.1)))) for _ in range(8)]
    reads = [np.asarray(x, dtype=np.uint8) for x in reads]
    cigs = None
    if k % 2 == 0:
        reads, cigs = _sam_cigars(reads, rng)
    r = P.run_ref_graph(reads, 8, p, record=False, refmode=1, cigars=cigs)
    assert r["bad"] == 0, [(i, rc["mismatch"]) for i, rc in enumerate(r["recs"]) if rc["mismatch"]]
    g = r["pog"]
    assert len(r["recs"]) == len(reads) - 1 and g["reads"] == len(reads) - 1 and g["declined"] == 0 and g["imports"] == 0
    assert g["sel_nodes"] > 0 and g["steps"] > 0 and g["graph_edges"] > 0


@pytest.mark.parametrize("k", [0, 2, 3, 4])
def test_a_patched_reference_re_aligns_on_the_librarys_graph(k):
    """product mode with the realn pass: the patched align_rd_bspoa's path (the reference's cut, then bsa_poa_align_rd_pog with its realn flag) against the
    untouched align_rd_bspoa(.., realn = 1, ..) driven the same way -- same consensus, qualities, MSA."""
    p, how = REALN_SETS[k]
    L = 900 if p["bandwidth"] else 230
    if how == 1 and p["bandwidth"]:
        L = 440
    reads = P.synth_reads(950 + k, L, 9, eps=(0.05, 0.12))
    ref = P.run_ref_graph(reads, 1, p, record=False, realn_pass=how)
    mine = P.run_ref_graph(reads, 9, p, record=False, realn_pass=how)
    aligned = min(len(reads) + 1, p["seqcore"] or 10 ** 9) - 1
    assert len(ref["recs"]) == len(mine["recs"]) == 2 * aligned
    for a, b in zip(ref["recs"], mine["recs"]):
        assert np.array_equal(a["rs"][:9], b["rs"][:9])
    g = mine["pog"]
    assert g["declined"] <= 1 and g["reads"] == 2 * aligned - g["declined"]
    assert np.array_equal(mine["cns"], ref["cns"]) and np.array_equal(mine["qlt"], ref["qlt"]) and np.array_equal(mine["alt"], ref["alt"]) and mine["msa"] == ref["msa"]


@pytest.mark.parametrize("k", [0, 1, 2, 5])
def test_a_patched_reference_in_refmode_on_the_librarys_graph(k):
    p = [P.par(shuffle=0), P.par(shuffle=0), P.par(shuffle=0, nrec=0, alnmode=0), None, None, P.par(shuffle=1)][k]
    rng = np.random.default_rng(4200 + k)
    T = rng.integers(0, 4, size=1300).astype(np.uint8)
    reads = [T] + [np.asarray(S.mutate(rng, T, float(rng.choice((0.04, 0.1)))), dtype=np.uint8) for _ in range(9)]
    cigs = None
    if k % 2 == 0:
        reads, cigs = _sam_cigars(reads, rng)
    ref = P.run_ref_graph(reads, 1, p, record=False, refmode=1, cigars=cigs)
    mine = P.run_ref_graph(reads, 9, p, record=False, refmode=1, cigars=cigs)
    g = mine["pog"]
    assert g["declined"] == 0 and g["reads"] == len(reads) - 1
    for a, b in zip(ref["recs"], mine["recs"]):
        assert np.array_equal(a["rs"][:9], b["rs"][:9])
    assert np.array_equal(mine["cns"], ref["cns"]) and np.array_equal(mine["qlt"], ref["qlt"]) and np.array_equal(mine["alt"], ref["alt"]) and mine["msa"] == ref["msa"]
