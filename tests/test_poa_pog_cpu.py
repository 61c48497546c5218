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
