"""The graph form of the POA binding (include/bsalign_poa_adapter.h: bsa_poa_flatten_graph, bsa_poa_align_rd_core, bsa_poa_apply_trace)
against the real reference on the CPU, with the oracle's scalar statement of the device kernel (orc_wf_backend) as backend:
* shadow mode (harness mode 5, reference built with the test-only recording hook): for every read the best end cell, EVERY step
  (node, x, bt) of the reference's own alignment2graph_bspoa walk and its end are the backend's;
* product mode (mode 6): nothing of the reference's sweep or walk runs, the binding applies the steps -- consensus, qualities
  and MSA equal the untouched end_bspoa's."""
import numpy as np
import pytest

import poa_support as P
import support as S

pytestmark = pytest.mark.skipif(not (S.have_ref() and P.have_ref_trace()), reason="needs oracle/_ref (the reference build, only in the build container)")

SETS = [P.par(), P.par(alnmode=0), P.par(alnmode=2), P.par(Q=0, P=0), P.par(O=0, E=-3, Q=0, P=0), P.par(bandwidth=64), P.par(bandwidth=256, nrec=3)]


@pytest.mark.parametrize("k", range(len(SETS)))
def test_every_step_of_the_reference_walk(k):
    p = SETS[k]
    reads = P.synth_reads(300 + k, 1200, 10, eps=(0.05, 0.12, 0.2))
    r = P.run_ref_graph(reads, 5, p, record=True, lib=P.ref_poa_trace())
    assert r["bad"] == 0, [(i, rc["mismatch"]) for i, rc in enumerate(r["recs"]) if rc["mismatch"]]
    assert r["graph_reads"] >= len(reads) - 3
    ref = P.run_ref_poa(reads, 0, p, record=False)
    assert np.array_equal(r["cns"], ref["cns"]) and r["msa"] == ref["msa"]
    steps = sum(len(rc["trace"]) for rc in r["recs"] if "trace" in rc)
    assert steps > 5 * 1200


@pytest.mark.parametrize("k", [0, 1, 4, 5])
def test_the_binding_applies_the_walk(k):
    p = SETS[k]
    reads = P.synth_reads(400 + k, 1500, 12, eps=(0.05, 0.12))
    ref = P.run_ref_poa(reads, 0, p, record=False)
    mine = P.run_ref_graph(reads, 6, p, record=False)
    assert mine["graph_reads"] >= len(reads) - 3
    assert np.array_equal(mine["cns"], ref["cns"]) and np.array_equal(mine["qlt"], ref["qlt"]) and np.array_equal(mine["alt"], ref["alt"])
    assert mine["msa"] == ref["msa"]
    for a, b in zip(mine["recs"], ref["recs"]) if ref["recs"] else []:
        assert np.array_equal(a["rs"], b["rs"])
