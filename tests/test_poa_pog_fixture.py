"""The library's own POA graph surface (include/bsalign_poa.h) against tests/golden/poa_pog.npz WITHOUT any reference build (this test also runs where
oracle/_ref is absent): every aligned read of five windows -- import the reference's flat graph once, then per read the selection list of sel_nodes_bspoa,
the band placement and auxiliary edges of prepare_rd_align_bspoa, the program byte for byte, the reference's recorded walk replayed as the backend, the
result of align_rd_bspoa, and after the library's own surgery the whole graph against the reference's graph before the next read."""
import numpy as np
import pytest

import pog_fixture as F


@pytest.mark.parametrize("c", range(5))
def test_replay_of_the_references_windows(c):
    case = F.load()[c]
    st = F.replay_window(case, F.recorded_walk)
    took = sum(1 for sn in case["snaps"][:-1] if sn["best"][5])          # reads the graph form took in the recording run: since round 6 ALL of them (a window's
    declined = len(case["snaps"]) - 1 - took                            # first read, whole-read band above 256 columns, included): one import, the first snapshot
    assert declined == 0 and st["reads"] == took and st["imports"] == 1
    assert st["sel"] > 0 and st["prog_bytes"] > 0 and st["steps"] > 100 * st["reads"]


def test_add_read_builds_the_graph_the_reference_starts_from():
    """bsa_pog_add_read = _add_read_bspoa_core (bspoa.h:916-951): the reads pushed one by one give the first snapshot's structure"""
    from bsalign_amd import poa as PG
    case = F.load()[0]
    sn = case["snaps"][0]
    pog = PG.Pog(**case["par"])
    try:
        for r in range(len(sn["ndoff"])):
            o, n = int(sn["ndoff"][r]), int(sn["rdlen"][r])
            assert pog.add_read(sn["nodes"]["base"][o:o + n]) == r
        assert F.same_structure(pog.export_graph(), sn) is None
    finally:
        pog.close()


def test_misuse_is_refused():
    from bsalign_amd import poa as PG
    import bsalign_amd as B
    pog = PG.Pog()
    try:
        with pytest.raises(B.BsaError):
            pog.add_read(np.array([0, 1, 7], np.uint8))          # a base code above 3
        pog.add_read(np.zeros(0, np.uint8)); pog.add_read(np.array([0, 1, 2, 3] * 10, np.uint8)); pog.add_read(np.array([0, 1, 2, 3] * 10, np.uint8))
        with pytest.raises(B.BsaError):
            pog.place(0)                                         # place before select
        with pytest.raises(B.BsaError):
            pog.select(9, 0, 4)                                  # no such read
        rd, sel = pog.select(1, 0, 40)
        assert rd.nsel == 2 and rd.qlen == 40
        with pytest.raises(B.BsaError):
            pog.select(2, 0, 40)                                 # a read is already being aligned
        pog.abort()
        assert len(pog.aux_edges()) == 0
    finally:
        pog.close()


def test_import_refuses_broken_snapshots():
    """bsa_pog_import is a public entry point fed from fixtures and bindings: offsets that decrease or pass the edge count, an edge
    named twice in an in-list, a header that is not one, a ring that does not close -- BSA_E_ARG, never a write past the arrays or an endless loop"""
    from bsalign_amd import poa as PG
    import bsalign_amd as B
    import copy
    case = F.load()[0]
    good = case["snaps"][2]
    pog = PG.Pog(**case["par"])
    try:
        pog.import_graph(good)                                   # the untouched snapshot is taken

        def broken(edit):
            sn = {k: (v.copy() if hasattr(v, "copy") else copy.copy(v)) for k, v in good.items()}
            edit(sn)
            with pytest.raises(B.BsaError):
                pog.import_graph(sn)

        n = len(good["nodes"])
        broken(lambda sn: sn["out_off"].__setitem__(n // 2, sn["out_off"][n] + 5))          # an out-list reaching past the edges
        broken(lambda sn: sn["in_off"].__setitem__(n // 2, sn["in_off"][n // 2 + 1] + 1))  # decreasing offsets
        broken(lambda sn: sn["out_off"].__setitem__(0, 1))
        k = next(i for i in range(n) if good["in_off"][i + 1] - good["in_off"][i] >= 1 and good["in_off"][i + 1] < good["in_off"][n])
        broken(lambda sn: sn["in_from"].__setitem__(int(sn["in_off"][k + 1]), sn["in_from"][int(sn["in_off"][k])]) or sn["in_off"].__setitem__(k + 1, sn["in_off"][k + 1] + 1))
        m = next(i for i in range(n) if good["nodes"]["header"][i] != i)
        broken(lambda sn: sn["nodes"]["header"].__setitem__(int(good["nodes"]["header"][m]), m))        # the header's header is a member
        broken(lambda sn: sn["nodes"]["next"].__setitem__(m, m))                                        # a ring that never returns to its header
        pog.import_graph(good)
    finally:
        pog.close()
