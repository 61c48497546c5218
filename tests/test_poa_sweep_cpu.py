"""CPU: the oracle's POA sweep (orc_sweep_run) against the committed reference fixtures, and -- when the reference build
is present (build container) -- the whole end_bspoa with the reference's align_rd_bspoacore replaced by
include/bsalign_poa_adapter.h + the oracle sweep."""
import numpy as np
import pytest

import poa_support as P
import support as S


def _run_case_programs(case, runner):
    p = case["par"]
    for k, pg in enumerate(case["programs"]):
        progs = np.array([(0, pg["ntasks"], 0, 0)], dtype=P.PROG_DTYPE)
        rows, res = runner(pg["tasks"], progs, pg["query"], np.zeros(1, np.uint64), np.array([pg["slen"]], np.uint32), p,
                           pg["bandwidth"], pg["nblocks"], pg["piecewise"])
        got = (int(res[0]["maxscr"]), int(res[0]["maxidx"]), int(res[0]["maxoff"]))
        assert got == (pg["maxscr"], pg["maxidx"], pg["maxoff"]), (p, k, got)
        assert P.hash_node_blocks(rows, pg["nblocks"], pg["bandwidth"], pg["piecewise"], pg["tasks"]) == pg["rows_hash"], (p, k)


def test_oracle_sweep_matches_reference_fixtures():
    cases = P.load_golden()
    assert len(cases) >= 9
    for case in cases:
        _run_case_programs(case, P.oracle_sweep)


def test_fixture_block_size_rule():
    """mmblk = roundup(bw * (piecewise + 1) + 17 * 4, 16) (bspoa.h:2217)"""
    assert P.block_bytes(128, 2) == 464 and P.block_bytes(128, 1) == 336 and P.block_bytes(16, 0) == 96


@pytest.mark.skipif(not S.have_ref(), reason="reference build (oracle/_ref) only exists in the build container")
@pytest.mark.parametrize("kw", [dict(), dict(bandwidth=64, alnmode=0), dict(nrec=2), dict(Q=0, P=0, alnmode=2)])
def test_end_bspoa_with_adapter_equals_reference(kw):
    p = P.par(**kw)
    reads = P.synth_reads(900 + len(kw), 700, 9)
    r0 = P.run_ref_poa(reads, 0, p, record=False)
    r1 = P.run_ref_poa(reads, 1, p, record=False)
    r2 = P.run_ref_poa(reads, 2, p, record=False)
    assert r2["bad"] == 0
    for r in (r1, r2):
        assert np.array_equal(r0["cns"], r["cns"]) and np.array_equal(r0["qlt"], r["qlt"]) and np.array_equal(r0["alt"], r["alt"])
        assert r0["msa"] == r["msa"]
    for a, b in zip(r1["recs"], r2["recs"]):
        assert np.array_equal(a["rs"], b["rs"]) and a["rows_hash"] == b["rows_hash"]
        assert (a["maxscr"], a["maxidx"], a["maxoff"]) == (b["maxscr"], b["maxidx"], b["maxoff"])
