"""GPU parity: the device POA sweep (bsa_sweep_host / bsa_sweep_run: flattened align_rd_bspoacore programs) against
the reference fixtures (best end cell + hash of the reference's row blocks) and, byte for byte, against the oracle."""
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


def _used(rows, nblocks, bw, pw):
    blk, used = P.block_bytes(bw, pw), bw * (pw + 1) + 68
    return rows[:nblocks * blk].reshape(nblocks, blk)[2:, :used]


def test_golden_programs_one_by_one(ctx):
    for case in P.load_golden():
        p = case["par"]
        for k, pg in enumerate(case["programs"]):
            progs = np.array([(0, pg["ntasks"], 0, 0)], dtype=P.PROG_DTYPE)
            rows, res = ctx.sweep_host(pg["tasks"], progs, pg["query"], np.zeros(1, np.uint64), np.array([pg["slen"]], np.uint32),
                                       _sweep_params(p, pg["bandwidth"]), pg["nblocks"])
            got = (int(res[0]["maxscr"]), int(res[0]["maxidx"]), int(res[0]["maxoff"]))
            assert got == (pg["maxscr"], pg["maxidx"], pg["maxoff"]), (p, k, got)
            orows, _ = P.oracle_sweep(pg["tasks"], progs, pg["query"], np.zeros(1, np.uint64), np.array([pg["slen"]], np.uint32), p,
                                      pg["bandwidth"], pg["nblocks"], pg["piecewise"])
            a, b = _used(rows, pg["nblocks"], pg["bandwidth"], pg["piecewise"]), _used(orows, pg["nblocks"], pg["bandwidth"], pg["piecewise"])
            bad = np.nonzero((a != b).any(axis=1))[0]
            assert len(bad) == 0, (p, k, "first differing block", int(bad[0]) + 2, pg["bandwidth"])
            assert P.hash_node_blocks(rows, pg["nblocks"], pg["bandwidth"], pg["piecewise"], pg["tasks"]) == pg["rows_hash"], (p, k)


def test_golden_programs_batched(ctx):
    """all programs of one case that share a bandwidth in ONE launch (many windows side by side)"""
    for case in P.load_golden():
        p = case["par"]
        by_bw = {}
        for pg in case["programs"]:
            by_bw.setdefault(pg["bandwidth"], []).append(pg)
        for bw, pgs in by_bw.items():
            reps = 5                                   # replicate to fill several wavefronts
            tasks, progs, queries, qoff, qlen = [], [], [], [], []
            tacc = bacc = qacc = 0
            for r in range(reps):
                for pg in pgs:
                    t = pg["tasks"].copy()
                    t["query"] = len(qlen)
                    tasks.append(t)
                    progs.append((tacc, pg["ntasks"], bacc, 0))
                    tacc += pg["ntasks"]
                    bacc += pg["nblocks"]
                    queries.append(pg["query"])
                    qoff.append(qacc)
                    qlen.append(pg["slen"])
                    qacc += pg["slen"]
            tasks = np.concatenate(tasks)
            progs = np.array(progs, dtype=P.PROG_DTYPE)
            rows, res = ctx.sweep_host(tasks, progs, np.concatenate(queries), np.array(qoff, np.uint64), np.array(qlen, np.uint32),
                                       _sweep_params(p, bw), bacc)
            pw = pgs[0]["piecewise"]
            blk = P.block_bytes(bw, pw)
            for i, pr in enumerate(progs):
                pg = pgs[i % len(pgs)]
                assert (int(res[i]["maxscr"]), int(res[i]["maxidx"]), int(res[i]["maxoff"])) == (pg["maxscr"], pg["maxidx"], pg["maxoff"]), (p, bw, i)
                sub = rows[int(pr["first_block"]) * blk:(int(pr["first_block"]) + pg["nblocks"]) * blk]
                assert P.hash_node_blocks(sub, pg["nblocks"], bw, pw, pg["tasks"]) == pg["rows_hash"], (p, bw, i)


def _chain_program(rng, n, bw, slen, bubbles=True):
    """a synthetic sub-graph: a backbone of n nodes with optional two-way bubbles, band offsets advancing ~1 per node"""
    tasks = [(2, 0, 2, 0, 0, 0, 0, 0, 0, 0)]
    blk_of, rpos, mpos = 2, 0, -1
    nxt = 3
    for i in range(n):
        base = int(rng.integers(4))
        step = int(rng.choice([0, 1, 1, 1, 2]))
        nr = min(max(rpos + step, 0), max(slen - bw, 0))
        if bubbles and i % 7 == 3:
            # two alternative nodes, then a merge node
            a, b, m = nxt, nxt + 1, nxt + 2
            nxt += 3
            tasks.append((0, blk_of, a, rpos, nr, mpos + 2, 0, base, int(rng.integers(4)), 0))
            tasks.append((0, blk_of, b, rpos, nr, mpos + 2, 0, (base + 1) & 3, int(rng.integers(4)), 0))
            nr2 = min(nr + 1, max(slen - bw, 0))
            mb = int(rng.integers(4))
            tasks.append((0, a, m, nr, nr2, mpos + 3, 0, mb, int(rng.integers(4)), 0))
            tasks.append((0, b, 1, nr, nr2, mpos + 3, 0, mb, int(rng.integers(4)), 0))
            tasks.append((1, 1, m, 0, 0, 0, 0, 0, 0, 0))
            blk_of, rpos, mpos = m, nr2, mpos + 2
        else:
            tasks.append((0, blk_of, nxt, rpos, nr, mpos + 2, 0, base, int(rng.integers(4)), 0))
            blk_of, rpos, mpos = nxt, nr, mpos + 1
            nxt += 1
        if rpos + bw >= slen and i % 5 == 0:
            tasks.append((4, blk_of, 0, rpos, 0, 1000 + i, 0, 0, 0, 0))
    tasks.append((3, blk_of, 0, rpos, 0, 7777, 0, 0, 0, 0))
    return np.array(tasks, dtype=P.TASK_DTYPE), nxt


@pytest.mark.parametrize("bw", [16, 64, 128, 176, 256])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_synthetic_programs_match_oracle(ctx, bw, mode):
    rng = np.random.default_rng(bw * 5 + mode)
    p = P.par(alnmode=mode) if bw != 64 else P.par(alnmode=mode, Q=0, P=0)
    nprog = 40
    tasks, progs, queries, qoff, qlen = [], [], [], [], []
    tacc = bacc = qacc = 0
    for k in range(nprog):
        slen = int(rng.integers(bw + 50, bw + 900))
        t, nb = _chain_program(rng, int(rng.integers(50, 700)), bw, slen)
        t["query"] = k
        tasks.append(t)
        progs.append((tacc, len(t), bacc, 0))
        tacc += len(t)
        bacc += nb
        queries.append(rng.integers(0, 4, size=slen).astype(np.uint8))
        qoff.append(qacc)
        qlen.append(slen)
        qacc += slen
    tasks, progs = np.concatenate(tasks), np.array(progs, dtype=P.PROG_DTYPE)
    queries, qoff, qlen = np.concatenate(queries), np.array(qoff, np.uint64), np.array(qlen, np.uint32)
    pw = S.oracle().orc_get_piecewise(p["O"], p["E"], p["Q"], p["P"], bw)
    rows, res = ctx.sweep_host(tasks, progs, queries, qoff, qlen, _sweep_params(p, bw), bacc)
    orows, ores = P.oracle_sweep(tasks, progs, queries, qoff, qlen, p, bw, bacc, pw)
    assert np.array_equal(res, ores), [(i, res[i], ores[i]) for i in range(nprog) if res[i] != ores[i]][:3]
    blk, used = P.block_bytes(bw, pw), bw * (pw + 1) + 68
    for i, pr in enumerate(progs):
        b0 = int(pr["first_block"])
        nb = (int(progs[i + 1]["first_block"]) if i + 1 < nprog else bacc) - b0
        a = rows[b0 * blk:(b0 + nb) * blk].reshape(nb, blk)[1:, :used]        # block 1 (merge temp) included, block 0 is scratch
        b = orows[b0 * blk:(b0 + nb) * blk].reshape(nb, blk)[1:, :used]
        bad = np.nonzero((a != b).any(axis=1))[0]
        assert len(bad) == 0, (bw, mode, i, "first differing block", int(bad[0]) + 1)
