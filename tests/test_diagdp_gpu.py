"""Anti-diagonal u8 DP of the MSA refinement on the device (bsa_diagdp_batch, k_diagdp_stage + k_diagdp_fill) against the
oracle's restatement and the committed reference fixture: every row the DP writes, guard cells included, bit for bit."""
import hashlib
import os

import numpy as np
import pytest

import diag_support as D
from test_diagdp_cpu import CASES, GOLD, _case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import bsalign_amd as B
    c = B.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("k", range(len(CASES)))
def test_device_equals_oracle_and_reference_fixture(ctx, k):
    planes, probs, nbytes = _case(k)
    got = D.written_rows(ctx.diagdp_batch(planes, D.to_struct(probs), nbytes), probs)
    want = D.written_rows(D.oracle_fill(planes, probs, nbytes), probs)
    assert np.array_equal(got, want)
    g = np.load(GOLD)
    assert hashlib.sha256(got.tobytes()).hexdigest() == str(g["rows_sha_%d" % k])


@pytest.mark.parametrize("W", [1, 2, 4])
def test_many_windows_of_ragged_reads(ctx, W):
    """several windows in one call: different MSA lengths, reads that start and end anywhere, a read of two bases"""
    rng = np.random.default_rng(77 + W)
    blobs, allp, off = [], [], 0
    for w in range(6):
        planes, probs = D.make_window(rng, int(rng.choice([40, 129, 333, 1000])), int(rng.integers(2, 12)), W, float(rng.choice([0.05, 0.2])), bool(w % 2))
        for p in probs:
            for f in ("seq0", "seq1"):
                p[f] += off
            p["mats0"] = [o + off for o in p["mats0"]]
            p["mats1"] = [o + off for o in p["mats1"]]
        blobs.append(planes)
        allp += probs
        off += planes.size
    planes = np.concatenate(blobs)
    nbytes = D.matrix_layout(allp)
    got = D.written_rows(ctx.diagdp_batch(planes, D.to_struct(allp), nbytes), allp)
    assert np.array_equal(got, D.written_rows(D.oracle_fill(planes, allp, nbytes), allp))


def test_c4_shaped_window(ctx):
    """64 reads over a 22 k-column MSA (the refinement of one C4 window, editbw 64 -> band 32): the device against the
    oracle on every row; prints the device time beside the oracle's"""
    import time
    rng = np.random.default_rng(4)
    planes, probs = D.make_window(rng, 22000, 64, 2, 0.12, False)
    nbytes = D.matrix_layout(probs)
    t0 = time.time()
    want = D.written_rows(D.oracle_fill(planes, probs, nbytes), probs)
    t_cpu = time.time() - t0
    got = D.written_rows(ctx.diagdp_batch(planes, D.to_struct(probs), nbytes), probs)
    assert np.array_equal(got, want)
    steps = sum(2 * (p["mend"] - p["mbeg"]) - 1 for p in probs)
    t_ref = float("nan")
    import support as S
    if S.have_ref():
        t0 = time.time()
        ref_rows = D.written_rows(D.ref_fill(planes, probs, nbytes), probs)
        t_ref = time.time() - t0
        assert np.array_equal(got, ref_rows)
    print("\n[diag DP, 64 reads x 22 k columns, band 32] %d steps: device kernels %.2f ms, the reference's SSE code on one core %.0f ms, scalar oracle %.0f ms"
          % (steps, ctx.diagdp_last_ms(), t_ref * 1e3, t_cpu * 1e3))


@pytest.mark.parametrize("W", [1, 2, 4])
def test_walk_on_the_device_equals_the_scalar_walk(ctx, W):
    """bsa_diagdp_walk_batch: fill AND traceback on the device, only two bits a step come back -- every step, the score and the end of
    the walk equal the scalar statement of remsa_pedit_rd_bspoacore's traceback run over the oracle's planes (several ragged windows in one call)"""
    rng = np.random.default_rng(300 + W)
    blobs, allp, off = [], [], 0
    for w in range(5):
        planes, probs = D.make_window(rng, int(rng.choice([40, 129, 333, 1000])), int(rng.integers(2, 12)), W, float(rng.choice([0.05, 0.2])), bool(w % 2))
        for p in probs:
            for f in ("seq0", "seq1"):
                p[f] += off
            p["mats0"] = [o + off for o in p["mats0"]]
            p["mats1"] = [o + off for o in p["mats1"]]
        blobs.append(planes)
        allp += probs
        off += planes.size
    planes = np.concatenate(blobs)
    nbytes = D.matrix_layout(allp)
    want = D.oracle_walk(planes, allp, D.oracle_fill(planes, allp, nbytes))
    walks, steps = ctx.diagdp_walk_batch(planes, D.to_struct(allp))
    for k, (st, sc, xe, ye, rc) in enumerate(want):
        assert rc == 0 and int(walks[k]["status"]) == 0, k
        assert (int(walks[k]["score"]), int(walks[k]["xi"]), int(walks[k]["yi"])) == (sc, xe, ye), k
        assert np.array_equal(steps[k], st), k


def test_walk_of_64_windows_in_one_call(ctx):
    """what the host-pointer fill cannot do: 64 windows x 24 reads x 4 k columns in ONE call -- the planes (0.85 GB) stay on the device, 3 MB of
    steps come back; a sample of the reads against the scalar walk"""
    import time
    rng = np.random.default_rng(9)
    planes, probs = D.make_window(rng, 4000, 24, 2, 0.12, False)
    nbytes = D.matrix_layout(probs)
    want = D.oracle_walk(planes, probs, D.oracle_fill(planes, probs, nbytes))
    nwin = 64
    big = np.tile(planes, nwin)
    allp = []
    for w in range(nwin):
        for p in probs:
            q = dict(p)
            q["seq0"] += w * planes.size; q["seq1"] += w * planes.size
            q["mats0"] = [o + w * planes.size for o in p["mats0"]]; q["mats1"] = [o + w * planes.size for o in p["mats1"]]
            allp.append(q)
    t0 = time.time()
    walks, steps = ctx.diagdp_walk_batch(big, D.to_struct(allp))
    dt = time.time() - t0
    assert int((walks["status"] != 0).sum()) == 0
    for w in (0, 17, 63):
        for k, (st, sc, xe, ye, rc) in enumerate(want):
            assert np.array_equal(steps[w * len(probs) + k], st) and int(walks[w * len(probs) + k]["score"]) == sc
    print("\n[diag DP + walk, %d windows x %d reads x 4 k columns in one call] device kernels %.1f ms, the call %.2f s, %.1f MB of steps back (the planes: %.2f GB)"
          % (nwin, len(probs), ctx.diagdp_last_ms(), dt, sum(len(s) for s in steps) / 4e6, nwin * nbytes / 1e9))


def test_bad_arguments(ctx):
    import bsalign_amd as B
    planes, probs = D.make_window(np.random.default_rng(1), 100, 2, 1)
    nbytes = D.matrix_layout(probs)
    st = D.to_struct(probs)
    st[0]["seq0"] = 3                                    # no padding in front of the plane
    with pytest.raises(B.BsaError):
        ctx.diagdp_batch(planes, st, nbytes)
    st = D.to_struct(probs)
    st[1]["W"] = 2                                       # mixed band widths
    with pytest.raises(B.BsaError):
        ctx.diagdp_batch(planes, st, nbytes)
