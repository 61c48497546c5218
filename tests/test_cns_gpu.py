"""GPU parity of the device consensus caller (bsa_msa_call_consensus_batch, bsa_cns_dev.hip; SURVEY 8(f) rank 4: cns_bspoa,
bspoa.h:3457-3733) against the host form (bsa_cns.cpp, bit-identical to the reference), the committed results of the real cns_bspoa
(tests/golden/cns.npz) and, where oracle/_ref travelled, 256 windows of the real end_bspoa.

Tolerance (floating point, stated here as the task asks): the sums over the reads are formed in the host's order from the host's own
logarithm tables -- identical bits; the merges log(exp a + exp b) and the quality formulas use the device's exp / log / erfc, which may
differ from glibc's in the last place.  Required: consensus bases, both quality strings and the three consensus bytes of EVERY column
byte for byte; the log probability of the best path to a relative 1e-12."""
import os

import numpy as np
import pytest

import support as S

pytestmark = pytest.mark.gpu

REL = 1e-12


def _same(dev, host_cns, host_qlt, host_alt, host_score, host_cols, what):
    cns, qlt, alt, score, cols = dev
    assert np.array_equal(cns, host_cns), what
    assert np.array_equal(qlt, host_qlt), what
    assert np.array_equal(alt, host_alt), what
    assert np.array_equal(cols, host_cols.reshape(-1)), what
    assert abs(score - host_score) <= REL * max(1.0, abs(host_score)), (what, score, host_score)


def test_fixture_of_the_real_cns_bspoa(ctx):
    """tests/golden/cns.npz: MSAs of real windows and what the reference's cns_bspoa made of them -- the three windows as ONE batch"""
    from bsalign_amd import msa as MSA
    g = np.load(os.path.join(S.ROOT, "tests", "golden", "cns.npz"))
    n = int(g["n"][0])
    wins = []
    for k in range(n):
        nmsa, nrds, nall, mlen = (int(x) for x in g["dims_%d" % k])
        mine = g["cols_%d" % k].copy(); mine[:, nall:] = 255            # nothing of the reference's answer left in the input
        wins.append((mine, None, nall, min(nmsa, nrds), nrds, mlen))
    assert all(np.array_equal(g["par_0"], g["par_%d" % k]) for k in range(n))
    out = MSA.call_consensus_batch(ctx, wins, g["par_0"])
    for k in range(n):
        _same(out[k], g["cns_%d" % k], g["qlt_%d" % k], g["alt_%d" % k], float(g["score_%d" % k][0]), g["cols_%d" % k], "fixture window %d" % k)


def _random_msa(rng, nall, mlen, template_eps=0.12, outside=0.1, ncols_extra=0):
    """a plausible MSA on plain arrays: a hidden consensus row (bases and gaps), reads that copy it with errors and are outside
    the window at their ends (codes 5 / 6)"""
    ncols = mlen + ncols_extra
    cols = np.full((ncols, nall + 3), 255, np.uint8)
    truth = rng.choice(5, size=ncols, p=[0.2, 0.2, 0.2, 0.2, 0.2]).astype(np.uint8)
    for r in range(nall):
        row = truth.copy()
        err = rng.random(ncols) < template_eps
        row[err] = rng.integers(0, 5, size=int(err.sum()))
        if rng.random() < 0.5:
            a = int(rng.integers(0, max(1, int(ncols * outside))))
            row[:a] = 5
        if rng.random() < 0.5:
            b = int(rng.integers(0, max(1, int(ncols * outside))))
            if b:
                row[-b:] = 6
        cols[:, r] = row
    return cols


@pytest.mark.parametrize("shape", [(3, 1), (5, 63), (12, 64), (12, 65), (64, 700), (70, 300), (150, 257), (9, 9000), (1, 40)])
def test_random_msas_equal_the_host_form(ctx, shape):
    """device = host (bsa_cns.cpp) on synthetic MSAs: few and many reads (more than a wave's 64: the per-read work in two or three
    trips), one column, more columns than one piece of the traceback chain (8192), a permuted column order, reads that do not vote
    (nseq < nmax < nall), non-default error rates"""
    from bsalign_amd import msa as MSA
    nall, mlen = shape
    rng = np.random.default_rng(1000 * nall + mlen)
    par = np.array([0.10, 0.10, 0.15, 0.15, 0.20, 0.20, 0.40], np.float32)
    wins, refs = [], []
    for v in range(4):
        extra = 7 if v >= 2 else 0
        cols = _random_msa(rng, nall, mlen, template_eps=[0.05, 0.12, 0.3, 0.12][v], ncols_extra=extra)
        idxs = rng.permutation(mlen + extra)[:mlen].astype(np.uint32) if v >= 2 else None
        nseq = nall if v != 3 else max(1, nall - nall // 3)
        nmax = nall if v != 3 else max(nseq, nall - 1)
        wins.append((cols.copy(), idxs, nall, nseq, nmax, mlen))
        h = cols.copy()
        cns, qlt, alt, score = MSA.call_consensus(h, idxs, nall, nseq, nmax, mlen, par)
        refs.append((cns, qlt, alt, score, h))
    out = MSA.call_consensus_batch(ctx, wins, par)
    for v in range(4):
        _same(out[v], *refs[v], "shape %s variant %d" % (shape, v))
    par2 = np.array([0.05, 0.02, 0.3, 0.1, 0.5, 0.25, 0.1], np.float32)
    h = wins[1][0].copy()
    want = MSA.call_consensus(h, None, nall, nall, nall, mlen, par2)
    got = MSA.call_consensus_batch(ctx, [(wins[1][0].copy(), None, nall, nall, nall, mlen)], par2)
    _same(got[0], *want, h, "other rates")


def test_empty_batch_and_empty_window(ctx):
    from bsalign_amd import msa as MSA
    par = np.array([0.10, 0.10, 0.15, 0.15, 0.20, 0.20, 0.40], np.float32)
    assert MSA.call_consensus_batch(ctx, [], par) == []
    rng = np.random.default_rng(5)
    cols = _random_msa(rng, 6, 50)
    out = MSA.call_consensus_batch(ctx, [(np.zeros(0, np.uint8), None, 4, 4, 4, 0), (cols.copy(), None, 6, 6, 6, 50)], par)
    assert len(out[0][0]) == 0 and out[0][3] == 0.0
    h = cols.copy()
    _same(out[1], *MSA.call_consensus(h, None, 6, 6, 6, 50, par), h, "window behind an empty one")


@pytest.mark.skipif(not S.have_ref(), reason="needs oracle/_ref")
def test_256_live_windows_of_the_real_end_bspoa(ctx):
    """256 windows of synthetic reads through the REAL end_bspoa (oracle/_ref); their final MSAs as one device batch: consensus, both
    quality strings and every column's consensus bytes equal what the reference's own cns_bspoa left in its BSPOA"""
    import ctypes as C
    import msa_support as MS
    import poa_support as P
    from bsalign_amd import msa as MSA
    wins, want = [], []
    rng = np.random.default_rng(256)
    par7 = None
    for k in range(256):
        L = int(rng.choice([90, 200, 300, 500, 800]))
        n = int(rng.choice([3, 5, 8, 12, 20, 30]))
        eps = (float(rng.choice([0.02, 0.08, 0.15])),)
        if k == 7:
            L, n = 400, 70                      # deeper than a wave and than seqcore: the normal-tail branch of the alternative quality
        w = MS.RefWindow(P.synth_reads(1000 + k, L, n, eps=eps))
        try:
            r = w.r
            r.ref_poa_cns_call.argtypes = [C.c_void_p]; r.ref_poa_cns_call.restype = C.c_double
            r.ref_poa_cns_inputs.argtypes = [C.c_void_p] * 5; r.ref_poa_cns_inputs.restype = None
            nmsa, nrds, nall = C.c_uint32(), C.c_uint32(), C.c_uint32()
            p7 = np.zeros(7, np.float32)
            r.ref_poa_cns_inputs(w.h, C.byref(nmsa), C.byref(nrds), C.byref(nall), p7.ctypes.data)
            score = r.ref_poa_cns_call(w.h)
            assert par7 is None or np.array_equal(par7, p7)
            par7 = p7
            mine = w.cols.copy()
            mine.reshape(-1, w.mrow)[:, nall.value:] = 255
            wins.append((mine, w.idxs.copy(), nall.value, min(nmsa.value, nrds.value), nrds.value, w.mlen))
            want.append((w.cns.copy(), w.qlt.copy(), w.alt.copy(), score, w.cols.copy(), w.idxs.astype(np.int64), w.mrow))
        finally:
            w.close()
    out = MSA.call_consensus_batch(ctx, wins, par7)
    for k in range(256):
        cns, qlt, alt, score, cols = out[k]
        wc, wq, wa, ws, wcols, used, mrow = want[k]
        assert np.array_equal(cns, wc) and np.array_equal(qlt, wq) and np.array_equal(alt, wa), k
        assert np.array_equal(cols.reshape(-1, mrow)[used], wcols.reshape(-1, mrow)[used]), k
        assert abs(score - ws) <= REL * max(1.0, abs(ws)), (k, score, ws)
