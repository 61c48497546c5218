"""GPU parity: HIP 8-bit banded path (through the C-ABI) vs the oracle, bit-exact."""
import numpy as np
import pytest

import support as S

pytestmark = pytest.mark.gpu

# scores outside the static exact-arithmetic guard (m + 3 g <= 64, n + m + g <= 100): lane-exact kernels, or the checked whole-query kernel
BIG_SCORINGS = {
    "big": (10, -30, -20, -10, 0, 0),
    "biglinear": (9, -25, 0, -24, 0, 0),
    "big2piece": (6, -20, -12, -8, -30, -3),
}

SCORINGS = {
    "affine": (2, -6, -3, -2, 0, 0),
    "paper": (2, -2, -4, -2, 0, 0),
    "linear": (2, -6, 0, -3, 0, 0),
    "twopiece": (2, -6, -3, -2, -8, -1),
}


def _mk_pairs(rng, n, lens, eps_list=(0.01, 0.1, 0.2), ratios=(1.0, 1.0, 0.9, 1.1)):
    pairs = []
    for _ in range(n):
        L = int(rng.choice(lens))
        T = rng.integers(0, 4, size=L).astype(np.uint8)
        Q = S.mutate(rng, T, float(rng.choice(eps_list)))
        r = float(rng.choice(ratios))
        if r != 1.0:
            Lq = max(1, int(len(Q) * r))
            Q = Q[:Lq] if Lq <= len(Q) else np.concatenate([Q, rng.integers(0, 4, size=Lq - len(Q)).astype(np.uint8)])
        if len(Q) == 0:
            Q = np.array([0], dtype=np.uint8)
        pairs.append((Q, T))
    return pairs


def _check(ctx, pairs, mode, bw, sc):
    import bsalign_amd as B
    par = B.make_params(mode, bw, *sc)
    out, cigs, status = ctx.align_batch(pairs, par)
    nbad = 0
    msgs = []
    for k, (q, t) in enumerate(pairs):
        res, cig, n = S.oracle_align(q, t, mode, bw, *sc)
        if n == S.ORC_ERR_TRACE:
            ok = bool(status[k] & B.ST_TRACE)
        else:
            got = np.array([out[k][f] for f in out.dtype.names], dtype=np.int32)
            ok = status[k] == 0 and np.array_equal(got, res) and np.array_equal(cigs[k], cig)
        if not ok:
            nbad += 1
            if len(msgs) < 5:
                msgs.append("pair %d qlen %d tlen %d status %d\n  gpu %s %s\n  orc %s %s" % (
                    k, len(q), len(t), status[k], out[k], S.cigar_str(cigs[k])[:120], res, S.cigar_str(cig)[:120]))
    assert nbad == 0, "%d/%d pairs differ (mode %d bw %d sc %s)\n%s" % (nbad, len(pairs), mode, bw, sc, "\n".join(msgs))


@pytest.mark.parametrize("scname", list(SCORINGS))
@pytest.mark.parametrize("mode", [S.MODE_GLOBAL, S.MODE_OVERLAP, S.MODE_EXTEND])
@pytest.mark.parametrize("bw", [128, 64])
def test_random_pairs(ctx, scname, mode, bw):
    rng = np.random.default_rng(1000 + bw + 7 * mode + len(scname))
    pairs = _mk_pairs(rng, 96, [1, 15, 16, 17, 63, 64, 65, 100, 300, 1000, 2000])
    _check(ctx, pairs, mode, bw, SCORINGS[scname])


@pytest.mark.parametrize("bw", [16, 32, 256, 512])
def test_other_bandwidths(ctx, bw):
    rng = np.random.default_rng(77 + bw)
    pairs = _mk_pairs(rng, 64, [10, 100, 700, 1500])
    for mode in (S.MODE_GLOBAL, S.MODE_OVERLAP):
        _check(ctx, pairs, mode, bw, SCORINGS["affine"])
        _check(ctx, pairs, mode, bw, SCORINGS["twopiece"])


@pytest.mark.parametrize("bw", [48, 80, 96, 176, 1024])
def test_generic_bandwidths(ctx, bw):
    """bandwidths whose W is not a power of two (or > 32) run the LDS-resident generic kernel"""
    rng = np.random.default_rng(99 + bw)
    pairs = _mk_pairs(rng, 48, [10, 100, 700, 1500])
    for mode in (S.MODE_GLOBAL, S.MODE_OVERLAP, S.MODE_EXTEND):
        _check(ctx, pairs, mode, bw, SCORINGS["affine"])
    _check(ctx, pairs, S.MODE_GLOBAL, bw, SCORINGS["twopiece"])
    _check(ctx, pairs, S.MODE_GLOBAL, bw, SCORINGS["linear"] if "linear" in SCORINGS else SCORINGS["affine"])
    # scores outside the exact-arithmetic guard, all three modes: the int8 saturation of the reference cell by cell
    for mode in (S.MODE_GLOBAL, S.MODE_OVERLAP, S.MODE_EXTEND):
        _check(ctx, pairs[:24], mode, bw, BIG_SCORINGS["big"])
    _check(ctx, pairs[:16], S.MODE_GLOBAL, bw, BIG_SCORINGS["big2piece"])


def test_bandwidth_zero_is_full_query(ctx):
    """bandwidth 0 = every pair gets its own band of roundup(qlen, 16) cells (bsalign.h:3861-3862)"""
    rng = np.random.default_rng(4242)
    pairs = _mk_pairs(rng, 64, [1, 15, 16, 17, 33, 100, 300, 777, 1200])
    for mode in (S.MODE_GLOBAL, S.MODE_OVERLAP, S.MODE_EXTEND):
        _check(ctx, pairs, mode, 0, SCORINGS["affine"])
    _check(ctx, pairs, S.MODE_GLOBAL, 0, SCORINGS["twopiece"])


@pytest.mark.parametrize("seg", ["64", "200"])
@pytest.mark.parametrize("scname", ["affine", "linear", "paper"])
def test_row_segments_of_the_persistent_forward_kernel(ctx, monkeypatch, seg, scname):
    """k_align8_fwd_xq: a pair's rows in segments that hand the band state on through memory (forced here: BSA_ALIGN8_XQ=1
    takes it for any batch, BSA_ALIGN8_XQ_SEG sets the rows per segment) -- lengths around the cuts, all three modes and widths"""
    monkeypatch.setenv("BSA_ALIGN8_XQ", "1")
    monkeypatch.setenv("BSA_ALIGN8_XQ_SEG", seg)
    rng = np.random.default_rng(4100 + int(seg) + len(scname))
    cut = int(seg)
    lens = [1, 7, 8, 9, cut - 1, cut, cut + 1, 2 * cut - 1, 2 * cut, 2 * cut + 8, 3 * cut + 5, 1000, 1777]
    pairs = _mk_pairs(rng, 120, lens)
    for bw in (128, 64, 256):
        for mode in (S.MODE_GLOBAL, S.MODE_OVERLAP, S.MODE_EXTEND):
            _check(ctx, pairs, mode, bw, SCORINGS[scname])
            fwd, _ = ctx.last_kernel_names()
            assert "k_align8_fwd_xq" in fwd, fwd


@pytest.mark.parametrize("xq", ["0", "1"])
def test_code_rows_with_two_bit_fields(ctx, monkeypatch, xq):
    """code format 1 (one-piece gaps with -gapo in 1 .. 3 at bandwidth 128): D and Od as one two-bit field per cell.  Gap openings 1, 2, 3,
    tie-rich scorings, divergent short pairs (the literal cell at query column 0, cells beyond the previous band end after jumps), whole-query
    bands in place -- and the planes format (BSA_ALIGN8_DO2=0) gives the same answers"""
    monkeypatch.setenv("BSA_ALIGN8_XQ", xq)
    monkeypatch.setenv("BSA_ALIGN8_XQ_SEG", "64")
    rng = np.random.default_rng(977)
    pairs = _mk_pairs(rng, 150, [1, 5, 17, 33, 64, 100, 129, 300, 1000, 2500], eps_list=(0.0, 0.05, 0.2, 0.4), ratios=(1.0, 1.0, 0.7, 1.4, 2.5))
    for _ in range(40):                                                # band jumps: lengths far apart
        Lt = int(rng.integers(20, 400))
        pairs.append((rng.integers(0, 4, size=max(int(Lt * float(rng.choice([3.0, 0.3]))), 1)).astype(np.uint8), rng.integers(0, 4, size=Lt).astype(np.uint8)))
    for sc in ((2, -6, -3, -2, 0, 0), (1, -1, -1, -1, 0, 0), (2, -3, -2, -1, 0, 0), (3, -4, -1, -2, 0, 0)):
        for mode in (S.MODE_GLOBAL, S.MODE_OVERLAP, S.MODE_EXTEND):
            _check(ctx, pairs, mode, 128, sc)
            fwd, trace = ctx.last_kernel_names()
            assert "two-bit" in fwd and trace == "k_align8_trace_codes_wave", (fwd, trace)
    short = [p for p in pairs if len(p[0]) <= 128]
    _check(ctx, short, S.MODE_GLOBAL, 0, (2, -6, -3, -2, 0, 0))         # whole-query bands, widened to 128 columns, band in place
    assert "k_align8_fwd_x_static" in ctx.last_kernel_names()[0] and "two-bit" in ctx.last_kernel_names()[0]
    monkeypatch.setenv("BSA_ALIGN8_DO2", "0")
    _check(ctx, pairs, S.MODE_GLOBAL, 128, (2, -6, -3, -2, 0, 0))
    assert "two-bit" not in ctx.last_kernel_names()[0]


@pytest.mark.parametrize("bw", [64, 256])
def test_two_piece_gaps_at_bandwidth_64_and_256_take_the_compact_path(ctx, bw, monkeypatch):
    """the two-piece code rows generalised from eight cells a reference block to four (one dword a block) and sixteen (four dwords):
    k_align8_fwd_x2 + k_align8_trace_codes2 instead of row records, all three modes, also in row segments"""
    rng = np.random.default_rng(640 + bw)
    pairs = _mk_pairs(rng, 120, [1, 9, 17, 63, 64, 65, 130, 300, 1000, 2200], eps_list=(0.0, 0.05, 0.15, 0.3), ratios=(1.0, 1.0, 0.8, 1.3))
    for _ in range(30):
        Lt = int(rng.integers(20, 400))
        pairs.append((rng.integers(0, 4, size=max(int(Lt * float(rng.choice([3.0, 0.4]))), 1)).astype(np.uint8), rng.integers(0, 4, size=Lt).astype(np.uint8)))
    for sc in (SCORINGS["twopiece"], (1, -3, -2, -2, -6, -1)):
        for mode in (S.MODE_GLOBAL, S.MODE_OVERLAP, S.MODE_EXTEND):
            _check(ctx, pairs, mode, bw, sc)
            fwd, trace = ctx.last_kernel_names()
            assert "k_align8_fwd_x2" in fwd and trace == "k_align8_trace_codes2", (fwd, trace)
    if bw == 64:
        monkeypatch.setenv("BSA_ALIGN8_XQ", "1")
        monkeypatch.setenv("BSA_ALIGN8_XQ_SEG", "64")
        _check(ctx, pairs, S.MODE_GLOBAL, bw, SCORINGS["twopiece"])
        # bandwidth 128 in row segments: the four-lane shape (sixteen cells a half, two reference blocks), and the eight-lane one
        for mode in (S.MODE_GLOBAL, S.MODE_OVERLAP, S.MODE_EXTEND):
            _check(ctx, pairs, mode, 128, SCORINGS["twopiece"])
            assert "four lanes" in ctx.last_kernel_names()[0]
        monkeypatch.setenv("BSA_ALIGN8_X2_LANES", "8")
        _check(ctx, pairs, S.MODE_EXTEND, 128, SCORINGS["twopiece"])
        assert "k_align8_fwd_xq" in ctx.last_kernel_names()[0] and "four lanes" not in ctx.last_kernel_names()[0]


def test_synthetic_10k_bw128(ctx):
    """the benchmark shape (C2): 10 kbp synthetic pairs, global, bw 128"""
    pairs = [S.synth_pair(k, 10000) for k in range(24)]
    _check(ctx, pairs, S.MODE_GLOBAL, 128, SCORINGS["affine"])
    _check(ctx, pairs[:8], S.MODE_GLOBAL, 128, SCORINGS["paper"])


def test_length_mismatch_and_jumps(ctx):
    """tlen << qlen forces band jumps >= W cells (generic movx path); qlen << tlen the opposite"""
    rng = np.random.default_rng(5)
    pairs = []
    for _ in range(40):
        Lt = int(rng.integers(20, 400))
        Lq = int(Lt * float(rng.choice([2.0, 3.0, 5.0, 0.5, 0.3])))
        pairs.append((rng.integers(0, 4, size=max(Lq, 1)).astype(np.uint8), rng.integers(0, 4, size=Lt).astype(np.uint8)))
    for bw in (16, 32, 64):
        for mode in (S.MODE_GLOBAL, S.MODE_OVERLAP, S.MODE_EXTEND):
            _check(ctx, pairs, mode, bw, SCORINGS["affine"])


def test_bad_and_empty_inputs(ctx):
    import bsalign_amd as B
    q = np.array([0, 1, 2, 3] * 10, dtype=np.uint8)
    bad = q.copy()
    bad[5] = 4
    pairs = [(q, q), (bad, q), (np.zeros(0, dtype=np.uint8), q), (q, np.zeros(0, dtype=np.uint8)), (q, q[::-1].copy())]
    out, cigs, status = ctx.align_batch(pairs, B.make_params(S.MODE_GLOBAL, 64))
    assert status[0] == 0 and status[4] == 0
    assert status[1] & B.ST_BAD_BASE
    assert status[2] & B.ST_EMPTY and status[3] & B.ST_EMPTY
    assert out[1]["aln"] == 0 and out[2]["aln"] == 0 and len(cigs[1]) == 0
    res, cig, _ = S.oracle_align(q, q, S.MODE_GLOBAL, 64, 2, -6, -3, -2, 0, 0)
    assert np.array_equal(np.array([out[0][f] for f in out.dtype.names], dtype=np.int32), res)
    assert np.array_equal(cigs[0], cig)


def test_golden_align8_cases(ctx):
    """the committed golden vectors (generated from the real reference) through the HIP path"""
    import os
    import bsalign_amd as B
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "align8.npz"))
    groups = {}
    for k in range(int(g["n"][0])):
        meta = tuple(int(x) for x in g["meta_%d" % k])
        groups.setdefault(meta, []).append(k)
    assert len(groups) > 30
    for meta, ks in groups.items():
        pairs = [(g["q_%d" % k], g["t_%d" % k]) for k in ks]
        out, cigs, status = ctx.align_batch(pairs, B.make_params(meta[0], meta[1], *meta[2:]))
        for i, k in enumerate(ks):
            got = np.array([out[i][f] for f in out.dtype.names], dtype=np.int32)
            assert status[i] == 0 and np.array_equal(got, g["res_%d" % k]) and np.array_equal(cigs[i], g["cig_%d" % k]), (meta, k)


def test_literal_row_record_path_still_matches(ctx, monkeypatch):
    """global alignments normally take the compact 4-bit-code path; BSA_ALIGN8_LITERAL=1 keeps the row-record kernels
    (the ones every other mode uses) on the same inputs"""
    monkeypatch.setenv("BSA_ALIGN8_LITERAL", "1")
    rng = np.random.default_rng(31337)
    pairs = _mk_pairs(rng, 64, [1, 17, 100, 300, 1000, 2000])
    for bw in (64, 128):
        _check(ctx, pairs, S.MODE_GLOBAL, bw, SCORINGS["affine"])
    _check(ctx, [S.synth_pair(k, 10000) for k in range(8)], S.MODE_GLOBAL, 128, SCORINGS["affine"])


@pytest.mark.parametrize("wave", ["1", "0"])
def test_compact_path_band_and_ratio_corners(ctx, monkeypatch, wave):
    """length mismatches (rush-to-end steering, band jumps), tiny inputs and high divergence through the compact path, with
    the one-walk-per-wave traceback (bandwidth 128) and with the pair-per-lane LDS-ring kernel"""
    monkeypatch.setenv("BSA_ALIGN8_TRACE_WAVE", wave)
    rng = np.random.default_rng(99)
    pairs = []
    for _ in range(120):
        Lt = int(rng.choice([1, 2, 15, 16, 17, 33, 80, 200, 600]))
        Lq = max(1, int(Lt * float(rng.choice([1.0, 0.5, 0.3, 2.0, 3.0, 1.1]))))
        T = rng.integers(0, 4, size=Lt).astype(np.uint8)
        Q = S.mutate(rng, T, float(rng.choice([0.0, 0.1, 0.4])))
        Q = Q[:Lq] if Lq <= len(Q) else np.concatenate([Q, rng.integers(0, 4, size=Lq - len(Q)).astype(np.uint8)])
        if len(Q) == 0:
            Q = np.array([2], np.uint8)
        pairs.append((Q, T))
    for bw in (64, 128, 256):
        for sc in ("affine", "paper", "linear"):
            _check(ctx, pairs, S.MODE_GLOBAL, bw, SCORINGS[sc])


@pytest.mark.parametrize("wave", ["1", "0"])
@pytest.mark.parametrize("mode", [S.MODE_GLOBAL, S.MODE_OVERLAP, S.MODE_EXTEND])
def test_two_piece_gaps_take_the_compact_path(ctx, monkeypatch, wave, mode):
    """2-piece gaps (POA default and others), global, bandwidth 128: 8-bit traceback codes written by k_align8_fwd_x2 and walked by
    k_align8_trace_codes2_wave (one walk per wave) or k_align8_trace_codes2 (one per lane) -- results identical to the oracle's
    literal backcal, corners included, and the plan really is compact"""
    import bsalign_amd as B
    monkeypatch.setenv("BSA_ALIGN8_TRACE_WAVE", wave)
    rng = np.random.default_rng(2025)
    pairs = _mk_pairs(rng, 160, [1, 2, 15, 16, 17, 63, 64, 65, 100, 300, 1000, 2500])
    for _ in range(60):                                      # length mismatches: steering rushes, band jumps
        Lt = int(rng.choice([5, 33, 80, 200, 600]))
        Lq = max(1, int(Lt * float(rng.choice([0.5, 0.3, 2.0, 3.0, 1.1]))))
        T = rng.integers(0, 4, size=Lt).astype(np.uint8)
        Q = S.mutate(rng, T, float(rng.choice([0.0, 0.1, 0.4])))
        Q = Q[:Lq] if Lq <= len(Q) else np.concatenate([Q, rng.integers(0, 4, size=Lq - len(Q)).astype(np.uint8)])
        pairs.append((Q if len(Q) else np.array([2], np.uint8), T))
    if mode != S.MODE_GLOBAL:                                # overlap-like inputs: the query is a suffix / prefix of what the target holds
        for _ in range(60):
            Lt = int(rng.choice([120, 400, 1200]))
            T = rng.integers(0, 4, size=Lt).astype(np.uint8)
            Q = S.mutate(rng, T, float(rng.choice([0.02, 0.1, 0.2])))
            cut = int(len(Q) * float(rng.choice([0.3, 0.5])))
            pairs.append((Q[cut:] if rng.random() < 0.5 else Q[:max(1, len(Q) - cut)], T))
    for sc in ((2, -6, -3, -2, -8, -1), (2, -4, -4, -2, -12, -1), (3, -5, -2, -3, -9, -1), (1, -3, -2, -2, -6, -1)):
        _check(ctx, pairs, mode, 128, sc)
        assert "k_align8_fwd_x2" in ctx.last_kernel_names()[0]
        assert ctx.last_kernel_names()[1] == ("k_align8_trace_codes2_wave" if wave == "1" else "k_align8_trace_codes2")


NONTERMINATING_2PIECE = (   # found by tools/stress_align8.py (seed 9202, batch 46): with two-piece gaps the reference's D test fires at query column 0
    # where no deletion ends and its run-length scan walks off the matrix -- the literal oracle reports ORC_ERR_TRACE
    [2, 1, 2, 2, 3, 0, 0, 2, 1, 2, 3, 3, 3, 1, 0, 0, 3, 0, 2, 1, 0, 2, 0, 3, 0, 2, 3, 0, 3, 2, 1, 1, 0, 2, 0, 0, 2, 0, 0, 2, 1, 2, 3, 0, 0, 0, 1, 2, 2, 1, 1, 3, 2, 0, 2, 0, 0,
     2, 0, 1, 2, 2, 2, 2, 3, 0, 1, 0, 2, 2],
    [1, 2, 1, 0, 0, 3, 0, 1, 1, 2, 3, 2, 3, 0, 0, 1, 0, 2, 0, 2, 1, 2, 3, 1, 0, 2, 2, 3, 0, 3, 2, 0, 1, 0, 0, 2, 0, 0, 2, 0, 0, 2, 1, 2, 3, 0, 2, 0, 0, 1, 1, 2, 2, 1, 3, 2, 1,
     3, 3, 2, 3, 0, 2, 0])


@pytest.mark.parametrize("wave", ["1", "0"])
def test_two_piece_pair_on_which_the_reference_does_not_terminate_is_flagged(ctx, monkeypatch, wave):
    """the code tracebacks returned a path with status 0 here.  The deletion the flags open at query column 0 is now
    handed to the literal traceback, which -- like the reference -- finds no run length: BSA_ST_TRACE, no invented answer."""
    import bsalign_amd as B
    monkeypatch.setenv("BSA_ALIGN8_TRACE_WAVE", wave)
    q, t = (np.array(x, np.uint8) for x in NONTERMINATING_2PIECE)
    sc = (2, -6, -3, -2, -8, -1)
    for bw in (0, 128):
        assert S.oracle_align(q, t, S.MODE_GLOBAL, bw, *sc)[2] == S.ORC_ERR_TRACE
        out, cigs, status = ctx.align_batch([(q, t)] * 3, B.make_params(S.MODE_GLOBAL, bw, *sc))
        assert all(int(s) & B.ST_TRACE for s in status) and all(len(c) == 0 for c in cigs)


def test_handover_to_literal_path_merges_results(ctx, monkeypatch):
    """bsa_align_batch re-runs pairs the compact traceback flags through the row-record kernels and splices their
    results and CIGARs back; the debug hook declares every 3rd pair undecided so that the merge is exercised"""
    monkeypatch.setenv("BSA_DEBUG_HANDOVER", "3")
    rng = np.random.default_rng(4711)
    pairs = _mk_pairs(rng, 50, [1, 17, 100, 300, 1000])
    _check(ctx, pairs, S.MODE_GLOBAL, 128, SCORINGS["affine"])
    _check(ctx, pairs, S.MODE_GLOBAL, 64, SCORINGS["paper"])


@pytest.mark.parametrize("literal", [False, True])
def test_chunked_batches_and_cigar_properties(monkeypatch, literal):
    """a workspace limit forces the batch through several chunks (and, with BSA_PIPELINE, two streams); results must not
    depend on the chunking, and every CIGAR must consume exactly [qb,qe) x [tb,te) with aln = mat+mis+ins+del"""
    import bsalign_amd as B
    if literal:
        monkeypatch.setenv("BSA_ALIGN8_LITERAL", "1")
    pairs = [S.synth_pair(k, 1500) for k in range(300)]
    par = B.make_params(S.MODE_GLOBAL, 128, *SCORINGS["affine"])
    big = B.Context(0)
    out0, cig0, st0 = big.align_batch(pairs, par)
    big.close()
    small = B.Context(0, workspace_limit=(24 << 20) if not literal else (80 << 20))
    out1, cig1, st1 = small.align_batch(pairs, par)
    monkeypatch.setenv("BSA_PIPELINE", "1")
    out2, cig2, st2 = small.align_batch(pairs, par)
    small.close()
    assert (st0 == 0).all() and np.array_equal(st0, st1) and np.array_equal(st0, st2)
    assert np.array_equal(out0, out1) and np.array_equal(out0, out2)
    for k, (q, t) in enumerate(pairs):
        assert np.array_equal(cig0[k], cig1[k]) and np.array_equal(cig0[k], cig2[k]), k
        qs, ts = S.cigar_spans(cig0[k])
        r = out0[k]
        assert (r["qb"], r["qe"], r["tb"], r["te"]) == (0, len(q), 0, len(t))
        assert qs == r["qe"] - r["qb"] and ts == r["te"] - r["tb"]
        assert r["aln"] == r["mat"] + r["mis"] + r["ins"] + r["del"]
    res, cig, _ = S.oracle_align(pairs[7][0], pairs[7][1], S.MODE_GLOBAL, 128, *SCORINGS["affine"])
    assert np.array_equal(np.array([out0[7][f] for f in out0.dtype.names], dtype=np.int32), res) and np.array_equal(cig0[7], cig)


def test_sharded_flow_over_rccl_single_rank():
    """examples/align_sharded.py under torch.distributed.run with the nccl (= RCCL) backend: scatter on device tensors,
    align through device pointers, gather -- with the one GPU this box has (the N > 1 message pattern is covered by the
    gloo test tests/test_shard_cpu.py)"""
    import os
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(S.ROOT, "examples", "align_sharded.py"), "--pairs", "300", "--length", "2000"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-800:] + r.stderr[-800:]
    assert "identical to the oracle: True" in r.stdout


@pytest.mark.parametrize("wave", ["1", "0"])
def test_ties_at_column_zero(ctx, monkeypatch, wave):
    """scorings full of ties (match == -mismatch == -gap) on short, very divergent pairs: cells of query column 0 where both the
    M and the D flag are set keep prior_match (`... && qb`, bsalign.h:3761-3764), so M wins there although the column is the first
    of the previous row's band -- a randomised campaign found the one-walk-per-wave kernel taking D; all three modes, 1- and 2-piece"""
    monkeypatch.setenv("BSA_ALIGN8_TRACE_WAVE", wave)
    rng = np.random.default_rng(5)
    pairs = []
    for _ in range(600):
        L = int(rng.choice([4, 10, 14, 17, 20, 33, 70]))
        T = rng.integers(0, 4, size=L).astype(np.uint8)
        Q = S.mutate(rng, T, 0.4)
        pairs.append((Q if len(Q) else np.array([1], np.uint8), T))
    for sc in ((1, -1, -1, -1, 0, 0), (2, -2, -4, -2, 0, 0), (3, -4, -6, -1, 0, 0)):
        for mode in (S.MODE_GLOBAL, S.MODE_OVERLAP, S.MODE_EXTEND):
            _check(ctx, pairs, mode, 128, sc)
    _check(ctx, pairs, S.MODE_GLOBAL, 128, (1, -3, -2, -2, -6, -1))


@pytest.mark.parametrize("bw", [0, 48, 80, 112, 176, 208])
def test_whole_query_bands_run_widened_on_the_compact_path(ctx, bw, monkeypatch):
    """a band that covers every query (the reference CLI's `-W 0`, or a bandwidth no shorter than any query, of a width the
    register kernels do not have): the band never moves and the DP does not depend on its width inside the exact-arithmetic
    guard, so the batch runs at the next register-kernel width on the compact path -- same results as the reference's own
    width (the oracle runs the requested one), and as the run-time-width kernel (BSA_ALIGN8_WIDEN=0); in overlap / extend mode
    the maximum of the last row is taken in the reference's striping of its own band"""
    import bsalign_amd as B
    rng = np.random.default_rng(4242 + bw)
    top = bw if bw else 256
    lens = [l for l in (1, 2, 15, 16, 17, 31, 33, 47, 48, 63, 64, 65, 79, 100, 111, 112, 127, 129, 150, 176, 200, 208, 240, 255, 256) if l <= top]
    pairs = [(q[:top] if len(q) > top else q, t) for q, t in _mk_pairs(rng, 160, lens, eps_list=(0.0, 0.05, 0.2, 0.4), ratios=(1.0, 1.0, 0.5, 0.9))]
    for scname in ("affine", "paper", "linear"):
        for mode in (S.MODE_GLOBAL, S.MODE_OVERLAP, S.MODE_EXTEND):        # (overlap / extend: row_max in the reference's striping)
            _check(ctx, pairs, mode, bw, SCORINGS[scname])
            assert "k_align8_fwd_x" in ctx.last_kernel_names()[0], ctx.last_kernel_names()
    out_w, cig_w, st_w = ctx.align_batch(pairs, B.make_params(S.MODE_GLOBAL, bw, *SCORINGS["affine"]))
    monkeypatch.setenv("BSA_ALIGN8_WIDEN", "0")
    out_g, cig_g, st_g = ctx.align_batch(pairs, B.make_params(S.MODE_GLOBAL, bw, *SCORINGS["affine"]))
    assert "gen" in ctx.last_kernel_names()[0]
    assert np.array_equal(out_w, out_g) and np.array_equal(st_w, st_g) and all(np.array_equal(a, b) for a, b in zip(cig_w, cig_g))
    monkeypatch.delenv("BSA_ALIGN8_WIDEN")
    for mode in (S.MODE_OVERLAP, S.MODE_EXTEND):
        out_w, cig_w, st_w = ctx.align_batch(pairs, B.make_params(mode, bw, *SCORINGS["paper"]))
        monkeypatch.setenv("BSA_ALIGN8_WIDEN", "0")
        out_g, cig_g, st_g = ctx.align_batch(pairs, B.make_params(mode, bw, *SCORINGS["paper"]))
        monkeypatch.delenv("BSA_ALIGN8_WIDEN")
        assert np.array_equal(out_w, out_g) and np.array_equal(st_w, st_g) and all(np.array_equal(a, b) for a, b in zip(cig_w, cig_g))
    # two-piece gaps: the compact path exists at bandwidth 128, so bands of up to 128 columns go there
    if bw and bw <= 128:
        _check(ctx, pairs, S.MODE_GLOBAL, bw, SCORINGS["twopiece"])
        assert "k_align8_fwd_x2" in ctx.last_kernel_names()[0], ctx.last_kernel_names()
    short = [(q[:128], t) for q, t in pairs]
    _check(ctx, short, S.MODE_GLOBAL, 0, SCORINGS["twopiece"])
    assert "k_align8_fwd_x2" in ctx.last_kernel_names()[0], ctx.last_kernel_names()
    _check(ctx, pairs, S.MODE_OVERLAP, bw, SCORINGS["twopiece"])          # (round 3: overlap / extend take the two-piece compact path as well)
    if bw and bw <= 128:
        assert "k_align8_fwd_x2" in ctx.last_kernel_names()[0], ctx.last_kernel_names()
    # a scoring outside the guard keeps the run-time-width kernel
    _check(ctx, pairs[:32], S.MODE_GLOBAL, bw, (10, -30, -20, -10, 0, 0))
    assert "k_align8_fwd_x" not in ctx.last_kernel_names()[0]


@pytest.mark.parametrize("bw", [0, 1008, 3008])
def test_whole_query_bands_above_256_columns_run_the_systolic_wavefront(ctx, bw, monkeypatch):
    """global mode, a band that covers every query and is wider than the register kernels (the reference CLI's `-W 0` on long
    reads): one wave per pair, a lane per target row, bit planes of traceback codes (bsa_align8_sys.hip) -- results equal the
    reference's own band (the oracle runs the requested width) and the run-time-width kernel's (BSA_ALIGN8_SYS=0)"""
    import bsalign_amd as B
    rng = np.random.default_rng(9000 + bw)
    top = bw if bw else 3000
    lens = [l for l in (257, 258, 271, 272, 273, 300, 319, 320, 321, 500, 511, 512, 513, 700, 1000, 1008, 1500, 2047, 2048, 2049, 3000) if l <= top]
    pairs = [(q[:top] if len(q) > top else q, t) for q, t in _mk_pairs(rng, 72, lens, eps_list=(0.0, 0.05, 0.2, 0.4), ratios=(1.0, 1.0, 0.5, 0.9, 1.1))]
    pairs = [(q, t) for q, t in pairs if len(q) > 256 or bw]
    pairs.append((pairs[0][0], pairs[0][1][:1]))            # a one-row target
    pairs.append((pairs[1][0], pairs[1][1][:63]))
    pairs.append((pairs[2][0], pairs[2][1][:64]))
    pairs.append((pairs[3][0], pairs[3][1][:65]))
    for scname in ("affine", "paper", "linear"):
        for mode in (S.MODE_GLOBAL, S.MODE_OVERLAP, S.MODE_EXTEND):      # (`bsalign align` defaults to overlap: the end cell is searched for)
            _check(ctx, pairs, mode, bw, SCORINGS[scname])
            names = ctx.last_kernel_names()
            assert "k_align8_fwd_sys" in names[0] and "k_align8_trace_sys" in names[1], names
    par = B.make_params(S.MODE_GLOBAL, bw, *SCORINGS["affine"])
    out_s, cig_s, st_s = ctx.align_batch(pairs, par)
    monkeypatch.setenv("BSA_ALIGN8_SYS", "0")
    out_g, cig_g, st_g = ctx.align_batch(pairs, par)
    assert "gen" in ctx.last_kernel_names()[0]
    monkeypatch.delenv("BSA_ALIGN8_SYS")
    assert np.array_equal(out_s, out_g) and np.array_equal(st_s, st_g) and all(np.array_equal(a, b) for a, b in zip(cig_s, cig_g))
    # the other shapes of the same kernels: one wave per pair in the forward pass, a pair per lane in the traceback
    for knob, val in (("BSA_ALIGN8_SYS_WAVES", "1"), ("BSA_ALIGN8_SYS_WAVES", "4"), ("BSA_ALIGN8_SYS_TRACE", "lane")):
        monkeypatch.setenv(knob, val)
        for mode in (S.MODE_GLOBAL, S.MODE_OVERLAP):
            out_v, cig_v, st_v = ctx.align_batch(pairs, B.make_params(mode, bw, *SCORINGS["affine"]))
            out_r, cig_r, st_r = (out_s, cig_s, st_s) if mode == S.MODE_GLOBAL else (None, None, None)
            if out_r is None:
                monkeypatch.delenv(knob)
                out_r, cig_r, st_r = ctx.align_batch(pairs, B.make_params(mode, bw, *SCORINGS["affine"]))
                monkeypatch.setenv(knob, val)
            assert np.array_equal(out_v, out_r) and np.array_equal(st_v, st_r) and all(np.array_equal(a, b) for a, b in zip(cig_v, cig_r)), (knob, val, mode)
        monkeypatch.delenv(knob)
    # two-piece gaps: eight facts per cell, the second gap chain and the second deletion state through the same kernels
    for mode in (S.MODE_GLOBAL, S.MODE_OVERLAP, S.MODE_EXTEND):
        _check(ctx, pairs, mode, bw, SCORINGS["twopiece"])
        assert "k_align8_fwd_sys" in ctx.last_kernel_names()[0]
    _check(ctx, pairs, S.MODE_GLOBAL, bw, (1, -3, -2, -2, -6, -1))
    monkeypatch.setenv("BSA_ALIGN8_SYS_TRACE", "lane")
    _check(ctx, pairs, S.MODE_OVERLAP, bw, SCORINGS["twopiece"])
    monkeypatch.delenv("BSA_ALIGN8_SYS_TRACE")
    # scorings the checked kernel cannot take either (row -1's first difference gapo + gape + min - max does not fit int8) keep the run-time-width kernel
    _check(ctx, pairs[:16], S.MODE_GLOBAL, bw, (50, -50, -20, -10, 0, 0))
    assert "sys" not in ctx.last_kernel_names()[0]


@pytest.mark.parametrize("bw", [0, 1008])
def test_whole_query_bands_with_scores_outside_the_guard_run_the_checked_systolic_kernel(ctx, bw, monkeypatch):
    """whole-query bands above 256 columns with scores outside the static guard (they ran the run-time-width
    kernel at 28 GCUPS): the systolic kernel in its CHECKED form -- every pair's cells are tested against the int8 range the
    reference computes in (and the -63 restart of its running blocks), a pair that fails is flagged and re-run by the literal
    kernels.  Results equal the lane-exact oracle's in all three modes and gap models; on ordinary pairs nothing is handed over;
    the unchecked run-time-width kernel (BSA_ALIGN8_SYS_CHK=0) gives the same."""
    import bsalign_amd as B
    rng = np.random.default_rng(9100 + bw)
    top = bw if bw else 3000
    lens = [l for l in (257, 272, 273, 300, 320, 500, 511, 512, 513, 700, 1000, 1008, 1500, 2047, 2048, 2049, 3000) if l <= top]
    pairs = [(q[:top] if len(q) > top else q, t) for q, t in _mk_pairs(rng, 60, lens, eps_list=(0.0, 0.05, 0.2, 0.4), ratios=(1.0, 1.0, 0.5, 0.9, 1.1))]
    pairs = [(q, t) for q, t in pairs if len(q) > 256 or bw]
    pairs.append((pairs[0][0], pairs[0][1][:1]))            # a one-row target
    pairs.append((pairs[1][0], pairs[1][1][:63]))
    pairs.append((pairs[2][0], pairs[2][1][:65]))
    for scname, sc in BIG_SCORINGS.items():
        for mode in (S.MODE_GLOBAL, S.MODE_OVERLAP, S.MODE_EXTEND):
            _check(ctx, pairs, mode, bw, sc)
            names = ctx.last_kernel_names()
            assert "k_align8_fwd_sys<CHK>" in names[0] and "k_align8_trace_sys" in names[1], names
            if scname != "big2piece":
                assert ctx.last_handover() <= 2, (scname, mode, ctx.last_handover())       # (a deletion run that reaches row -1 is the literal path's, HISTORY section 5)
    par = B.make_params(S.MODE_OVERLAP, bw, *BIG_SCORINGS["big"])
    out_s, cig_s, st_s = ctx.align_batch(pairs, par)
    monkeypatch.setenv("BSA_ALIGN8_SYS_CHK", "0")
    out_g, cig_g, st_g = ctx.align_batch(pairs, par)
    assert "gen" in ctx.last_kernel_names()[0]
    monkeypatch.delenv("BSA_ALIGN8_SYS_CHK")
    assert np.array_equal(out_s, out_g) and np.array_equal(st_s, st_g) and all(np.array_equal(a, b) for a, b in zip(cig_s, cig_g))
    # the checked kernel inside the guard (forced): same results as the unchecked one, nothing flagged
    monkeypatch.setenv("BSA_ALIGN8_SYS_CHK", "1")
    for mode in (S.MODE_GLOBAL, S.MODE_EXTEND):
        _check(ctx, pairs, mode, bw, SCORINGS["affine"])
        assert "k_align8_fwd_sys<CHK>" in ctx.last_kernel_names()[0] and ctx.last_handover() <= 2


def test_checked_systolic_kernel_flags_what_the_int8_arithmetic_clamps(ctx):
    """gap costs so large that the F entering a running block can lie below the block's own restart value of -63 (gapo + gape
    below -31: the striping of the reference's band then shows in the result): the checked kernel must flag such pairs (they are
    re-run by the lane-exact kernels) -- results equal the oracle's either way, and pairs are indeed handed over"""
    import bsalign_amd as B
    rng = np.random.default_rng(9200)
    pairs = _mk_pairs(rng, 40, [300, 500, 700, 1000], eps_list=(0.02, 0.1, 0.3), ratios=(1.0, 0.9, 1.1))
    handed = 0
    # (37, -59, -13, -1): the campaign's find of round 6 -- a mismatch at (0, 0) seeds row 0 with (min - max) + min = -155, which the reference inserts as
    # a BYTE (bsalign.h:2910: it wraps to 101); the kernel wraps it too and the differences that then leave int8 flag the pair
    for sc in ((40, -40, -20, -25, 0, 0), (20, -40, -25, -15, 0, 0), (5, -60, -3, -60, 0, 0), (37, -59, -13, -1, 0, 0)):
        for mode in (S.MODE_GLOBAL, S.MODE_OVERLAP, S.MODE_EXTEND):
            _check(ctx, pairs, mode, 0, sc)
            assert "k_align8_fwd_sys<CHK>" in ctx.last_kernel_names()[0], ctx.last_kernel_names()
            handed += ctx.last_handover()
    assert handed > 0


@pytest.mark.parametrize("mode", [S.MODE_GLOBAL, S.MODE_OVERLAP])
def test_host_pointer_batch_in_two_slices(ctx, mode, monkeypatch):
    """bsa_align_batch on host buffers cuts a large batch in two so that the second half's sequences travel while the first half's kernels run and
    the first half's results go back while the second half's run (forced here: BSA_BATCH_SLICES=2): results, status words, CIGAR offsets and
    words equal the unsliced call's (BSA_BATCH_SLICES=1) and the oracle's.  (bench.py's full-size check `host_pointer_call_identical` runs the
    sliced path on its own blob layout: all targets, then all queries -- two byte intervals a slice.)"""
    import bsalign_amd as B
    rng = np.random.default_rng(5150 + mode)
    pairs = _mk_pairs(rng, 300, [40, 300, 900, 1500], eps_list=(0.02, 0.1, 0.25))
    par = B.make_params(mode, 128, *SCORINGS["affine"])
    monkeypatch.setenv("BSA_BATCH_SLICES", "1")
    out1, cig1, st1 = ctx.align_batch(pairs, par)
    monkeypatch.setenv("BSA_BATCH_SLICES", "2")
    out2, cig2, st2 = ctx.align_batch(pairs, par)
    assert np.array_equal(out1, out2) and np.array_equal(st1, st2) and all(np.array_equal(a, b) for a, b in zip(cig1, cig2))
    _check(ctx, pairs, mode, 128, SCORINGS["affine"])
    # two-piece gaps (8-bit codes, CIGAR arenas of another size), an odd number of pairs
    shared = pairs[:150] + [(pairs[0][0], pairs[151][1])] + pairs[151:]
    _check(ctx, shared, mode, 128, SCORINGS["twopiece"])
    monkeypatch.setenv("BSA_BATCH_SLICES", "1")
    _check(ctx, shared, mode, 128, SCORINGS["twopiece"])


def test_whole_query_plan_with_mixed_lengths_on_device_pointers(ctx):
    """the two-phase API does not sort pairs into width classes (the caller groups them): a plan over queries of 1 .. 3000 bases at
    bandwidth 0 runs the systolic kernels for all of them, four waves per pair included -- same results as the oracle, pair by pair"""
    import torch
    import bsalign_amd as B
    rng = np.random.default_rng(77)
    pairs = _mk_pairs(rng, 96, [1, 2, 15, 63, 64, 65, 130, 300, 1000, 1100, 2500, 3000], eps_list=(0.0, 0.1, 0.3), ratios=(1.0, 0.5, 1.2))
    seqs, qoff, qlen, toff, tlen = B.pack_pairs(pairs)
    n = len(pairs)
    for mode, sc in ((S.MODE_GLOBAL, SCORINGS["affine"]), (S.MODE_OVERLAP, SCORINGS["paper"]), (S.MODE_EXTEND, SCORINGS["twopiece"])):
        par = B.make_params(mode, 0, *sc)
        plan = B.AlignPlan(ctx, qoff, qlen, toff, tlen, par)
        d_seqs = torch.from_numpy(seqs).cuda()
        d_out = torch.zeros(n * 10, dtype=torch.int32, device="cuda")
        d_cig = torch.empty(int(sum(len(q) + len(t) + 8 for q, t in pairs)), dtype=torch.int32, device="cuda")
        d_off = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
        d_st = torch.zeros(n, dtype=torch.int32, device="cuda")
        plan.run(d_seqs, d_out, d_cig, d_off, d_st)
        torch.cuda.synchronize()
        assert "k_align8_fwd_sys" in ctx.last_kernel_names()[0]
        out = d_out.cpu().numpy().reshape(n, 10); off = d_off.cpu().numpy(); st = d_st.cpu().numpy(); cig = d_cig.cpu().numpy().view(np.uint32)
        handed = []
        for k, (q, t) in enumerate(pairs):
            res, ocig, m = S.oracle_align(q, t, mode, 0, *sc)
            if m == S.ORC_ERR_TRACE:              # the reference's own traceback does not terminate here: the device must say so
                assert st[k] & B.ST_TRACE, (mode, k, len(q), len(t))
                continue
            if st[k] & B.ST_TRACE:                # the code traceback declined (a deletion run reaching row -1 / decided at column 0, HISTORY section 5)
                handed.append(k)
                continue
            assert st[k] == 0 and np.array_equal(out[k], res) and np.array_equal(cig[int(off[k]):int(off[k + 1])], ocig), (mode, k, len(q), len(t))
        # a flag the oracle does not raise is only legitimate as a hand-over: the host entry gives exactly these pairs to the literal
        # kernels and must come back with the oracle's alignment for every one of them
        assert len(handed) <= 4, (mode, handed)
        if handed:
            _check(ctx, [pairs[k] for k in handed], mode, 0, sc)
        plan.close()


def test_whole_query_bands_10k(ctx):
    """`bsalign align -W 0` on the benchmark's 10 kbp pairs (bench.py --length 10000 --bw -1)"""
    pairs = [S.synth_pair(k, 10000) for k in range(6)]
    _check(ctx, pairs, S.MODE_GLOBAL, 0, SCORINGS["affine"])
    assert "k_align8_fwd_sys" in ctx.last_kernel_names()[0]
    _check(ctx, pairs[:3], S.MODE_GLOBAL, 0, SCORINGS["linear"])
    _check(ctx, pairs[:4], S.MODE_OVERLAP, 0, SCORINGS["paper"])          # example/run.sh "NoBand": align -M 2 -X 2 -O 4 -E 2, overlap
    assert "k_align8_fwd_sys" in ctx.last_kernel_names()[0]
    _check(ctx, pairs[:2], S.MODE_EXTEND, 0, SCORINGS["affine"])
    _check(ctx, pairs[:3], S.MODE_OVERLAP, 0, SCORINGS["twopiece"])
    assert "k_align8_fwd_sys" in ctx.last_kernel_names()[0]


@pytest.mark.parametrize("bw", [0, 112])
def test_mixed_whole_query_batch_runs_one_sub_batch_per_width_class(ctx, bw):
    """bandwidth 0 (or an odd one) over queries of very different lengths: the host-pointer entry sends every width class of the
    widened dispatch down as a sub-batch of its own and the rest to the run-time-width kernel; results, status words and the
    CIGAR arena come back in the caller's order"""
    rng = np.random.default_rng(515 + bw)
    pairs = _mk_pairs(rng, 240, [1, 5, 30, 60, 64, 65, 100, 128, 129, 200, 256, 257, 300, 700, 1200], eps_list=(0.0, 0.05, 0.2), ratios=(1.0, 1.0, 0.7))
    order = rng.permutation(len(pairs))
    pairs = [pairs[i] for i in order]
    for mode in (S.MODE_GLOBAL, S.MODE_OVERLAP, S.MODE_EXTEND):
        _check(ctx, pairs, mode, bw, SCORINGS["affine"])
    _check(ctx, pairs, S.MODE_GLOBAL, bw, SCORINGS["twopiece"])
    # the arena too small: the required number of words is reported (bsalign_hip.h)
    import bsalign_amd as B
    import ctypes as C
    seqs, qoff, qlen, toff, tlen = B.pack_pairs(pairs)
    n = len(pairs)
    out = np.zeros(n, dtype=B.RESULT_DTYPE)
    cig = np.zeros(8, dtype=np.uint32)
    coff = np.zeros(n + 1, dtype=np.uint64)
    st = np.zeros(n, dtype=np.uint32)
    par = B.make_params(S.MODE_GLOBAL, bw, *SCORINGS["affine"])
    rc = B.lib().bsa_align_batch(ctx.h, seqs.ctypes.data, seqs.size, qoff.ctypes.data, qlen.ctypes.data, toff.ctypes.data, tlen.ctypes.data, n,
                                 C.byref(par), out.ctypes.data, cig.ctypes.data, cig.size, coff.ctypes.data, st.ctypes.data)
    assert rc == -5 and coff[n] > 8
    full_out, full_cigs, _ = ctx.align_batch(pairs, par)
    assert int(coff[n]) == sum(len(c) for c in full_cigs)


def test_many_short_pairs_take_the_multi_block_scan(ctx):
    """more than 262144 pairs per batch: the exclusive scans of the CIGAR compaction run over many blocks (k_scan_tile_*) and the
    staging kernel handles a pair per wave -- the batch as a whole gives what its thirds give on their own (single-block scan),
    and a sample equals the oracle"""
    import bsalign_amd as B
    rng = np.random.default_rng(31337)
    n = 300000
    lens = rng.integers(20, 61, size=n)
    pairs = []
    for k in range(n):
        T = rng.integers(0, 4, size=int(lens[k])).astype(np.uint8)
        Q = T.copy()
        if k % 3:
            Q[int(rng.integers(len(Q)))] ^= 1
        if k % 5 == 0 and len(Q) > 25:
            Q = np.delete(Q, int(rng.integers(5, 20)))
        pairs.append((Q, T))
    par = B.make_params(S.MODE_GLOBAL, 0, *SCORINGS["affine"])
    out, cigs, st = ctx.align_batch(pairs, par)
    assert not st.any()
    for lo in range(0, n, 100000):
        o2, c2, s2 = ctx.align_batch(pairs[lo:lo + 100000], par)
        assert np.array_equal(out[lo:lo + 100000], o2)
        assert all(np.array_equal(a, b) for a, b in zip(cigs[lo:lo + 100000], c2))
    for k in rng.integers(0, n, size=300):
        res, cig, m = S.oracle_align(pairs[k][0], pairs[k][1], S.MODE_GLOBAL, 0, *SCORINGS["affine"])
        got = np.array([out[k][f] for f in out.dtype.names], dtype=np.int32)
        assert np.array_equal(got, res) and np.array_equal(cigs[k], cig), k
