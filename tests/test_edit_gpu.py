"""GPU parity: HIP 2-bit edit path (through the C-ABI) vs the oracle, bit-exact."""
import numpy as np
import pytest

import support as S

pytestmark = pytest.mark.gpu


def _mk(rng, L, eps, ratio):
    T = rng.integers(0, 4, size=L).astype(np.uint8)
    Q = S.mutate(rng, T, eps)
    if ratio != 1.0:
        Lq = max(1, int(len(Q) * ratio))
        Q = Q[:Lq] if Lq <= len(Q) else np.concatenate([Q, rng.integers(0, 4, size=Lq - len(Q)).astype(np.uint8)])
    if len(Q) == 0:
        Q = np.array([0], dtype=np.uint8)
    return Q, T


def _check(ctx, pairs, mode, bw):
    import bsalign_amd as B
    out, cigs, status = ctx.edit_batch(pairs, mode, bw)
    bad = []
    for k, (q, t) in enumerate(pairs):
        res, cig, n = S.oracle_edit(q, t, mode, bw)
        got = np.array([out[k][f] for f in out.dtype.names], dtype=np.int32)
        if not (status[k] == 0 and np.array_equal(got, res) and np.array_equal(cigs[k], cig)):
            bad.append("pair %d qlen %d tlen %d status %d\n  gpu %s %s\n  orc %s %s" % (
                k, len(q), len(t), status[k], got, S.cigar_str(cigs[k])[:100], res, S.cigar_str(cig)[:100]))
    assert not bad, "%d/%d pairs differ (mode %d bw %d)\n%s" % (len(bad), len(pairs), mode, bw, "\n".join(bad[:5]))


@pytest.mark.parametrize("bw", [64, 128, 256, 512])
def test_global_banded(ctx, bw):
    rng = np.random.default_rng(400 + bw)
    pairs = [_mk(rng, int(rng.choice([300, 700, 1500, 3000])), float(rng.choice([0.01, 0.1, 0.2])), float(rng.choice([1.0, 1.0, 0.9, 1.1])))
             for _ in range(96)]
    _check(ctx, pairs, S.MODE_GLOBAL, bw)


def test_global_short_queries_use_full_width(ctx):
    """qlen < bandwidth (or bandwidth 0): the band is the whole rounded query (bsalign.h:1059-1061)"""
    rng = np.random.default_rng(8)
    pairs = [_mk(rng, int(rng.choice([1, 2, 30, 63, 64, 65, 100, 129, 200, 500, 800])), 0.15, float(rng.choice([1.0, 0.8, 1.2]))) for _ in range(128)]
    _check(ctx, pairs, S.MODE_GLOBAL, 0)
    _check(ctx, pairs, S.MODE_GLOBAL, 1024)


@pytest.mark.parametrize("mode", [S.MODE_OVERLAP, S.MODE_EXTEND])
def test_overlap_extend_full_band(ctx, mode):
    rng = np.random.default_rng(21 + mode)
    pairs = [_mk(rng, int(rng.choice([5, 64, 100, 300, 800])), 0.15, float(rng.choice([1.0, 0.7, 1.2]))) for _ in range(96)]
    _check(ctx, pairs, mode, 0)


def test_benchmark_shape_100k_bw256(ctx):
    pairs = [S.synth_pair(k, 100000) for k in range(4)]
    _check(ctx, pairs, S.MODE_GLOBAL, 256)


@pytest.mark.parametrize("mode,bw", [(S.MODE_GLOBAL, 256), (S.MODE_GLOBAL, 128), (S.MODE_GLOBAL, 64), (S.MODE_OVERLAP, 256), (S.MODE_EXTEND, 192)])
def test_rows_tiled_eight_at_a_time(ctx, mode, bw, monkeypatch):
    """row format 1 (round 6; bsa_common.h): k_edit_fwd_grp32 writes a pair's rows eight to a tile, a 64-byte block per 32-bit column word, and
    k_edit_trace_wave fetches three dwords per plane and row around the walk's diagonal.  Pairs that stay on the diagonal, pairs whose indels
    carry the path across the band (the window misses: the literal lookups), rows that end a tile and targets of 1 .. 17 rows; the oracle's
    result, and the same bytes as format 0 (BSA_EDIT_TILED=0)"""
    rng = np.random.default_rng(600 + bw + mode)
    pairs = [_mk(rng, int(rng.choice([300, 700, 1500, 3000, 6000])), float(rng.choice([0.0, 0.01, 0.1, 0.2, 0.35])), float(rng.choice([1.0, 1.0, 0.97, 1.03])))
             for _ in range(80)]
    pairs += [_mk(rng, 4000, 0.3, 1.0) for _ in range(6)]
    if mode == S.MODE_GLOBAL:
        # one launch class, the whole batch at this band width: the reference widens the band of a pair whose query is much longer than its target (bsalign.h:1055-1067)
        pairs = [(q, t) for q, t in pairs if len(q) > bw and (len(q) + len(t) - 1) // len(t) + 1 <= bw]
    _check(ctx, pairs, mode, bw)
    names = ctx.last_kernel_names()
    out1, cig1, st1 = ctx.edit_batch(pairs, mode, bw)
    monkeypatch.setenv("BSA_EDIT_TILED", "0")
    out0, cig0, st0 = ctx.edit_batch(pairs, mode, bw)
    assert "tiled" not in ctx.last_kernel_names()[0]
    assert np.array_equal(out1, out0) and np.array_equal(st1, st0) and all(np.array_equal(x, y) for x, y in zip(cig1, cig0))
    if bw in (64, 128, 256) and mode == S.MODE_GLOBAL:
        assert len(pairs) > 40 and "tiled" in names[0] and "tiled" in names[1], names
    monkeypatch.delenv("BSA_EDIT_TILED")
    # ragged companions (targets of 1 .. 17 rows, queries cut in half: other launch classes join the batch, whatever kernels take them)
    rag = [(q, t[:k]) for k, (q, t) in zip(range(1, 18), pairs)] + [(q[:max(1, len(q) // 2)], t) for q, t in pairs[:6]]
    _check(ctx, rag + pairs[:8], mode, bw)


def test_golden_edit_cases(ctx):
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "edit.npz"))
    import bsalign_amd as B
    groups = {}
    for k in range(int(g["n"][0])):
        mode, bw = [int(x) for x in g["meta_%d" % k]]
        q = g["q_%d" % k]
        groups.setdefault((mode, bw), []).append(k)
    assert groups
    for (mode, bw), ks in groups.items():
        pairs = [(g["q_%d" % k], g["t_%d" % k]) for k in ks]
        out, cigs, status = ctx.edit_batch(pairs, mode, bw)
        for i, k in enumerate(ks):
            got = np.array([out[i][f] for f in out.dtype.names], dtype=np.int32)
            assert status[i] == 0 and np.array_equal(got, g["res_%d" % k]) and np.array_equal(cigs[i], g["cig_%d" % k]), (mode, bw, k)


@pytest.mark.parametrize("mode", [S.MODE_GLOBAL, S.MODE_OVERLAP, S.MODE_EXTEND])
def test_wide_bands(ctx, mode):
    """full-width bands of queries longer than 1024 bp (overlap / extend, bandwidth 0: the wave-per-pair kernel) and
    banded widths above 1024 (moving band: the generic kernel), mixed in one batch"""
    rng = np.random.default_rng(77 + mode)
    pairs = [_mk(rng, int(rng.choice([1100, 1500, 2500, 4000])), float(rng.choice([0.02, 0.1, 0.2])), float(rng.choice([1.0, 0.8, 1.2])))
             for _ in range(24)]
    _check(ctx, pairs, mode, 0)
    if mode == S.MODE_GLOBAL:
        _check(ctx, pairs, mode, 2048)
        _check(ctx, pairs, mode, 1088)


@pytest.mark.parametrize("mode", [S.MODE_GLOBAL, S.MODE_OVERLAP, S.MODE_EXTEND])
def test_wide_band_classes(ctx, mode):
    """every launch class above 1024 columns in one batch: 1 / 2 / 4 / 8 words per lane of the wave-per-pair kernel
    (up to 4096 / 8192 / 16384 / 32768 columns), the generic kernel beyond, and short pairs of the register kernels beside them;
    query lengths on both sides of the word and class borders"""
    rng = np.random.default_rng(177 + mode)
    lens = [70, 1000, 1024, 1025, 1088, 2047, 2048, 4032, 4096, 4097, 4160, 6000, 8192, 8193, 9000, 12000, 16384, 16390, 17000, 24000, 32768, 32790]
    pairs = []
    for L in lens:
        T = rng.integers(0, 4, size=max(8, int(L * float(rng.choice([0.3, 1.0, 1.1]))))).astype(np.uint8)
        Q = rng.integers(0, 4, size=L).astype(np.uint8)
        M = S.mutate(rng, T, 0.08)[:L]                # related over the common prefix, so the alignment is not trivial
        Q[:len(M)] = M
        pairs.append((Q, T))
    _check(ctx, pairs, mode, 0)
    if mode == S.MODE_GLOBAL:
        _check(ctx, pairs, mode, 3000)                # moving wide bands for the long queries, full width for the short ones


def test_wide_kernel_with_extreme_rows(ctx):
    """rows of the wave-per-pair kernel where the delta chain runs through many words: identical sequences (all
    matches), homopolymers against each other and a query that only matches at its very end"""
    rng = np.random.default_rng(5)
    T = rng.integers(0, 4, size=3000).astype(np.uint8)
    z = np.zeros(3000, dtype=np.uint8)
    pairs = [(T.copy(), T), (z, z.copy()), (z, T), (T, z), (np.concatenate([z[:2900], T[:100]]), T[:100].copy()),
             (T[:2500].copy(), np.concatenate([rng.integers(0, 4, size=400).astype(np.uint8), T[:2500]])),
             (np.tile(np.array([0, 1], dtype=np.uint8), 1500), np.tile(np.array([1, 0], dtype=np.uint8), 1400))]
    for mode in (S.MODE_GLOBAL, S.MODE_OVERLAP, S.MODE_EXTEND):
        _check(ctx, pairs, mode, 0)


def _golden_groups(name):
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name))
    groups = {}
    for k in range(int(g["n"][0])):
        mode, bw = [int(x) for x in g["meta_%d" % k]]
        groups.setdefault((mode, bw), []).append(k)
    return g, groups


def test_golden_wide_bands_from_the_reference(ctx):
    """results of the real reference (tests/golden/edit_wide.npz) on bands above 1024 columns: every launch class of
    the wave-per-pair kernel, the generic kernel, overlap / extend / bandwidth 0 and a moving wide band"""
    g, groups = _golden_groups("edit_wide.npz")
    for (mode, bw), ks in groups.items():
        out, cigs, status = ctx.edit_batch([(g["q_%d" % k], g["t_%d" % k]) for k in ks], mode, bw)
        for i, k in enumerate(ks):
            got = np.array([out[i][f] for f in out.dtype.names], dtype=np.int32)
            assert status[i] == 0 and np.array_equal(got, g["res_%d" % k]) and np.array_equal(cigs[i], g["cig_%d" % k]), (mode, bw, k)


@pytest.mark.parametrize("env", [{"BSA_EDIT_GRP": "0"}, {"BSA_EDIT_GRP": "1"}, {"BSA_EDIT_TRACE_COOP": "0"}, {"BSA_EDIT_TRACE_LANES": "16"},
                                 {"BSA_EDIT_NO_MERGE": "1"}, {"BSA_EDIT_TILED": "0"}, {"BSA_EDIT_TRACE_WAVE": "0"}, {"BSA_EDIT_TRACE_WAVE": "1"}, {"BSA_EDIT_GRP32": "0"}, {"BSA_EDIT_GRP32": "1"}], ids=lambda e: "-".join("%s=%s" % kv for kv in e.items()))
def test_golden_cases_on_every_kernel_variant(ctx, monkeypatch, env):
    """the launchers pick kernels by batch size; force each alternative (pair-per-lane / grouped forward kernels, plain /
    cooperative traceback, many pairs per wave, one walk per wave or never, no class merging) and replay the reference's results"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for name in ("edit.npz", "edit_wide.npz"):
        g, groups = _golden_groups(name)
        for (mode, bw), ks in groups.items():
            out, cigs, status = ctx.edit_batch([(g["q_%d" % k], g["t_%d" % k]) for k in ks], mode, bw)
            for i, k in enumerate(ks):
                got = np.array([out[i][f] for f in out.dtype.names], dtype=np.int32)
                assert status[i] == 0 and np.array_equal(got, g["res_%d" % k]) and np.array_equal(cigs[i], g["cig_%d" % k]), (name, mode, bw, k)


@pytest.mark.parametrize("mode,bw", [(S.MODE_GLOBAL, 256), (S.MODE_EXTEND, 0), (S.MODE_GLOBAL, 0)])
def test_flagged_pairs_do_not_disturb_their_neighbours(ctx, mode, bw):
    """a base code above 3 and empty sequences inside a batch: those pairs come back flagged with the zero result, the
    pairs around them (same wave, same group of lanes) are unaffected -- for the grouped, the pair-per-lane and the
    wave-per-pair kernels"""
    import bsalign_amd as B
    rng = np.random.default_rng(31 + mode + bw)
    pairs = [_mk(rng, int(rng.choice([300, 900, 1500, 2600])), 0.1, 1.0) for _ in range(24)]
    bad = pairs[5][0].copy()
    bad[len(bad) // 2] = 9
    pairs[5] = (bad, pairs[5][1])
    pairs[11] = (np.zeros(0, np.uint8), pairs[11][1])
    pairs[12] = (pairs[12][0], np.zeros(0, np.uint8))
    out, cigs, status = ctx.edit_batch(pairs, mode, bw)
    assert status[5] & B.ST_BAD_BASE and status[11] & B.ST_EMPTY and status[12] & B.ST_EMPTY
    for k in (5, 11, 12):
        assert all(out[k][f] == 0 for f in out.dtype.names) and len(cigs[k]) == 0
    for k, (q, t) in enumerate(pairs):
        if k in (5, 11, 12):
            continue
        res, cig, n = S.oracle_edit(q, t, mode, bw)
        got = np.array([out[k][f] for f in out.dtype.names], dtype=np.int32)
        assert status[k] == 0 and np.array_equal(got, res) and np.array_equal(cigs[k], cig), k


@pytest.mark.parametrize("mode", [S.MODE_GLOBAL, S.MODE_EXTEND])
def test_chunked_batches_with_mixed_band_classes(mode):
    """a workspace limit cuts a batch of many different band widths into several chunks (each with several forward
    launches and one traceback): results must not depend on the chunking"""
    import bsalign_amd as B
    rng = np.random.default_rng(91 + mode)
    pairs = [_mk(rng, int(rng.choice([40, 200, 700, 1300, 2100, 4200, 5000])), float(rng.choice([0.05, 0.15])), float(rng.choice([1.0, 0.9, 1.1])))
             for _ in range(120)]
    big = B.Context(0)
    out0, cig0, st0 = big.edit_batch(pairs, mode, 0)
    big.close()
    small = B.Context(0, workspace_limit=48 << 20)
    out1, cig1, st1 = small.edit_batch(pairs, mode, 0)
    small.close()
    assert (st0 == 0).all() and np.array_equal(st0, st1) and np.array_equal(out0, out1)
    for k in range(len(pairs)):
        assert np.array_equal(cig0[k], cig1[k]), k
    for k in range(0, len(pairs), 7):
        res, cig, n = S.oracle_edit(pairs[k][0], pairs[k][1], mode, 0)
        assert np.array_equal(np.array([out0[k][f] for f in out0.dtype.names], dtype=np.int32), res) and np.array_equal(cig0[k], cig)
