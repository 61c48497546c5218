"""GPU: k-mer anchored edit alignment on the device (bsa_kmer_edit_batch, compat kmer_striped_seqedit_pairwise)
against the committed results of the real reference (tests/golden/kmer_edit.npz) and, on larger random batches,
against the host pieces driven with the oracle's edit DP between the anchors."""
import ctypes as C
import os

import numpy as np
import pytest

import support as S
import kmer_support as K
from test_kmer_cpu import golden_cases, oracle_segment

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import bsalign_amd as B
    c = B.Context(0)
    yield c
    c.close()


def test_batch_equals_reference_fixture(ctx):
    cases = list(golden_cases())
    for ksz in sorted({c[1] for c in cases}):
        sel = [c for c in cases if c[1] == ksz]
        out, cigs, st = ctx.kmer_edit_batch([(c[2], c[3]) for c in sel], ksz=ksz)
        assert not st.any()
        for (k, _, q, t, res, cig), o, c in zip(sel, out, cigs):
            assert np.array_equal(np.array(o.tolist(), dtype=np.int32), res), (k, ksz, o, res)
            assert np.array_equal(c, cig), (k, ksz)


def test_batch_equals_host_pieces_on_a_random_batch(ctx):
    rng = np.random.default_rng(2024)
    pairs = []
    for it in range(96):
        L = int(rng.integers(100, 6000))
        T = rng.integers(0, 4, L).astype(np.uint8)
        Q = S.mutate(rng, T, float(rng.choice([0.02, 0.08, 0.15])))
        if it % 5 == 2 and len(Q) > 300:
            a = int(rng.integers(50, len(Q) - 100))
            Q = np.concatenate([Q[:a], Q[a + int(rng.integers(40, 90)):]])
        pairs.append((Q, T))
    for ksz, threads in ((13, 0), (9, 1)):
        out, cigs, st = ctx.kmer_edit_batch(pairs, ksz=ksz, threads=threads)
        assert not st.any()
        for k, (q, t) in enumerate(pairs):
            r, c, _ = K.kmer_host(ksz, q, t, oracle_segment)
            assert np.array_equal(np.array(out[k].tolist(), dtype=np.int32), r), (k, ksz)
            assert np.array_equal(cigs[k], c), (k, ksz)


def test_results_without_a_cigar_arena(ctx):
    import bsalign_amd as B
    rng = np.random.default_rng(9)
    T = rng.integers(0, 4, 3000).astype(np.uint8)
    pairs = [(S.mutate(rng, T, 0.1), T) for _ in range(5)]
    want, _, _ = ctx.kmer_edit_batch(pairs, ksz=13)
    seqs, qoff, qlen, toff, tlen = B.pack_pairs(pairs)
    out = np.zeros(len(pairs), dtype=B.RESULT_DTYPE)
    par = B.KmerParams()
    par.ksz, par.threads = 13, 2
    rc = B.lib().bsa_kmer_edit_batch(ctx.h, B._p(seqs), seqs.size, B._p(qoff), B._p(qlen), B._p(toff), B._p(tlen), len(pairs),
                                     C.byref(par), B._p(out), None, 0, None, None)
    assert rc == 0
    assert np.array_equal(out, want)
    # an arena that is too small is reported, not overrun
    cig = np.zeros(8, dtype=np.uint32)
    off = np.zeros(len(pairs) + 1, dtype=np.uint64)
    rc = B.lib().bsa_kmer_edit_batch(ctx.h, B._p(seqs), seqs.size, B._p(qoff), B._p(qlen), B._p(toff), B._p(tlen), len(pairs),
                                     C.byref(par), B._p(out), B._p(cig), 8, B._p(off), None)
    assert rc == -5


def test_compat_function_equals_reference_fixture():
    lib = C.CDLL(os.path.join(S.ROOT, "bsalign_amd", "libbsalign_compat.so"))

    class Res(C.Structure):
        _fields_ = [(n, C.c_int) for n in ("score", "qb", "qe", "tb", "te", "mat", "mis", "ins", "del_", "aln")]

    class U4V(C.Structure):
        _fields_ = [("buffer", C.POINTER(C.c_uint32)), ("size", C.c_ulonglong), ("cap", C.c_ulonglong), ("bits", C.c_ulonglong)]

    lib.adv_init_b1v.restype = C.c_void_p
    lib.adv_init_b1v.argtypes = [C.c_ulonglong, C.c_int, C.c_int, C.c_uint32]
    lib.init_u4v.restype = C.POINTER(U4V)
    lib.init_u4v.argtypes = [C.c_ulonglong]
    lib.kmer_striped_seqedit_pairwise.restype = Res
    lib.kmer_striped_seqedit_pairwise.argtypes = [C.c_uint8, S.u8p, C.c_uint32, S.u8p, C.c_uint32, C.c_void_p, C.POINTER(U4V), C.c_int]
    pool = lib.adv_init_b1v(1024, 0, 16, 0)
    cigars = lib.init_u4v(64)
    for k, ksz, q, t, res, cig in list(golden_cases())[:12]:
        q = np.ascontiguousarray(q)
        t = np.ascontiguousarray(t)
        q0, t0 = q.copy(), t.copy()
        r = lib.kmer_striped_seqedit_pairwise(ksz, S.ptr(q, S.u8p), len(q), S.ptr(t, S.u8p), len(t), pool, cigars, 0)
        got = np.array([getattr(r, f[0]) for f in Res._fields_], dtype=np.int32)
        assert np.array_equal(got, res), (k, got, res)
        n = cigars.contents.size
        assert np.array_equal(np.ctypeslib.as_array(cigars.contents.buffer, shape=(max(n, 1),))[:n], cig), k
        assert np.array_equal(q, q0) and np.array_equal(t, t0)


def test_degenerate_batches(ctx):
    import bsalign_amd as B
    # nothing to do
    out, cigs, st = ctx.kmer_edit_batch([], ksz=13)
    assert len(out) == 0 and len(cigs) == 0
    # pairs with an empty side give the zero result (bsalign.h:1051-1054 through :1436-1438); a base code above 3 is flagged
    rng = np.random.default_rng(1)
    t = rng.integers(0, 4, 500).astype(np.uint8)
    bad = S.mutate(rng, t, 0.05)
    bad[17] = 7
    pairs = [(np.zeros(0, np.uint8), t), (t, np.zeros(0, np.uint8)), (t.copy(), t), (bad, t)]
    out, cigs, st = ctx.kmer_edit_batch(pairs, ksz=11)
    assert all(out[0][f] == 0 for f in out.dtype.names) and len(cigs[0]) == 0
    assert all(out[1][f] == 0 for f in out.dtype.names) and len(cigs[1]) == 0
    # identical sequences: every k-mer is an anchor; the reference appends segment CIGARs without merging (bsalign.h:974-975)
    assert out[2]["mat"] == 500 and out[2]["score"] == 0 and S.cigar_str(cigs[2]) == "5M490M5M"
    assert st[3] & B.ST_BAD_BASE if hasattr(B, "ST_BAD_BASE") else st[3] != 0
    # k-mer size 0 is refused
    par = B.KmerParams()
    par.ksz, par.threads = 0, 1
    seqs, qoff, qlen, toff, tlen = B.pack_pairs(pairs[2:3])
    o = np.zeros(1, dtype=B.RESULT_DTYPE)
    assert B.lib().bsa_kmer_edit_batch(ctx.h, B._p(seqs), seqs.size, B._p(qoff), B._p(qlen), B._p(toff), B._p(tlen), 1, C.byref(par),
                                       B._p(o), None, 0, None, None) == -2
