"""The POA's binary / text MSA formats (include/bsalign_msa.h, host-only code of libbsalign_hip.so) against the REAL
reference's writers and loader (oracle/_ref: dump_binary_msa_bspoa bspoa.h:1555-1586, load_binary_msa_bspoa :1588-1685,
print_msa_bspoa :1491-1553) and against the committed fixture tests/golden/msa_formats.npz."""
import os

import numpy as np
import pytest

import support as S

os.environ.setdefault("BSA_NO_TORCH_PRELOAD", "1")
from bsalign_amd import msa as MSA  # noqa: E402

HAVE_REF = os.path.exists(os.path.join(S.ORACLE_DIR, "_ref", "libbsref.so"))
GOLD = os.path.join(S.ROOT, "tests", "golden", "msa_formats.npz")


def test_fixture_binary_and_text():
    import msa_support as M
    g = np.load(GOLD)
    for ci, (nseq, mlen) in enumerate(g["meta"]):
        nseq, mlen = int(nseq), int(mlen)
        cols, idxs = g["cols%d" % ci], g["idxs%d" % ci]
        assert MSA.binary_write(cols, idxs, nseq, mlen, b"window %d" % ci) == g["bin%d" % ci].tobytes()
        # the container read back: the same columns in MSA order, the same consensus
        n2, m2, c2, meta, used = MSA.binary_read(g["bin%d" % ci].tobytes())
        assert (n2, m2, meta, used) == (nseq, mlen, b"window %d" % ci, g["bin%d" % ci].size)
        assert np.array_equal(c2, cols.reshape(-1, nseq + 3)[idxs])
        cns, qlt, alt = MSA.consensus(c2, None, nseq, mlen)
        assert np.array_equal(cns, g["cns%d" % ci]) and np.array_equal(qlt, g["qlt%d" % ci]) and np.array_equal(alt, g["alt%d" % ci])
        for ti, (mb, me, lw) in enumerate(M.TEXT_ARGS):
            key = "txt%d_%d" % (ci, ti)
            if key not in g:
                continue
            got = MSA.text(cols, idxs, nseq, mlen, g["cns%d" % ci], g["qlt%d" % ci], g["alt%d" % ci], "W%d" % ci, mb, me, lw, g["var%d" % ci])
            assert got == g[key].tobytes(), "text differs: case %d args %s" % (ci, (mb, me, lw))


def test_reader_rejects_truncated_and_reports_sizes():
    g = np.load(GOLD)
    blob = g["bin0"].tobytes()
    for cut in (1, 5, 9, 10, len(blob) // 2, len(blob) - 1):
        with pytest.raises(Exception):
            MSA.binary_read(blob[:cut])
    # two containers back to back: `consumed` finds the second
    n, m, c, meta, used = MSA.binary_read(blob + blob)
    assert used == len(blob)
    n2, m2, c2, meta2, used2 = MSA.binary_read((blob + blob)[used:])
    assert (n, m, meta) == (n2, m2, meta2) and np.array_equal(c, c2)
    # empty MSA
    e = MSA.binary_write(np.zeros(0, np.uint8), None, 4, 0)
    assert e == bytes([0x22]) + (0).to_bytes(4, "little") + (4).to_bytes(4, "little") + bytes([0xFF])
    assert MSA.binary_read(e)[:2] == (4, 0)


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not built")
def test_against_the_reference_live():
    import msa_support as M
    import poa_support as P
    for ci, (seed, L, n, eps) in enumerate(M.CASES):
        w = M.RefWindow(P.synth_reads(seed + 100, L, n, eps))
        try:
            for meta in (b"", b"some metadata\n"):
                mine = MSA.binary_write(w.cols, w.idxs, w.nseq, w.mlen, meta)
                assert mine == w.binary(meta), "binary container differs: case %d" % ci
            # the reference's loader on our bytes, our reader on its bytes: the same MSA and consensus
            rc, rn, rm, rcols, rcns, rqlt, ralt, rmeta = M.ref_load_binary(mine)
            assert rc == 0 and (rn, rm, rmeta) == (w.nseq, w.mlen, b"some metadata\n")
            n2, m2, c2, meta2, used = MSA.binary_read(w.binary(b"some metadata\n"))
            assert (n2, m2, meta2, used) == (rn, rm, rmeta, len(mine)) and np.array_equal(c2, rcols)
            cns, qlt, alt, rdseqs, rdoffs = MSA.consensus(c2, None, n2, m2, reads=True)
            assert np.array_equal(cns, rcns) and np.array_equal(qlt, rqlt) and np.array_equal(alt, ralt)
            assert np.array_equal(cns, w.cns) and np.array_equal(qlt, w.qlt) and np.array_equal(alt, w.alt)
            assert int(rdoffs[-1]) == sum(int((c2[:, r] < 4).sum()) for r in range(n2))
            for (mb, me, lw) in M.TEXT_ARGS:
                if mb >= w.mlen:
                    continue
                got = MSA.text(w.cols, w.idxs, w.nseq, w.mlen, w.cns, w.qlt, w.alt, "LBL", mb, me, lw, w.var)
                assert got == w.text("LBL", mb, me, lw), "text differs: case %d args %s" % (ci, (mb, me, lw))
        finally:
            w.close()


import msa_support as MS  # noqa: E402
import poa_support as P  # noqa: E402


# ---- consensus calling (bsa_msa_call_consensus = cns_bspoa, bspoa.h:3457-3733) ----
def _call_consensus(w, cols, nmsa, nrds, nall, par7):
    import ctypes as C
    M = MSA
    L = M.lib()
    L.bsa_msa_call_consensus.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    cns, qlt, alt = (np.zeros(w.mlen + 1, np.uint8) for _ in range(3))
    clen, score = C.c_uint32(), C.c_double()
    rc = L.bsa_msa_call_consensus(cols.ctypes.data, w.idxs.ctypes.data, nall, min(nmsa, nrds), nrds, w.mlen, par7.ctypes.data,
                                  cns.ctypes.data, qlt.ctypes.data, alt.ctypes.data, C.byref(clen), C.byref(score))
    assert rc == 0
    return cns[:clen.value], qlt[:clen.value], alt[:clen.value], score.value


@pytest.mark.skipif(not HAVE_REF, reason="needs oracle/_ref")
@pytest.mark.parametrize("case", MS.CASES + [(15, 400, 70, (0.1,)), (16, 2500, 45, (0.08, 0.15))])
def test_consensus_calling_is_the_references(case):
    """the finished window's MSA through the product's consensus caller: consensus bases, both quality strings, the three consensus
    bytes of EVERY column and the DP's log probability (an IEEE double, compared bit for bit) equal the real cns_bspoa's.  The two
    deep windows (70 and 45 reads) take the normal-tail branch of the alternative-allele quality and seqcore < reads."""
    import ctypes as C
    seed, L, n, eps = case
    w = MS.RefWindow(P.synth_reads(seed, L, n, eps=eps))
    try:
        r = w.r
        r.ref_poa_cns_call.argtypes = [C.c_void_p]
        r.ref_poa_cns_call.restype = C.c_double
        r.ref_poa_cns_inputs.argtypes = [C.c_void_p] * 5
        r.ref_poa_cns_inputs.restype = None
        nmsa, nrds, nall = C.c_uint32(), C.c_uint32(), C.c_uint32()
        par7 = np.zeros(7, np.float32)
        r.ref_poa_cns_inputs(w.h, C.byref(nmsa), C.byref(nrds), C.byref(nall), par7.ctypes.data)
        want_score = r.ref_poa_cns_call(w.h)                    # the real function once more on the final MSA: same bytes, and its return value
        mine = w.cols.copy()
        mine.reshape(-1, w.mrow)[:, nall.value:] = 255           # nothing of the reference's answer left in the input
        cns, qlt, alt, score = _call_consensus(w, mine, nmsa.value, nrds.value, nall.value, par7)
        assert np.array_equal(cns, w.cns) and np.array_equal(qlt, w.qlt) and np.array_equal(alt, w.alt)
        used = w.idxs.astype(np.int64)
        a = mine.reshape(-1, w.mrow)[used]; b = w.cols.reshape(-1, w.mrow)[used]
        assert np.array_equal(a, b)
        assert np.float64(score).tobytes() == np.float64(want_score).tobytes()
    finally:
        w.close()


def test_consensus_calling_fixture():
    """tests/golden/cns.npz (make_golden_cns.py): MSAs of real windows and what the reference's cns_bspoa made of them"""
    import ctypes as C
    g = np.load(os.path.join(S.ROOT, "tests", "golden", "cns.npz"))
    L = MSA.lib()
    L.bsa_msa_call_consensus.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    for k in range(int(g["n"][0])):
        want = g["cols_%d" % k]
        nmsa, nrds, nall, mlen = (int(x) for x in g["dims_%d" % k])
        mine = want.copy(); mine[:, nall:] = 255
        cns, qlt, alt = (np.zeros(mlen + 1, np.uint8) for _ in range(3))
        clen, score = C.c_uint32(), C.c_double()
        par7 = np.ascontiguousarray(g["par_%d" % k], np.float32)
        assert L.bsa_msa_call_consensus(mine.ctypes.data, None, nall, min(nmsa, nrds), nrds, mlen, par7.ctypes.data, cns.ctypes.data, qlt.ctypes.data, alt.ctypes.data,
                                        C.byref(clen), C.byref(score)) == 0
        assert np.array_equal(mine, want)
        assert np.array_equal(cns[:clen.value], g["cns_%d" % k]) and np.array_equal(qlt[:clen.value], g["qlt_%d" % k]) and np.array_equal(alt[:clen.value], g["alt_%d" % k])
        assert np.float64(score.value).tobytes() == g["score_%d" % k].tobytes()
