"""GPU: the drop-in POA surface.  oracle/_ref/libbsref_patched.so is the reference's own bspoa.h with
patches/bspoa_device_sweep.diff applied (built by oracle/Makefile from a temporary patched copy) plus
include/bsalign_poa_batch.h -- what a maintainer who applies the patch ships.  beg_bspoa / push_bspoa as always, then
  * end_bspoa untouched (devsweep == NULL),
  * bsa_poa_end_one: the same end_bspoa with every sweep on the MI355X,
  * bsa_poa_end_many: all windows in lock-step through the batcher,
must give the same consensus, qualities, alternative bases and MSA."""
import ctypes as C
import os

import numpy as np
import pytest

import poa_support as P
import support as S

PATCHED = os.path.join(S.ROOT, "oracle", "_ref", "libbsref_patched.so")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(PATCHED), reason="oracle/_ref/libbsref_patched.so not built")]


def _lib(ctx):
    import bsalign_amd as B
    L = C.CDLL(PATCHED)
    L.refp_create.restype = C.c_void_p
    L.refp_create.argtypes = [C.c_int] * 16
    L.refp_destroy.argtypes = [C.c_void_p]
    L.refp_push.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.refp_end.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.refp_cns_len.argtypes = [C.c_void_p]
    L.refp_cns_len.restype = C.c_uint32
    L.refp_cns.argtypes = [C.c_void_p] * 4
    L.refp_msa_hash.argtypes = [C.c_void_p] * 3
    L.refp_msa_hash.restype = C.c_uint64
    L.refp_attach.argtypes = [C.c_void_p] * 6
    b = B.lib()
    L.refp_attach(ctx.h, C.cast(b.bsa_sweep_host, C.c_void_p), C.cast(b.bsa_sweep_batcher_create, C.c_void_p), C.cast(b.bsa_sweep_batcher_destroy, C.c_void_p),
                  C.cast(b.bsa_sweep_batcher_submit, C.c_void_p), C.cast(b.bsa_sweep_batcher_leave, C.c_void_p))
    L.refp_attach_graph.argtypes = [C.c_void_p] * 2
    L.refp_attach_graph(C.cast(b.bsa_poa_graph_host, C.c_void_p), C.cast(b.bsa_poa_batcher_submit_graph, C.c_void_p))      # graph form: sweep + walk on the device
    L.refp_attach_enter.argtypes = [C.c_void_p]
    L.refp_attach_enter(C.cast(b.bsa_sweep_batcher_enter, C.c_void_p))
    # the library's own POA graph surface (include/bsalign_poa.h): bsa_poa_end_one / _end_many select, place, build programs and do the surgery on it
    L.refp_attach_product.argtypes = [C.c_void_p]
    L.refp_attach_product.restype = C.c_int
    assert L.refp_attach_product(C.c_void_p(b._handle)) == 0
    return L


def _run(L, windows, p, how):
    hs = []
    for reads in windows:
        h = L.refp_create(*[int(p[k]) for k in P.PAR_ORDER])
        lens = np.array([len(x) for x in reads], dtype=np.uint32)
        offs = np.zeros(len(reads), dtype=np.uint64)
        offs[1:] = np.cumsum(lens)[:-1]
        blob = np.concatenate(reads).astype(np.uint8)
        L.refp_push(h, blob.ctypes.data, offs.ctypes.data, lens.ctypes.data, len(reads))
        hs.append(h)
    arr = (C.c_void_p * len(hs))(*hs)
    assert L.refp_end(arr, len(hs), how) == 0
    out = []
    for h in hs:
        n = L.refp_cns_len(h)
        cns, qlt, alt = (np.zeros(n, np.uint8) for _ in range(3))
        L.refp_cns(h, cns.ctypes.data, qlt.ctypes.data, alt.ctypes.data)
        nc, nr = C.c_uint32(), C.c_uint32()
        mh = L.refp_msa_hash(h, C.byref(nc), C.byref(nr))
        out.append((cns, qlt, alt, (mh, nc.value, nr.value)))
        L.refp_destroy(h)
    return out


def _same(a, b):
    for w, (x, y) in enumerate(zip(a, b)):
        assert all(np.array_equal(x[k], y[k]) for k in range(3)) and x[3] == y[3], w


def test_patched_end_bspoa_one_window_and_many(ctx, monkeypatch):
    monkeypatch.setenv("BSA_POA_MIN_WINDOWS", "1")          # the device whatever the number of windows (the default attaches it from 64 windows in flight)
    L = _lib(ctx)
    p = P.par()
    rng = np.random.default_rng(3)
    windows = [P.synth_reads(300 + w, int(rng.integers(300, 1200)), int(rng.integers(4, 12)), eps=(0.1,)) for w in range(24)]
    ref = _run(L, windows, p, 0)
    _same(ref, _run(L, windows[:3], p, 1))
    _same(ref, _run(L, windows, p, 2))
    # and the untouched path of the patched header is the unpatched reference
    plain, _ = P.run_many(windows, 0, p, threads=4)
    for (cns, qlt, alt, msa), d in zip(ref, plain):
        assert np.array_equal(cns, d["cns"]) and np.array_equal(qlt, d["qlt"]) and np.array_equal(alt, d["alt"])


def test_few_windows_stay_on_the_host_and_one_window_is_never_slower(ctx, capsys):
    """the binding's policy (include/bsalign_poa_batch.h, BSA_POA_MIN_WINDOWS = 64, measured): below that many windows in flight bsa_poa_end_many / _end_one
    leave the device unattached -- the reference's own end_bspoa on host threads.  Identical results, and BASELINE's C4 as stated (ONE window of 64 x 20 kbp)
    through bsa_poa_end_one takes no longer than the untouched end_bspoa (it was 30 % slower with the device forced)"""
    import time
    L = _lib(ctx)
    p = P.par()
    rng = np.random.default_rng(4)
    windows = [P.synth_reads(900 + w, int(rng.integers(300, 1200)), int(rng.integers(4, 12)), eps=(0.1,)) for w in range(12)]
    ref = _run(L, windows, p, 0)
    _same(ref, _run(L, windows, p, 2))
    _same(ref, _run(L, windows[:2], p, 1))
    big = [P.synth_reads(20240611 & 0xFFFF, 20000, 64, eps=(0.1,))]
    t0 = time.time(); a = _run(L, big, p, 0); t_ref = time.time() - t0
    t0 = time.time(); b = _run(L, big, p, 1); t_one = time.time() - t0
    _same(a, b)
    with capsys.disabled():
        print("\n[C4 as stated, bsa_poa_end_one under the binding's default policy] reference end_bspoa %.2f s, bsa_poa_end_one %.2f s" % (t_ref, t_one))
    assert t_one <= 1.15 * t_ref + 0.2


def test_refinement_dp_on_the_device_inside_end_bspoa(ctx, monkeypatch):
    monkeypatch.setenv("BSA_POA_MIN_WINDOWS", "1")
    """patches/bspoa_device_diagdp.diff + include/bsalign_poa_diagdp.h: end_bspoa with remsa_pedits filling the DP matrices of
    all reads of a window in one bsa_diagdp_batch call (how = 3), and with the graph sweeps on the device as well (how = 4):
    consensus, qualities, alternative bases and MSA of the untouched run"""
    import time
    import bsalign_amd as B
    L = _lib(ctx)
    L.refp_attach_diagdp.argtypes = [C.c_void_p]
    L.refp_diagdp_stats.argtypes = [C.c_void_p] * 3
    L.refp_attach_diagdp(C.cast(B.lib().bsa_diagdp_batch, C.c_void_p))
    p = P.par()
    rng = np.random.default_rng(8)
    windows = [P.synth_reads(700 + w, int(rng.integers(300, 1500)), int(rng.integers(4, 16)), eps=(0.1,)) for w in range(12)]
    ref = _run(L, windows, p, 0)
    _same(ref, _run(L, windows, p, 3))
    _same(ref, _run(L, windows[:3], p, 4))
    calls, reads, steps = C.c_uint64(), C.c_uint64(), C.c_uint64()
    L.refp_diagdp_stats(C.byref(calls), C.byref(reads), C.byref(steps))
    assert calls.value >= 15 and reads.value >= sum(len(w) for w in windows)
    # one C4-shaped window (64 reads x 20 kbp): wall time of end_bspoa with and without the device DP
    big = [P.synth_reads(4242, 20000, 64, eps=(0.1,))]
    t0 = time.time(); a = _run(L, big, p, 0); t_ref = time.time() - t0
    t0 = time.time(); b = _run(L, big, p, 3); t_dev = time.time() - t0
    _same(a, b)
    # round 4: the traceback of the refinement on the device as well -- the planes stay there, two bits a step come back and the patched
    # loop replays them (how = 5; 6: with the graph sweeps on the device too).  The same consensus and MSA is the proof that every step
    # is the one the reference's own traceback takes: a different step merges different nodes.
    L.refp_attach_diagdp_walk.argtypes = [C.c_void_p]
    L.refp_attach_diagdp_walk(C.cast(B.lib().bsa_diagdp_walk_batch, C.c_void_p))
    _same(ref, _run(L, windows, p, 5))
    _same(ref, _run(L, windows[:3], p, 6))
    t0 = time.time(); c = _run(L, big, p, 5); t_walk = time.time() - t0
    _same(a, c)
    print("\n[C4 full size] end_bspoa 64 x 20 kbp: reference %.2f s, with the MSA refinement's DP on the device %.2f s, DP and traceback on the device %.2f s" % (t_ref, t_dev, t_walk))
