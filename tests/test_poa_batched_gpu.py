"""GPU: MANY POA windows in lock-step.  Every window runs the real reference's end_bspoa orchestration (oracle/_ref) on a
host thread of its own; wherever that code would call align_rd_bspoacore the sweep goes through the product's batcher
(bsa_sweep_batcher_submit, bsalign_amd/csrc/bsa_batcher.hip), which executes read r of ALL windows as one device launch.
Every window's consensus, qualities, alternative bases and MSA must equal the untouched end_bspoa of the same reads."""
import ctypes as C

import numpy as np
import pytest

import poa_support as P
import support as S

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not S.have_ref(), reason="oracle/_ref/libbsref.so not built")]


class Batcher:
    def __init__(self, ctx, participants, gate=True):
        import bsalign_amd as B
        self.L = B.lib()
        self.L.bsa_sweep_batcher_create.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        self.L.bsa_sweep_batcher_destroy.argtypes = [C.c_void_p]
        self.L.bsa_sweep_batcher_destroy.restype = None
        self.L.bsa_sweep_batcher_stats.argtypes = [C.c_void_p, C.c_void_p]
        self.L.bsa_sweep_batcher_stats.restype = None
        self.h = C.c_void_p()
        assert self.L.bsa_sweep_batcher_create(ctx.h, participants, C.byref(self.h)) == 0
        r = P.ref_poa()
        r.ref_poa_set_batcher.argtypes = [C.c_void_p] * 3
        r.ref_poa_set_batcher.restype = None
        r.ref_poa_set_batcher(C.cast(self.L.bsa_sweep_batcher_submit, C.c_void_p), C.cast(self.L.bsa_sweep_batcher_leave, C.c_void_p), self.h)
        r.ref_poa_set_batcher_graph(C.cast(self.L.bsa_poa_batcher_submit_graph, C.c_void_p))
        r.ref_poa_set_batcher_enter.argtypes = [C.c_void_p]
        r.ref_poa_set_batcher_enter.restype = None
        r.ref_poa_set_batcher_enter(C.cast(self.L.bsa_sweep_batcher_enter, C.c_void_p) if gate else None)

    def stats(self):
        out = np.zeros(8, np.uint64)
        self.L.bsa_sweep_batcher_stats(self.h, out.ctypes.data)
        return dict(zip(("batches", "launches", "programs", "tasks", "bytes_up", "bytes_down", "device_us", "wall_us"), (int(x) for x in out)))

    def close(self):
        P.ref_poa().ref_poa_set_batcher(None, None, None)
        P.ref_poa().ref_poa_set_batcher_graph(None)
        self.L.bsa_sweep_batcher_destroy(self.h)


def _compare(a, b):
    for w, (x, y) in enumerate(zip(a, b)):
        for k in ("cns", "qlt", "alt"):
            assert np.array_equal(x[k], y[k]), (w, k)
        assert x["msa"] == y["msa"], w


@pytest.mark.parametrize("kw", [dict(), dict(bandwidth=64, alnmode=0), dict(Q=0, P=0, alnmode=2)])
def test_windows_of_different_sizes_in_lock_step(ctx, kw):
    """ragged: windows with different read counts and lengths leave the batcher at different times"""
    p = P.par(**kw)
    rng = np.random.default_rng(5 + len(kw))
    windows = [P.synth_reads(900 + w, int(rng.integers(200, 700)), int(rng.integers(3, 9))) for w in range(12)]
    ref, _ = P.run_many(windows, 0, p, threads=4)
    for mode in (4, 7):            # 4: row blocks come back, the reference's host traceback; 7: graph form, sweep and walk on the device
        bt = Batcher(ctx, len(windows))
        try:
            dev, _ = P.run_many(windows, mode, p)
            st = bt.stats()
        finally:
            bt.close()
        _compare(ref, dev)
        assert st["programs"] >= sum(len(w) - 1 for w in windows) and st["launches"] < st["programs"]       # programs were really run together


def test_256_windows_of_c4_shaped_reads(ctx, capsys):
    """256 windows x 12 reads x 1.5 kbp (C4's shape, scaled in length and depth), default POA parameters: identical
    results, and the wall times of the two ways of running them"""
    p = P.par()
    windows = [P.synth_reads(7000 + w, 1500, 12, eps=(0.1,)) for w in range(256)]
    ref, t_ref = P.run_many(windows, 0, p, threads=16)
    c_ref = P.run_many.last_cpu_seconds
    for mode, what in ((7, "graph form: sweep and walk on the device"), (4, "rows form: row blocks back, host traceback")):
        bt = Batcher(ctx, len(windows))
        try:
            dev, t_dev = P.run_many(windows, mode, p)
            c_dev = P.run_many.last_cpu_seconds
            b_dev = P.run_many.last_binding_seconds
            st = bt.stats()
        finally:
            bt.close()
        _compare(ref, dev)
        with capsys.disabled():
            print("\n[256 windows x 12 reads x 1.5 kbp] reference end_bspoa on 16 host threads %.2f s (%.1f CPU-seconds); through the batcher, %s %.2f s (%.1f CPU-seconds; "
                  "binding, summed over the windows: building programs %.2f s, waiting for the device %.2f s, applying walks %.2f s; "
                  "%d batches, %d launches, %d programs, %.1f MB up, %.1f MB down, device %.2f s, inside batches %.2f s)"
                  % (t_ref, c_ref, what, t_dev, c_dev, b_dev[0], b_dev[1], b_dev[2], st["batches"], st["launches"], st["programs"], st["bytes_up"] / 1e6, st["bytes_down"] / 1e6, st["device_us"] / 1e6, st["wall_us"] / 1e6))
