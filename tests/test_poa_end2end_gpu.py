"""GPU: the REAL reference POA (beg/push/end_bspoa, compiled into oracle/_ref/libbsref.so, which travels to the GPU box)
with its per-read sweep align_rd_bspoacore replaced by include/bsalign_poa_adapter.h + bsa_sweep_host on the MI355X.
After every read the harness re-runs the reference's own sweep on the same graph and compares every row block and the
best end cell with what the device returned; at the end consensus, qualities and MSA must equal the untouched run."""
import ctypes as C
import time

import numpy as np
import pytest

import poa_support as P
import support as S

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not S.have_ref(), reason="oracle/_ref/libbsref.so not built")]


def _attach(ctx):
    import bsalign_amd as B
    r = P.ref_poa()
    r.ref_poa_set_device.argtypes = [C.c_void_p, C.c_void_p]
    r.ref_poa_set_device.restype = None
    r.ref_poa_set_device(C.cast(B.lib().bsa_sweep_host, C.c_void_p), ctx.h)


@pytest.mark.parametrize("kw", [dict(), dict(bandwidth=64, alnmode=0), dict(Q=0, P=0, alnmode=2), dict(bandwidth=0), dict(nrec=3)])
def test_end_bspoa_with_the_sweep_on_the_device(ctx, kw):
    _attach(ctx)
    p = P.par(**kw)
    reads = P.synth_reads(4100 + len(kw) * 7 + sum(kw.values()), 900 if kw.get("bandwidth", 1) else 300, 10)
    r0 = P.run_ref_poa(reads, 0, p, record=False)
    r3 = P.run_ref_poa(reads, 3, p, record=False)
    assert r3["bad"] == 0, "device rows / end cell differ from the reference's own sweep"
    for k in ("cns", "qlt", "alt"):
        assert np.array_equal(r0[k], r3[k]), k
    assert r0["msa"] == r3["msa"]


def test_c4_scaled_wall_time(ctx, capsys):
    """BASELINE config C4 scaled down (32 reads x 4 kbp, default POA parameters): same consensus, and the wall time of
    end_bspoa with the sweep on the CPU (reference) and on the device (one window = latency-bound, HISTORY section 4)"""
    _attach(ctx)
    p = P.par()
    reads = P.synth_reads(20240611 & 0xFFFF, 4000, 32, eps=(0.1,))
    t0 = time.time()
    r0 = P.run_ref_poa(reads, 0, p, record=False)
    t1 = time.time()
    r3 = P.run_ref_poa(reads, 3, p, record=False)
    t2 = time.time()
    assert r3["bad"] == 0 and np.array_equal(r0["cns"], r3["cns"]) and r0["msa"] == r3["msa"]
    with capsys.disabled():
        print("\n[C4 scaled] end_bspoa 32 x 4 kbp: reference %.2f s, sweep on the device %.2f s (includes the harness re-running the reference sweep for comparison)" % (t1 - t0, t2 - t1))


def test_end_bspoa_with_sweep_and_kmer_alignment_on_the_device(ctx):
    """both alignment steps of a POA read on the MI355X: the k-mer anchored edit alignment against the consensus that
    places the band (bspoa.h:2087-2090 -> bsa_kmer_edit_batch) and the sweep (align_rd_bspoacore -> bsa_sweep_host);
    consensus, qualities and MSA must equal the untouched reference run"""
    import bsalign_amd as B
    _attach(ctx)
    r = P.ref_poa()
    r.ref_poa_set_kmer.argtypes = [C.c_void_p, C.c_void_p]
    r.ref_poa_set_kmer.restype = None
    r.ref_poa_kmer_calls.argtypes = [C.c_int]
    r.ref_poa_kmer_calls.restype = C.c_long
    p = P.par()
    reads = P.synth_reads(777, 1500, 12, eps=(0.1,))
    r0 = P.run_ref_poa(reads, 0, p, record=False)
    r.ref_poa_set_kmer(C.cast(B.lib().bsa_kmer_edit_batch, C.c_void_p), ctx.h)
    try:
        before = r.ref_poa_kmer_calls(1)
        r3 = P.run_ref_poa(reads, 3, p, record=False)
        used = r.ref_poa_kmer_calls(1) - before
    finally:
        r.ref_poa_set_kmer(None, None)
    assert used >= len(reads) - 2, "the k-mer alignments did not go to the device"
    assert r3["bad"] == 0
    for k in ("cns", "qlt", "alt"):
        assert np.array_equal(r0[k], r3[k]), k
    assert r0["msa"] == r3["msa"]


def test_c4_full_size(ctx, capsys):
    """BASELINE config C4 at its stated size: 64 ONT-like reads x 20 kbp, default POA parameters (overlap mode, bandwidth
    128, 2-piece gaps; the first read is aligned with the whole read as band, i.e. the run-time-W sweep at bw ~ 20000).
    The real end_bspoa with every sweep on the MI355X gives the untouched run's consensus, qualities and MSA, and after
    every read the device's row blocks / best end cell equal the reference's own sweep on the same graph."""
    _attach(ctx)
    p = P.par()
    reads = P.synth_reads(20240611 & 0xFFFF, 20000, 64, eps=(0.1,))
    t0 = time.time()
    r0 = P.run_ref_poa(reads, 0, p, record=False)
    t1 = time.time()
    r3 = P.run_ref_poa(reads, 3, p, record=False)
    t2 = time.time()
    assert r3["bad"] == 0
    for k in ("cns", "qlt", "alt"):
        assert np.array_equal(r0[k], r3[k]), k
    assert r0["msa"] == r3["msa"]
    with capsys.disabled():
        print("\n[C4 full size] end_bspoa 64 x 20 kbp: reference %.1f s, with the sweeps on the device (one window, plus the harness's re-check of every read) %.1f s" % (t1 - t0, t2 - t1))
