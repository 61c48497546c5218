"""Helpers for the POA sweep tests: ctypes access to the reference POA harness (oracle/_ref, only in the build
container) and to the oracle's orc_sweep_run, plus the fixture format of tests/golden/poa_sweep.npz."""
import ctypes as C
import os

import numpy as np

import support as S

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden", "poa_sweep.npz")

TASK_DTYPE = np.dtype([("op", np.uint32), ("src", np.uint32), ("dst", np.uint32), ("qoff_src", np.uint32), ("qoff_dst", np.uint32),
                       ("toff", np.uint32), ("query", np.uint32), ("base", np.uint8), ("prof", np.uint8), ("reserved", np.uint16)])
PROG_DTYPE = np.dtype([("first_task", np.uint32), ("ntasks", np.uint32), ("first_block", np.uint32), ("reserved", np.uint32)])
RESULT_DTYPE = np.dtype([("maxscr", np.int32), ("maxidx", np.int32), ("maxoff", np.int32), ("reserved", np.int32)])
FNV0, FNVP = 0xCBF29CE484222325, 0x100000001B3

# POA parameter sets: (bandwidth, bwtrigger, alnmode, nrec, realn, seqcore, shuffle, M, X, O, E, Q, P, T, refbonus, ksz)
DEFAULT = dict(bandwidth=128, bwtrigger=1, alnmode=1, nrec=20, realn=3, seqcore=40, shuffle=1,
               M=2, X=-6, O=-3, E=-2, Q=-8, P=-1, T=20, refbonus=1, ksz=15)
PAR_ORDER = ("bandwidth", "bwtrigger", "alnmode", "nrec", "realn", "seqcore", "shuffle", "M", "X", "O", "E", "Q", "P", "T", "refbonus", "ksz")


def par(**kw):
    d = dict(DEFAULT)
    d.update(kw)
    return d


def _poa_proto(r):
    r.ref_poa_create.restype = C.c_void_p
    r.ref_poa_create.argtypes = [C.c_int] * 16
    r.ref_poa_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    r.ref_poa_destroy.argtypes = [C.c_void_p]
    r.ref_poa_destroy.restype = None
    r.ref_poa_cns_len.argtypes = [C.c_void_p]
    r.ref_poa_cns_len.restype = C.c_uint32
    r.ref_poa_cns.argtypes = [C.c_void_p] * 4
    r.ref_poa_cns.restype = None
    r.ref_poa_msa_hash.argtypes = [C.c_void_p] * 3
    r.ref_poa_msa_hash.restype = C.c_uint64
    r.ref_poa_nrec.argtypes = [C.c_void_p]
    r.ref_poa_nrec.restype = C.c_uint32
    r.ref_poa_rec.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    r.ref_poa_rec.restype = None
    r.ref_poa_ntasks.argtypes = [C.c_void_p]
    r.ref_poa_ntasks.restype = C.c_uint64
    r.ref_poa_nquery_bytes.argtypes = [C.c_void_p]
    r.ref_poa_nquery_bytes.restype = C.c_uint64
    r.ref_poa_programs.argtypes = [C.c_void_p] * 3
    r.ref_poa_programs.restype = None
    r.ref_poa_core_stats.argtypes = [C.c_void_p] * 4
    r.ref_poa_core_stats.restype = None
    r.ref_poa_set_graph_backend.argtypes = [C.c_void_p, C.c_void_p]
    r.ref_poa_set_graph_backend.restype = None
    r.ref_poa_set_graph_host.argtypes = [C.c_void_p, C.c_void_p]
    r.ref_poa_set_graph_host.restype = None
    r.ref_poa_set_batcher_graph.argtypes = [C.c_void_p]
    r.ref_poa_set_batcher_graph.restype = None
    r.ref_poa_graph_rec.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    r.ref_poa_graph_rec.restype = None
    r.ref_poa_graph_sizes.argtypes = [C.c_void_p, C.c_void_p]
    r.ref_poa_graph_sizes.restype = None
    r.ref_poa_graph_data.argtypes = [C.c_void_p] * 5
    r.ref_poa_graph_data.restype = None
    r.ref_poa_form_counts.argtypes = [C.c_void_p] * 3
    r.ref_poa_form_counts.restype = None
    r._poa_ready = True
    return r


def ref_poa():
    r = S.ref()
    return r if getattr(r, "_poa_ready", False) else _poa_proto(r)


_TRACE = None


def have_ref_trace():
    return os.path.exists(os.path.join(S.ORACLE_DIR, "_ref", "libbsref_trace.so"))


def ref_poa_trace():
    """the same harness built against the reference WITH the test-only recording hook in alignment2graph_bspoa
    (oracle/bspoa_trace_record.diff): the reference's own traceback steps"""
    global _TRACE
    if _TRACE is None:
        _TRACE = _poa_proto(C.CDLL(os.path.join(S.ORACLE_DIR, "_ref", "libbsref_trace.so")))
        assert _TRACE.ref_poa_can_record_trace() == 1
    return _TRACE


def attach_product(r):
    """modes 8 - 10 of the harness call the bsa_pog_* entry points (include/bsalign_poa.h): hand it the handle of the REAL libbsalign_hip.so
    (it loads without a GPU; the graph surface is host code), so that every call lands in the product's own code"""
    import bsalign_amd as B
    if getattr(r, "_product_attached", False):
        return
    r.ref_poa_attach_product.argtypes = [C.c_void_p]
    r.ref_poa_attach_product.restype = C.c_int
    assert r.ref_poa_attach_product(C.c_void_p(B.lib()._handle)) == 0, "libbsalign_hip.so lacks bsa_pog_* entry points"
    r.ref_poa_pog_counts.argtypes = [C.c_void_p, C.c_void_p]
    r.ref_poa_pog_counts.restype = None
    r.ref_poa_pog_seconds.argtypes = [C.c_void_p, C.c_void_p]
    r.ref_poa_pog_seconds.restype = None
    r._product_attached = True


POG_NODE = np.dtype([("header", np.uint32), ("next", np.uint32), ("prev", np.uint32), ("pos", np.int32), ("cpos", np.int32),
                     ("rid", np.uint16), ("cov", np.uint16), ("base", np.uint8), ("flags", np.uint8), ("reserved", np.uint16)])
SNAP_HDR = ("nnodes", "nreads", "nedges", "ncigar", "nsel", "naux", "head", "tail", "have", "gqb", "gqe", "gtb", "gte", "reflen", "bandwidth", "slen", "qb", "qe", "rid", "rlen")


def parse_snapshot(blob, o):
    """one record of the harness's snapshot blob (oracle/ref_poa_harness.c: snap_record) -> dict"""
    hdr = blob[o:o + 160].view(np.int64)
    d = dict(zip(SNAP_HDR, (int(x) for x in hdr)))
    o += 160

    def take(n, dt):
        nonlocal o
        nb = n * np.dtype(dt).itemsize
        a = blob[o:o + nb].view(dt).copy()
        o += (nb + 7) & ~7
        return a
    n, nr, ne = d["nnodes"], d["nreads"], d["nedges"]
    d["nodes"] = take(n, POG_NODE); d["ndoff"] = take(nr, np.uint32); d["rdlen"] = take(nr, np.uint32)
    d["out_off"] = take(n + 1, np.uint32); d["out_to"] = take(ne, np.uint32); d["out_cov"] = take(ne, np.uint32)
    d["in_off"] = take(n + 1, np.uint32); d["in_from"] = take(ne, np.uint32)
    d["cigar"] = take(d["ncigar"], np.uint32); d["sels"] = take(d["nsel"], np.uint32); d["aux"] = take(d["naux"], np.uint64)
    return d


def run_ref_graph(reads, mode, p, record=True, lib=None, backend="oracle", refmode=0, cigars=None, realn_pass=0):
    """modes 5 / 6 of the harness (graph form of the binding).  backend "oracle": orc_wf_backend + orc_sweep_run on the CPU;
    "device": whatever the GPU test attached with ref_poa_set_graph_host / ref_poa_set_device.
    -> dict like run_ref_poa, plus per read (mode 5, record) nodes / edges / cands / trace / fin and the counts of reads per form"""
    r, o = (lib or ref_poa()), S.oracle()
    if mode >= 8:
        attach_product(r)
    if backend == "oracle":
        _wf_lib()
        r.ref_poa_set_graph_backend(C.cast(o.orc_wf_backend, C.c_void_p), None)
    else:
        r.ref_poa_set_graph_backend(None, None)
    h = r.ref_poa_create(*[int(p[k]) for k in PAR_ORDER])
    lens = np.array([len(x) for x in reads], dtype=np.uint32)
    offs = np.zeros(len(reads), dtype=np.uint64)
    offs[1:] = np.cumsum(lens)[:-1]
    blob = np.concatenate(reads).astype(np.uint8)
    # refmode: read 0 is the reference, `cigars` (one uint32 array per read, len << 4 | op) the reads' SAM CIGARs against it (bspoa.h:2055-2085);
    # realn_pass (mode 8): after the first stage a stretch of every aligned read goes through the realn entry of align_rd_bspoa (1: middle half, 2: whole read)
    if refmode or cigars is not None or realn_pass:
        r.ref_poa_set_refmode.argtypes = [C.c_void_p, C.c_int]; r.ref_poa_set_refmode.restype = None
        r.ref_poa_set_cigars.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]; r.ref_poa_set_cigars.restype = None
        r.ref_poa_set_realn_pass.argtypes = [C.c_void_p, C.c_int]; r.ref_poa_set_realn_pass.restype = None
        r.ref_poa_set_refmode(h, int(refmode))
        r.ref_poa_set_realn_pass(h, int(realn_pass))
    cig_blob = coffs = None
    if cigars is not None:
        coffs = np.zeros(len(reads) + 1, dtype=np.uint64)
        coffs[1:] = np.cumsum([len(c) for c in cigars])
        cig_blob = np.concatenate([np.asarray(c, dtype=np.uint32) for c in cigars] + [np.zeros(1, np.uint32)])
        r.ref_poa_set_cigars(h, cig_blob.ctypes.data, coffs.ctypes.data)
    bad = r.ref_poa_run(h, blob.ctypes.data, offs.ctypes.data, lens.ctypes.data, len(reads), mode,
                        C.cast(o.orc_sweep_run, C.c_void_p) if backend == "oracle" else None, int(record))
    n = r.ref_poa_cns_len(h)
    cns, qlt, alt = (np.zeros(n, np.uint8) for _ in range(3))
    r.ref_poa_cns(h, cns.ctypes.data, qlt.ctypes.data, alt.ctypes.data)
    nc, nr = C.c_uint32(), C.c_uint32()
    mh = r.ref_poa_msa_hash(h, C.byref(nc), C.byref(nr))
    sizes = np.zeros(4, np.uint64)
    r.ref_poa_graph_sizes(h, sizes.ctypes.data)
    gn, ge, gc, gt = np.zeros(int(sizes[0]), WF_NODE), np.zeros(int(sizes[1]), WF_EDGE), np.zeros(int(sizes[2]), WF_CAND), np.zeros(int(sizes[3]), WF_EVENT)
    r.ref_poa_graph_data(h, gn.ctypes.data, ge.ctypes.data, gc.ctypes.data, gt.ctypes.data)
    queries = np.zeros(int(r.ref_poa_nquery_bytes(h)), dtype=np.uint8)
    if len(queries):
        dummy = np.zeros(max(int(r.ref_poa_ntasks(h)), 1), dtype=TASK_DTYPE)
        r.ref_poa_programs(h, dummy.ctypes.data, queries.ctypes.data)
    recs = []
    for k in range(r.ref_poa_nrec(h)):
        out = np.zeros(20, np.int32)
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        r.ref_poa_rec(h, k, out.ctypes.data, C.byref(a), C.byref(b), C.byref(c))
        gr = np.zeros(11, np.int64)
        r.ref_poa_graph_rec(h, k, gr.ctypes.data)
        d = dict(rs=out[:10].copy(), maxscr=int(out[10]), maxidx=int(out[11]), maxoff=int(out[12]), bandwidth=int(out[13]), slen=int(out[14]),
                 qb=int(out[15]), piecewise=int(out[18]), mismatch=int(out[19]), rows_hash=a.value, query_off=c.value)
        if mode == 5 and (int(record) & 1):
            d.update(nodes=gn[gr[0]:gr[0] + gr[4]], edges=ge[gr[1]:gr[1] + gr[5]], cands=gc[gr[2]:gr[2] + gr[6]], trace=gt[gr[3]:gr[3] + gr[7]],
                     fin_gnode=int(gr[8]), fin_x=int(gr[9]), maxidx_local=int(gr[10]), query=queries[c.value:c.value + int(out[14])])
        recs.append(d)
    g1, g2 = C.c_uint64(), C.c_uint64()
    r.ref_poa_form_counts(h, C.byref(g1), C.byref(g2))
    secs = np.zeros(3, np.float64)
    r.ref_poa_binding_seconds.argtypes = [C.c_void_p, C.c_void_p]
    r.ref_poa_binding_seconds.restype = None
    r.ref_poa_binding_seconds(h, secs.ctypes.data)
    snaps = None
    if isinstance(record, int) and not isinstance(record, bool) and (record & 4):
        r.ref_poa_snap_count.argtypes = [C.c_void_p]; r.ref_poa_snap_count.restype = C.c_uint64
        r.ref_poa_snap_bytes.argtypes = [C.c_void_p]; r.ref_poa_snap_bytes.restype = C.c_uint64
        r.ref_poa_snap_data.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]; r.ref_poa_snap_data.restype = None
        ns, nb = int(r.ref_poa_snap_count(h)), int(r.ref_poa_snap_bytes(h))
        blob, so = np.zeros(nb, np.uint8), np.zeros(ns, np.uint64)
        r.ref_poa_snap_data(h, blob.ctypes.data, so.ctypes.data)
        snaps = [parse_snapshot(blob, int(o)) for o in so]
    pog = None
    if mode >= 8:
        pc, ps = np.zeros(9, np.uint64), np.zeros(9, np.float64)
        r.ref_poa_pog_counts(h, pc.ctypes.data)
        r.ref_poa_pog_seconds(h, ps.ctypes.data)
        pog = dict(reads=int(pc[0]), imports=int(pc[1]), declined=int(pc[2]), sel_nodes=int(pc[3]), placed_nodes=int(pc[4]), program_bytes=int(pc[5]),
                   steps=int(pc[6]), graph_nodes=int(pc[7]), graph_edges=int(pc[8]), binding_seconds=ps[:4].copy(), library_seconds=ps[4:].copy())
    r.ref_poa_destroy(h)
    return dict(bad=bad, binding_seconds=secs, cns=cns, qlt=qlt, alt=alt, msa=(mh, nc.value, nr.value), recs=recs, graph_reads=g1.value, rows_reads=g2.value, pog=pog, snaps=snaps)


def run_ref_poa(reads, mode, p, record=True):
    """mode 0: end_bspoa untouched; 1: orchestrated from the harness with the reference's own sweep; 2: sweep replaced by
    include/bsalign_poa_adapter.h + the oracle's orc_sweep_run (every read re-checked against the reference's sweep)"""
    r, o = ref_poa(), S.oracle()
    h = r.ref_poa_create(*[int(p[k]) for k in PAR_ORDER])
    lens = np.array([len(x) for x in reads], dtype=np.uint32)
    offs = np.zeros(len(reads), dtype=np.uint64)
    offs[1:] = np.cumsum(lens)[:-1]
    blob = np.concatenate(reads).astype(np.uint8)
    bad = r.ref_poa_run(h, blob.ctypes.data, offs.ctypes.data, lens.ctypes.data, len(reads), mode, C.cast(o.orc_sweep_run, C.c_void_p), int(record))
    n = r.ref_poa_cns_len(h)
    cns, qlt, alt = (np.zeros(n, np.uint8) for _ in range(3))
    r.ref_poa_cns(h, cns.ctypes.data, qlt.ctypes.data, alt.ctypes.data)
    nc, nr = C.c_uint32(), C.c_uint32()
    mh = r.ref_poa_msa_hash(h, C.byref(nc), C.byref(nr))
    recs = []
    for k in range(r.ref_poa_nrec(h)):
        out = np.zeros(20, np.int32)
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        r.ref_poa_rec(h, k, out.ctypes.data, C.byref(a), C.byref(b), C.byref(c))
        recs.append(dict(rs=out[:10].copy(), maxscr=int(out[10]), maxidx=int(out[11]), maxoff=int(out[12]), bandwidth=int(out[13]),
                         slen=int(out[14]), qb=int(out[15]), nblocks=int(out[16]), ntasks=int(out[17]), piecewise=int(out[18]),
                         mismatch=int(out[19]), rows_hash=a.value, task_off=b.value, query_off=c.value))
    tasks = np.zeros(int(r.ref_poa_ntasks(h)), dtype=TASK_DTYPE)
    queries = np.zeros(int(r.ref_poa_nquery_bytes(h)), dtype=np.uint8)
    if len(tasks):
        r.ref_poa_programs(h, tasks.ctypes.data, queries.ctypes.data)
    secs, nu, nm = C.c_double(), C.c_uint64(), C.c_uint64()
    r.ref_poa_core_stats(h, C.byref(secs), C.byref(nu), C.byref(nm))
    r.ref_poa_destroy(h)
    return dict(core_seconds=secs.value, core_updates=nu.value, core_merges=nm.value, bad=bad, cns=cns, qlt=qlt, alt=alt, msa=(mh, nc.value, nr.value), recs=recs, tasks=tasks, queries=queries)


def run_many(windows, mode, p, threads=8, record=False):
    """many POA windows through the harness (ref_poa_run_many): mode 0 / 1 on `threads` host threads with the reference's
    own sweep, mode 4 in lock-step through the product's batcher (attach it first, see tests/test_poa_batched_gpu.py).
    windows = list of read lists.  -> (per-window dicts with cns / qlt / alt / msa, wall seconds)"""
    import time
    r = ref_poa()
    r.ref_poa_run_many.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    nwin = len(windows)
    hs = (C.c_void_p * nwin)(*[r.ref_poa_create(*[int(p[k]) for k in PAR_ORDER]) for _ in range(nwin)])
    allreads = [x for w in windows for x in w]
    lens = np.array([len(x) for x in allreads], dtype=np.uint32)
    offs = np.zeros(len(allreads), dtype=np.uint64)
    offs[1:] = np.cumsum(lens)[:-1]
    blob = np.concatenate(allreads).astype(np.uint8)
    count = np.array([len(w) for w in windows], dtype=np.int32)
    first = np.zeros(nwin, dtype=np.int32)
    first[1:] = np.cumsum(count)[:-1]
    t0 = time.time(); c0 = time.process_time()
    bad = r.ref_poa_run_many(hs, nwin, blob.ctypes.data, offs.ctypes.data, lens.ctypes.data, first.ctypes.data, count.ctypes.data, mode, threads, int(record))
    secs = time.time() - t0
    run_many.last_cpu_seconds = time.process_time() - c0          # CPU time of the whole process over the run (all threads)
    assert bad == 0, "ref_poa_run_many: %d windows failed" % bad
    out = []
    bs = np.zeros(3)
    try:
        r.ref_poa_binding_seconds.argtypes = [C.c_void_p, C.c_void_p]
        r.ref_poa_binding_seconds.restype = None
        for h in hs:
            b3 = np.zeros(3)
            r.ref_poa_binding_seconds(h, b3.ctypes.data)
            bs += b3
    except AttributeError:
        pass
    run_many.last_binding_seconds = bs          # summed over the windows: building programs, inside the backend (waiting), applying walks
    run_many.last_pog_seconds = None
    if mode >= 8 and getattr(r, "_product_attached", False):
        tot = np.zeros(9)
        for h in hs:
            one = np.zeros(9)
            r.ref_poa_pog_seconds(h, one.ctypes.data)
            tot += one
        run_many.last_pog_seconds = tot         # binding: mirror, guide + columns, inside the library, reference-side surgery; library: select, place, program, run, apply
    for h in hs:
        n = r.ref_poa_cns_len(h)
        cns, qlt, alt = (np.zeros(n, np.uint8) for _ in range(3))
        r.ref_poa_cns(h, cns.ctypes.data, qlt.ctypes.data, alt.ctypes.data)
        nc, nr = C.c_uint32(), C.c_uint32()
        mh = r.ref_poa_msa_hash(h, C.byref(nc), C.byref(nr))
        d = dict(cns=cns, qlt=qlt, alt=alt, msa=(mh, nc.value, nr.value))
        if record == 2:
            # graph-form programs (bsa_poa_node_t ...) of this window, one per aligned read, with the reference's best end cell
            sizes = np.zeros(4, np.uint64)
            r.ref_poa_graph_sizes(h, sizes.ctypes.data)
            gn, ge, gc = np.zeros(int(sizes[0]), WF_NODE), np.zeros(int(sizes[1]), WF_EDGE), np.zeros(int(sizes[2]), WF_CAND)
            r.ref_poa_graph_data(h, gn.ctypes.data, ge.ctypes.data, gc.ctypes.data, None)
            queries = np.zeros(int(r.ref_poa_nquery_bytes(h)), dtype=np.uint8)
            if len(queries):
                dummy = np.zeros(max(int(r.ref_poa_ntasks(h)), 1), dtype=TASK_DTYPE)
                r.ref_poa_programs(h, dummy.ctypes.data, queries.ctypes.data)
            recs = []
            for k in range(r.ref_poa_nrec(h)):
                o = np.zeros(20, np.int32)
                a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
                r.ref_poa_rec(h, k, o.ctypes.data, C.byref(a), C.byref(b), C.byref(c))
                gr = np.zeros(11, np.int64)
                r.ref_poa_graph_rec(h, k, gr.ctypes.data)
                recs.append(dict(maxscr=int(o[10]), maxidx=int(o[11]), maxoff=int(o[12]), bandwidth=int(o[13]), slen=int(o[14]), piecewise=int(o[18]),
                                 nodes=gn[gr[0]:gr[0] + gr[4]], edges=ge[gr[1]:gr[1] + gr[5]], cands=gc[gr[2]:gr[2] + gr[6]], query=queries[c.value:c.value + int(o[14])]))
            secs_c, nu, nm = C.c_double(), C.c_uint64(), C.c_uint64()
            r.ref_poa_core_stats(h, C.byref(secs_c), C.byref(nu), C.byref(nm))
            d.update(recs=recs, core_seconds=secs_c.value, core_updates=nu.value, core_merges=nm.value)
        elif record:
            # the sweep programs of this window, one per aligned read (mode 1: recorded beside the reference's own sweep)
            recs = []
            for k in range(r.ref_poa_nrec(h)):
                o = np.zeros(20, np.int32)
                a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
                r.ref_poa_rec(h, k, o.ctypes.data, C.byref(a), C.byref(b), C.byref(c))
                recs.append(dict(maxscr=int(o[10]), maxidx=int(o[11]), maxoff=int(o[12]), bandwidth=int(o[13]), slen=int(o[14]), nblocks=int(o[16]),
                                 ntasks=int(o[17]), piecewise=int(o[18]), task_off=b.value, query_off=c.value))
            tasks = np.zeros(int(r.ref_poa_ntasks(h)), dtype=TASK_DTYPE)
            queries = np.zeros(int(r.ref_poa_nquery_bytes(h)), dtype=np.uint8)
            if len(tasks):
                r.ref_poa_programs(h, tasks.ctypes.data, queries.ctypes.data)
            secs_c, nu, nm = C.c_double(), C.c_uint64(), C.c_uint64()
            r.ref_poa_core_stats(h, C.byref(secs_c), C.byref(nu), C.byref(nm))
            d.update(recs=recs, tasks=tasks, queries=queries, core_seconds=secs_c.value, core_updates=nu.value, core_merges=nm.value)
        out.append(d)
        r.ref_poa_destroy(h)
    return out, secs


def block_bytes(bw, pw):
    return (bw * (pw + 1) + 68 + 15) & ~15


def hash_node_blocks(rows, nblocks, bw, pw, tasks):
    """FNV-1a over the used bytes of the node blocks (2..) the program writes -- all but the tail node's, which the
    reference never touches (oracle/ref_poa_harness.c hash_node_blocks)"""
    blk, used = block_bytes(bw, pw), bw * (pw + 1) + 68
    h = FNV0
    written = np.zeros(nblocks, dtype=bool)
    written[tasks["dst"][tasks["op"] <= 2]] = True
    written[:2] = False
    body = rows[: nblocks * blk].reshape(nblocks, blk)[written, :used].reshape(-1)
    # vectorised FNV is not possible (sequential); blocks are small enough for a plain loop in C-speed chunks
    for b in body.tobytes():
        h = ((h ^ b) * FNVP) & 0xFFFFFFFFFFFFFFFF
    return h


def oracle_sweep(tasks, progs, queries, qoff, qlen, p, bandwidth, nblocks, pw):
    o = S.oracle()
    o.orc_sweep_run.restype = None
    o.orc_sweep_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int, C.c_uint32] + [C.c_int] * 8 + [C.c_void_p]
    tasks = np.ascontiguousarray(tasks, dtype=TASK_DTYPE)
    progs = np.ascontiguousarray(progs, dtype=PROG_DTYPE)
    queries = np.ascontiguousarray(queries, dtype=np.uint8)
    qoff = np.ascontiguousarray(qoff, dtype=np.uint64)
    qlen = np.ascontiguousarray(qlen, dtype=np.uint32)
    rows = np.zeros(nblocks * block_bytes(bandwidth, pw), dtype=np.uint8)
    res = np.zeros(len(progs), dtype=RESULT_DTYPE)
    o.orc_sweep_run(rows.ctypes.data, tasks.ctypes.data, progs.ctypes.data, len(progs), queries.ctypes.data, qoff.ctypes.data, qlen.ctypes.data,
                    int(p["alnmode"]), int(bandwidth), int(p["M"]), int(p["X"]), int(p["refbonus"]), int(p["O"]), int(p["E"]), int(p["Q"]), int(p["P"]),
                    int(p["T"]), res.ctypes.data)
    return rows, res


def synth_reads(seed, L, n, eps=(0.05, 0.1, 0.15)):
    rng = np.random.default_rng(seed)
    T = rng.integers(0, 4, size=L).astype(np.uint8)
    return [S.mutate(rng, T, float(rng.choice(eps))) for _ in range(n)]


def load_golden():
    """-> list of cases: dict(par, programs=[dict(tasks, query, bandwidth, nblocks, piecewise, maxscr, maxidx, maxoff, rows_hash, rs)])"""
    g = np.load(GOLDEN)
    cases = []
    for c in range(int(g["ncases"][0])):
        pv = g["par_%d" % c]
        p = {k: int(v) for k, v in zip(PAR_ORDER, pv)}
        meta = g["meta_%d" % c]
        tasks, queries = g["tasks_%d" % c].view(TASK_DTYPE), g["queries_%d" % c]
        hashes = g["hash_%d" % c]
        progs = []
        for k in range(meta.shape[0]):
            m = meta[k]
            progs.append(dict(rs=m[:10].copy(), maxscr=int(m[10]), maxidx=int(m[11]), maxoff=int(m[12]), bandwidth=int(m[13]), slen=int(m[14]),
                              nblocks=int(m[16]), ntasks=int(m[17]), piecewise=int(m[18]), rows_hash=int(hashes[k]),
                              tasks=tasks[int(m[20]):int(m[20]) + int(m[17])], query=queries[int(m[21]):int(m[21]) + int(m[14])]))
        cases.append(dict(par=p, programs=progs, cns=g["cns_%d" % c]))
    return cases


# ---- second formulation of the sweep (oracle/bsalign_oracle_wf.c, device: bsa_poa_wf.hip): nodes / inputs / in-edges / candidates ----
WF_CELL = np.dtype([("h", np.int32), ("e", np.int8), ("q", np.int8), ("tag", np.uint16)])
WF_NODE = np.dtype([("rpos", np.uint32), ("gnode", np.uint32), ("first_in", np.uint32), ("n_in", np.uint16), ("base", np.uint8), ("flags", np.uint8),
                    ("in0_src", np.uint32), ("in0_movx", np.uint32), ("in0_tk", np.uint32), ("in1_src", np.uint32), ("in1_movx", np.uint32), ("in1_tk", np.uint32),
                    ("r0", np.uint32), ("r1", np.uint32)])
WF_EDGE = np.dtype([("src", np.uint32), ("cov", np.uint32), ("src_rpos", np.uint32), ("reserved", np.uint32)])
WF_CAND = np.dtype([("node", np.uint32), ("kind", np.uint32)])
WF_EVENT = np.dtype([("node", np.uint32), ("x", np.int32), ("bt", np.uint32)])
WF_PROG = np.dtype([("first_node", np.uint32), ("nnodes", np.uint32), ("first_edge", np.uint32), ("nedges", np.uint32), ("first_cand", np.uint32), ("ncands", np.uint32),
                    ("slen", np.uint32), ("event_cap", np.uint32), ("query_off", np.uint64), ("first_event", np.uint64)])
WF_RESULT = np.dtype([("maxscr", np.int32), ("maxidx", np.int32), ("maxoff", np.int32), ("status", np.int32), ("nevents", np.int32),
                      ("fin_node", np.int32), ("fin_x", np.int32), ("reserved", np.int32)])
WF_PARAMS = np.dtype([("mode", np.int32), ("bandwidth", np.uint32), ("M", np.int8), ("X", np.int8), ("refbonus", np.int8),
                      ("gapo1", np.int8), ("gape1", np.int8), ("gapo2", np.int8), ("gape2", np.int8), ("pad", np.int8), ("T", np.int32)])   # == bsa_sweep_params_t
IN_PRESENT, IN_MERGE, IN_SAME = 0x80000000, 0x40000000, 0x20000000
assert WF_NODE.itemsize == 48 and WF_PROG.itemsize == 48 and WF_PARAMS.itemsize == 20


def wf_params(p, bandwidth):
    a = np.zeros(1, WF_PARAMS)
    a[0] = (p["alnmode"], bandwidth, p["M"], p["X"], p["refbonus"], p["O"], p["E"], p["Q"], p["P"], 0, p["T"])
    return a


def tasks_to_graph(tasks):
    """a recorded task program (include/bsalign_hip.h bsa_row_task_t) in the node form of bsa_poa_node_t: graph nodes in completion
    order, every node's in-edges folded two at a time through partial nodes in front of it, candidates in visiting order.  The
    traceback view comes out in visiting order with cov 1 (the tasks carry no edge coverage): good for the forward pass only.
    -> nodes, edges, cands, block_of_node (mmidx per local index, 0 for partial nodes)"""
    last, ins, rpos, base, bonus, cands, gn = {}, {}, {}, {}, {}, [], {}
    pend = None
    for i, t in enumerate(tasks):
        op = int(t["op"])
        if op == 2:
            last[int(t["dst"])] = i; ins.setdefault(int(t["dst"]), []); rpos[int(t["dst"])] = 0; base[int(t["dst"])] = 4; bonus[int(t["dst"])] = 0
        elif op == 0:
            d = int(t["dst"])
            rpos[int(t["src"])] = int(t["qoff_src"])
            if d == 1:
                pend = t
            else:
                last[d] = i; ins.setdefault(d, []).append(t); rpos[d] = int(t["qoff_dst"]); base[d] = int(t["base"]); bonus[d] = int(t["prof"]) & 1
        elif op == 1:
            d = int(t["dst"])
            last[d] = i; ins.setdefault(d, []).append(pend); rpos[d] = int(pend["qoff_dst"]); base[d] = int(pend["base"]); bonus[d] = int(pend["prof"]) & 1
        else:
            cands.append((int(t["src"]), 0 if op == 3 else 1)); gn[int(t["src"])] = int(t["toff"]); rpos[int(t["src"])] = int(t["qoff_src"])
    order = sorted(last, key=lambda m: last[m])
    loc, recs, edges, blocks = {}, [], [], []

    def upd(t):
        return (loc[int(t["src"])], int(t["qoff_dst"]) - int(t["qoff_src"]), IN_PRESENT | (IN_SAME if int(t["prof"]) & 2 else 0) | int(t["toff"]))

    for m in order:
        tl = ins[m]
        first_edge = len(edges)
        for t in tl:
            edges.append((loc[int(t["src"])], 1, int(t["qoff_src"]), 0))
        none = (0, 0, 0)
        if len(tl) <= 2:
            a = upd(tl[0]) if len(tl) > 0 else none
            b = upd(tl[1]) if len(tl) > 1 else none
        else:
            recs.append((rpos[m], 0xFFFFFFFF, first_edge, 0, base[m], bonus[m]) + upd(tl[0]) + upd(tl[1]) + (0, 0)); blocks.append(0)
            for t in tl[2:-1]:
                recs.append((rpos[m], 0xFFFFFFFF, first_edge, 0, base[m], bonus[m]) + (len(recs) - 1, 0, IN_PRESENT | IN_MERGE) + upd(t) + (0, 0)); blocks.append(0)
            a = (len(recs) - 1, 0, IN_PRESENT | IN_MERGE)
            b = upd(tl[-1])
        loc[m] = len(recs)
        recs.append((rpos[m], gn.get(m, 0x80000000 | m), first_edge, len(tl), base[m], bonus[m]) + a + b + (0, 0)); blocks.append(m)
    nodes = np.array(recs, dtype=WF_NODE)
    e = np.array(edges, dtype=WF_EDGE) if edges else np.zeros(0, WF_EDGE)
    c = np.array([(loc[m], k) for m, k in cands], dtype=WF_CAND) if cands else np.zeros(0, WF_CAND)
    return nodes, e, c, np.array(blocks, dtype=np.uint32)


def _wf_lib():
    o = S.oracle()
    if not getattr(o, "_wf_ready", False):
        o.orc_wf_forward.restype = None
        o.orc_wf_forward.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        o.orc_wf_row_to_block.restype = None
        o.orc_wf_row_to_block.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_int, C.c_void_p]
        o.orc_wf_best.restype = None
        o.orc_wf_best.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        o.orc_wf_trace.restype = C.c_long
        o.orc_wf_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int,
                                   C.c_void_p, C.c_long, C.c_void_p]
        o._wf_ready = True
    return o


def oracle_wf_forward(nodes, query, p, bandwidth):
    """-> rows (nnodes x bw cells), u0"""
    o = _wf_lib()
    bw = (bandwidth + 15) // 16 * 16
    nodes = np.ascontiguousarray(nodes); query = np.ascontiguousarray(query, dtype=np.uint8)
    rows = np.zeros((len(nodes), bw), WF_CELL)
    u0 = np.zeros(len(nodes), np.int32)
    par = wf_params(p, bandwidth)
    o.orc_wf_forward(nodes.ctypes.data, len(nodes), query.ctypes.data, len(query), par.ctypes.data, rows.ctypes.data, u0.ctypes.data)
    return rows, u0


def wf_rows_to_blocks(rows, u0, block_of_node, nblocks, bandwidth, pw):
    """absolute rows -> the reference's row blocks (partial nodes, block 0, are skipped)"""
    o = _wf_lib()
    bw = (bandwidth + 15) // 16 * 16
    blk = block_bytes(bw, pw)
    out = np.zeros(nblocks * blk, np.uint8)
    rows = np.ascontiguousarray(rows)
    for k in range(len(rows)):
        if block_of_node[k]:
            o.orc_wf_row_to_block(rows[k].ctypes.data, int(u0[k]), bw, pw, out.ctypes.data + int(block_of_node[k]) * blk)
    return out


def oracle_wf_best(nodes, cands, slen, p, bandwidth, rows):
    o = _wf_lib()
    res = np.zeros(1, RESULT_DTYPE)
    par = wf_params(p, bandwidth)
    cands = np.ascontiguousarray(cands)
    o.orc_wf_best(nodes.ctypes.data, cands.ctypes.data, len(cands), slen, par.ctypes.data, rows.ctypes.data, res.ctypes.data)
    return res[0]


def oracle_wf_trace(nodes, edges, query, p, bandwidth, rows, u0, head, midx, xe):
    o = _wf_lib()
    ev = np.zeros(4 * (len(query) + len(nodes)) + 64, WF_EVENT)
    fin = np.zeros(2, np.int32)
    par = wf_params(p, bandwidth)
    query = np.ascontiguousarray(query, dtype=np.uint8)
    edges = np.ascontiguousarray(edges)
    n = o.orc_wf_trace(nodes.ctypes.data, edges.ctypes.data, query.ctypes.data, len(query), par.ctypes.data, rows.ctypes.data, u0.ctypes.data,
                       head, midx, xe, ev.ctypes.data, len(ev), fin.ctypes.data)
    return n, ev[:max(n, 0)], fin


GOLDEN_GRAPH = os.path.join(HERE, "golden", "poa_graph.npz")


def load_golden_graph():
    """tests/golden/poa_graph.npz (make_golden_poa_graph.py): per case the parameters and, per read, the graph-form program with the
    REFERENCE's best end cell and the steps of its own traceback.  -> list of dict(par, reads=[dict(...)])"""
    g = np.load(GOLDEN_GRAPH)
    cases = []
    for c in range(int(g["ncases"][0])):
        p = {k: int(v) for k, v in zip(PAR_ORDER, g["par_%d" % c])}
        nodes, edges, cands, trace = (g["%s_%d" % (n, c)].view(t) for n, t in (("nodes", WF_NODE), ("edges", WF_EDGE), ("cands", WF_CAND), ("trace", WF_EVENT)))
        query = g["query_%d" % c]
        reads = []
        for m in g["meta_%d" % c]:
            bw, slen, maxscr, maxidx, maxoff, fin_g, fin_x, nn, ne, nc, nt, n0, e0, c0, t0, q0 = (int(x) for x in m)
            reads.append(dict(bandwidth=bw, slen=slen, maxscr=maxscr, maxidx=maxidx, maxoff=maxoff, fin_gnode=fin_g, fin_x=fin_x, nodes=nodes[n0:n0 + nn],
                              edges=edges[e0:e0 + ne], cands=cands[c0:c0 + nc], trace=trace[t0:t0 + nt], query=query[q0:q0 + slen]))
        cases.append(dict(par=p, reads=reads))
    return cases
