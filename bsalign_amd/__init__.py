"""bsalign_amd -- Python binding of libbsalign_hip.so (the MI355X implementation of bsalign's DP hot path).

This package is plumbing for tests / bench.py: it loads the in-tree C-ABI library
(include/bsalign_hip.h) with ctypes and mirrors its entry points.  There is NO CPU
fallback anywhere: if the library is missing or no GPU is usable every compute call
raises.  The host-side drop-in for C callers is include/bsalign_compat.h.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BSA_LIB_PATH") or os.path.join(_HERE, "libbsalign_hip.so")      # BSA_LIB_PATH: development builds

MODE_GLOBAL, MODE_OVERLAP, MODE_EXTEND = 0, 1, 2
ST_BAD_BASE, ST_EMPTY, ST_TRACE, ST_DEVICE = 1, 2, 4, 8

E_NAMES = {0: "OK", -1: "BSA_E_NODEVICE", -2: "BSA_E_ARG", -3: "BSA_E_NOMEM", -4: "BSA_E_HIP",
           -5: "BSA_E_CIGAR_CAP", -6: "BSA_E_UNSUPPORTED"}


class BsaError(RuntimeError):
    def __init__(self, code, msg=""):
        self.code = code
        super().__init__("%s (%d) %s" % (E_NAMES.get(code, "?"), code, msg))


class AlignParams(C.Structure):
    _fields_ = [("mode", C.c_int32), ("bandwidth", C.c_uint32), ("matrix", C.c_int8 * 16),
                ("gapo1", C.c_int8), ("gape1", C.c_int8), ("gapo2", C.c_int8), ("gape2", C.c_int8)]


class RowTask(C.Structure):
    _fields_ = [("op", C.c_uint32), ("src", C.c_uint32), ("dst", C.c_uint32), ("qoff_src", C.c_uint32), ("qoff_dst", C.c_uint32),
                ("toff", C.c_uint32), ("query", C.c_uint32), ("base", C.c_uint8), ("prof", C.c_uint8), ("reserved", C.c_uint16)]


class RowsParams(C.Structure):
    _fields_ = [("mode", C.c_int32), ("bandwidth", C.c_uint32), ("M", C.c_int8), ("X", C.c_int8), ("refbonus", C.c_int8),
                ("gapo1", C.c_int8), ("gape1", C.c_int8), ("gapo2", C.c_int8), ("gape2", C.c_int8)]


ROW_TASK_DTYPE = np.dtype([("op", np.uint32), ("src", np.uint32), ("dst", np.uint32), ("qoff_src", np.uint32), ("qoff_dst", np.uint32),
                           ("toff", np.uint32), ("query", np.uint32), ("base", np.uint8), ("prof", np.uint8), ("reserved", np.uint16)])


class SweepParams(C.Structure):
    _fields_ = [("rows", RowsParams), ("T", C.c_int32)]


SWEEP_PROG_DTYPE = np.dtype([("first_task", np.uint32), ("ntasks", np.uint32), ("first_block", np.uint32), ("reserved", np.uint32)])
SWEEP_RESULT_DTYPE = np.dtype([("maxscr", np.int32), ("maxidx", np.int32), ("maxoff", np.int32), ("reserved", np.int32)])
# the wavefront sweep's program (include/bsalign_hip.h: bsa_poa_node_t, bsa_poa_edge_t, bsa_poa_cand_t, bsa_poa_prog_t, ...)
POA_CELL_DTYPE = np.dtype([("h", np.int32), ("e", np.int8), ("q", np.int8), ("tag", np.uint16)])
POA_NODE_DTYPE = np.dtype([("rpos", np.uint32), ("gnode", np.uint32), ("first_in", np.uint32), ("n_in", np.uint16), ("base", np.uint8), ("flags", np.uint8),
                           ("in0_src", np.uint32), ("in0_movx", np.uint32), ("in0_tk", np.uint32), ("in1_src", np.uint32), ("in1_movx", np.uint32), ("in1_tk", np.uint32),
                           ("r0", np.uint32), ("r1", np.uint32)])
POA_EDGE_DTYPE = np.dtype([("src", np.uint32), ("cov", np.uint32), ("src_rpos", np.uint32), ("reserved", np.uint32)])
POA_CAND_DTYPE = np.dtype([("node", np.uint32), ("kind", np.uint32)])
POA_EVENT_DTYPE = np.dtype([("node", np.uint32), ("x", np.int32), ("bt", np.uint32)])
POA_PROG_DTYPE = np.dtype([("first_node", np.uint32), ("nnodes", np.uint32), ("first_edge", np.uint32), ("nedges", np.uint32), ("first_cand", np.uint32), ("ncands", np.uint32),
                           ("slen", np.uint32), ("event_cap", np.uint32), ("query_off", np.uint64), ("first_event", np.uint64)])
POA_RESULT_DTYPE = np.dtype([("maxscr", np.int32), ("maxidx", np.int32), ("maxoff", np.int32), ("status", np.int32), ("nevents", np.int32),
                             ("fin_node", np.int32), ("fin_x", np.int32), ("reserved", np.int32)])
ROW_OP_UPDATE, ROW_OP_MERGE, ROW_OP_INIT, ROW_OP_SCORE_TAIL, ROW_OP_SCORE_END = 0, 1, 2, 3, 4


class EditParams(C.Structure):
    _fields_ = [("mode", C.c_int32), ("bandwidth", C.c_uint32)]


class KmerParams(C.Structure):
    """bsa_kmer_params_t: k-mer size (the reference's CLI default is 13) and host threads (0 = all)"""
    _fields_ = [("ksz", C.c_uint32), ("threads", C.c_uint32)]


RESULT_DTYPE = np.dtype([(n, np.int32) for n in ("score", "qb", "qe", "tb", "te", "mat", "mis", "ins", "del", "aln")])

_lib = None


DIAGDP_PROB_DTYPE = np.dtype([("seq0", np.uint64), ("seq1", np.uint64), ("mats0", np.uint64, (4,)), ("mats1", np.uint64, (4,)),
                              ("out0", np.uint64), ("out1", np.uint64), ("mlen", np.uint32), ("mbeg", np.uint32), ("mend", np.uint32), ("W", np.uint32)])


def build():
    """compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)"""
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(_HERE, "csrc"), "all"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libbsalign_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                               "there is no CPU fallback")
        # PyTorch ships its own copy of the HIP runtime: when torch is imported AFTER this library has loaded the system's
        # libamdhip64 the process ends up with two runtimes and torch finds no GPU.  Loading torch first makes the
        # library bind to the copy torch uses (BSA_NO_TORCH_PRELOAD=1 skips this, e.g. for host-only helper processes).
        if "torch" not in sys.modules and not os.environ.get("BSA_NO_TORCH_PRELOAD"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        vp, u8p, u32p, u64p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
        L.bsa_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.bsa_ctx_destroy.argtypes = [vp]
        L.bsa_ctx_destroy.restype = None
        L.bsa_ctx_set_stream.argtypes = [vp, vp]
        L.bsa_ctx_set_workspace_limit.argtypes = [vp, C.c_size_t]
        L.bsa_ctx_sync.argtypes = [vp]
        L.bsa_last_error.argtypes = [vp]
        L.bsa_last_error.restype = C.c_char_p
        L.bsa_ctx_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_double)]
        L.bsa_ctx_last_trace_ms.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_long)]
        L.bsa_ctx_last_kernel_name.argtypes = [vp, C.c_int]
        L.bsa_ctx_last_kernel_name.restype = C.c_char_p
        L.bsa_ctx_last_handover.argtypes = [vp]
        L.bsa_ctx_last_handover.restype = C.c_long
        L.bsa_set_score_matrix.argtypes = [C.POINTER(C.c_int8), C.c_int8, C.c_int8]
        L.bsa_set_score_matrix.restype = None
        L.bsa_align_batch.argtypes = [vp, u8p, C.c_size_t, u64p, u32p, u64p, u32p, C.c_size_t, C.POINTER(AlignParams),
                                      vp, u32p, C.c_size_t, u64p, u32p]
        L.bsa_align_plan_create.argtypes = [vp, u64p, u32p, u64p, u32p, C.c_size_t, C.POINTER(AlignParams), C.POINTER(vp)]
        L.bsa_align_plan_destroy.argtypes = [vp]
        L.bsa_align_plan_destroy.restype = None
        L.bsa_align_plan_cells.argtypes = [vp]
        L.bsa_align_plan_cells.restype = C.c_double
        L.bsa_align_run.argtypes = [vp, u8p, vp, u32p, C.c_size_t, u64p, u32p]
        L.bsa_synth_stride.argtypes = [C.c_uint32]
        L.bsa_synth_stride.restype = C.c_size_t
        L.bsa_synth_pairs_host.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, C.c_uint32, C.c_uint32, u8p, u32p]
        L.bsa_synth_pairs_dev.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_size_t, C.c_uint32, C.c_uint32, u8p, u32p]
        if hasattr(L, "bsa_edit_batch"):
            L.bsa_edit_batch.argtypes = [vp, u8p, C.c_size_t, u64p, u32p, u64p, u32p, C.c_size_t, C.POINTER(EditParams),
                                         vp, u32p, C.c_size_t, u64p, u32p]
            L.bsa_edit_plan_create.argtypes = [vp, u64p, u32p, u64p, u32p, C.c_size_t, C.POINTER(EditParams), C.POINTER(vp)]
            L.bsa_edit_plan_destroy.argtypes = [vp]
            L.bsa_edit_plan_destroy.restype = None
            L.bsa_edit_plan_cells.argtypes = [vp]
            L.bsa_edit_plan_cells.restype = C.c_double
            L.bsa_edit_run.argtypes = [vp, u8p, vp, u32p, C.c_size_t, u64p, u32p]
        L.bsa_kmer_edit_batch.argtypes = [vp, u8p, C.c_size_t, u64p, u32p, u64p, u32p, C.c_size_t, C.POINTER(KmerParams),
                                          vp, u32p, C.c_size_t, u64p, u32p]
        L.bsa_rows_block_bytes.argtypes = [C.c_uint32, C.c_int8, C.c_int8, C.c_int8, C.c_int8]
        L.bsa_rows_block_bytes.restype = C.c_size_t
        L.bsa_rows_run.argtypes = [vp, vp, vp, C.c_size_t, vp, vp, vp, C.POINTER(RowsParams)]
        L.bsa_sweep_run.argtypes = [vp, vp, vp, vp, C.c_size_t, vp, vp, vp, C.POINTER(SweepParams), vp]
        L.bsa_sweep_host.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp, vp, vp, C.c_size_t, C.POINTER(SweepParams), vp, C.c_size_t, vp]
        L.bsa_poa_graph_supported.argtypes = [C.POINTER(SweepParams), C.c_uint32]
        L.bsa_poa_graph_host.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(SweepParams),
                                         vp, vp, C.c_size_t, vp, vp]
        L.bsa_poa_graph_run.argtypes = [vp, vp, C.c_size_t, vp, vp, vp, C.c_size_t, vp, C.c_uint32, C.POINTER(SweepParams), vp, vp, vp, vp, vp, vp]
        L.bsa_align_debug_rows.argtypes = [vp, C.c_uint32, u8p, C.c_size_t, C.POINTER(C.c_uint32)]
        L.bsa_diagdp_batch.argtypes = [vp, u8p, C.c_size_t, vp, C.c_size_t, u8p, C.c_size_t]
        L.bsa_diagdp_last_ms.argtypes = [vp]
        L.bsa_diagdp_last_ms.restype = C.c_double
        L.bsa_env_reload.restype = None
        _lib = L
    _sync_env()
    return _lib


_env_seen = None


def _sync_env():
    """the library reads its BSA_* knobs once; a test that flips one inside this process gets a fresh snapshot"""
    global _env_seen
    cur = tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("BSA_")))
    if cur != _env_seen:
        _lib.bsa_env_reload()
        _env_seen = cur


def _np(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def make_params(mode=MODE_GLOBAL, bandwidth=128, M=2, X=-6, O=-3, E=-2, Q=0, P=0, matrix=None):
    p = AlignParams()
    p.mode, p.bandwidth = mode, bandwidth
    if matrix is None:
        lib().bsa_set_score_matrix(p.matrix, M, X)
    else:
        for i in range(16):
            p.matrix[i] = int(matrix[i])
    p.gapo1, p.gape1, p.gapo2, p.gape2 = O, E, Q, P
    return p


def pack_pairs(pairs):
    """[(q, t), ...] of uint8 code arrays -> (seqs blob, qoff, qlen, toff, tlen)"""
    n = len(pairs)
    qlen = np.array([len(q) for q, _ in pairs], dtype=np.uint32)
    tlen = np.array([len(t) for _, t in pairs], dtype=np.uint32)
    qoff = np.zeros(n, dtype=np.uint64)
    toff = np.zeros(n, dtype=np.uint64)
    parts, acc = [], 0
    for k, (q, t) in enumerate(pairs):
        qoff[k] = acc
        parts.append(_np(q, np.uint8))
        acc += len(q)
        toff[k] = acc
        parts.append(_np(t, np.uint8))
        acc += len(t)
    seqs = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)
    if seqs.size == 0:
        seqs = np.zeros(1, dtype=np.uint8)
    return seqs, qoff, qlen, toff, tlen


DIAGDP_WALK_DTYPE = np.dtype([("nsteps", "<u4"), ("score", "<i4"), ("xi", "<i4"), ("yi", "<i4"), ("status", "<u4"), ("reserved", "<u4"), ("first_word", "<u8")])


class Context:
    """one context per (thread, device) -- bsa_ctx_create / bsa_ctx_destroy"""

    def __init__(self, device=0, workspace_limit=0):
        h = C.c_void_p()
        rc = lib().bsa_ctx_create(device, C.byref(h))
        if rc != 0:
            raise BsaError(rc, "bsa_ctx_create: no usable GPU; this library has no CPU fallback")
        self.h = h
        if workspace_limit:
            lib().bsa_ctx_set_workspace_limit(self.h, workspace_limit)

    def close(self):
        if self.h:
            lib().bsa_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise BsaError(rc, (lib().bsa_last_error(self.h) or b"").decode())

    def set_stream(self, stream_ptr):
        self._chk(lib().bsa_ctx_set_stream(self.h, C.c_void_p(stream_ptr)))

    def sync(self):
        self._chk(lib().bsa_ctx_sync(self.h))

    def last_kernel_ms(self):
        ms, n, cells = C.c_double(), C.c_long(), C.c_double()
        self._chk(lib().bsa_ctx_last_kernel_ms(self.h, C.byref(ms), C.byref(n), C.byref(cells)))
        return ms.value, n.value, cells.value

    def last_trace_ms(self):
        ms, n = C.c_double(), C.c_long()
        self._chk(lib().bsa_ctx_last_trace_ms(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def last_handover(self):
        """pairs of the last align_batch call that were re-run through the literal kernels"""
        return int(lib().bsa_ctx_last_handover(self.h))

    def last_kernel_names(self):
        return lib().bsa_ctx_last_kernel_name(self.h, 0).decode(), lib().bsa_ctx_last_kernel_name(self.h, 1).decode()

    def _batch(self, fn, pairs, par, cigar_cap=None):
        seqs, qoff, qlen, toff, tlen = pack_pairs(pairs)
        n = len(pairs)
        out = np.zeros(n, dtype=RESULT_DTYPE)
        status = np.zeros(max(n, 1), dtype=np.uint32)
        if cigar_cap is None:
            cigar_cap = int(qlen.sum() + tlen.sum()) + 2 * n + 16
        cig = np.zeros(cigar_cap, dtype=np.uint32)
        off = np.zeros(n + 1, dtype=np.uint64)
        rc = fn(self.h, _p(seqs), seqs.size, _p(qoff), _p(qlen), _p(toff), _p(tlen), n, C.byref(par),
                _p(out), _p(cig), cigar_cap, _p(off), _p(status))
        self._chk(rc)
        cigs = [cig[int(off[k]):int(off[k + 1])].copy() for k in range(n)]
        return out, cigs, status[:n]

    def align_batch(self, pairs, par, cigar_cap=None):
        """host-pointer form of bsa_align_batch: returns (results, [cigar arrays], status)"""
        return self._batch(lib().bsa_align_batch, pairs, par, cigar_cap)

    def sweep_host(self, tasks, progs, queries, qoff, qlen, par, nblocks, want_rows=True):
        """host-pointer form of the POA sweep (bsa_sweep_host): tasks ROW_TASK_DTYPE, progs SWEEP_PROG_DTYPE,
        queries one base per byte; returns (row blocks as uint8 [nblocks * block_bytes] or None, results)"""
        L = lib()
        tasks = np.ascontiguousarray(tasks, dtype=ROW_TASK_DTYPE)
        progs = np.ascontiguousarray(progs, dtype=SWEEP_PROG_DTYPE)
        queries = np.ascontiguousarray(queries, dtype=np.uint8)
        qoff = np.ascontiguousarray(qoff, dtype=np.uint64)
        qlen = np.ascontiguousarray(qlen, dtype=np.uint32)
        r = par.rows
        blk = L.bsa_rows_block_bytes(r.bandwidth, r.gapo1, r.gape1, r.gapo2, r.gape2)
        rows = np.zeros(nblocks * blk, dtype=np.uint8) if want_rows else None
        res = np.zeros(len(progs), dtype=SWEEP_RESULT_DTYPE)
        rc = L.bsa_sweep_host(self.h, tasks.ctypes.data, len(tasks), progs.ctypes.data, len(progs), queries.ctypes.data,
                              qoff.ctypes.data, qlen.ctypes.data, len(qlen), C.byref(par),
                              rows.ctypes.data if want_rows else None, nblocks, res.ctypes.data)
        self._chk(rc)
        return rows, res

    def poa_graph_host(self, nodes, edges, cands, progs, queries, par, events_cap, want_rows=False):
        """host-pointer form of the wavefront sweep + device traceback (bsa_poa_graph_host): nodes POA_NODE_DTYPE, edges
        POA_EDGE_DTYPE, cands POA_CAND_DTYPE, progs POA_PROG_DTYPE, queries one base per byte.
        -> (results POA_RESULT_DTYPE, events POA_EVENT_DTYPE [events_cap], rows POA_CELL_DTYPE [nnodes, bw] or None, u0 or None)"""
        L = lib()
        nodes = np.ascontiguousarray(nodes, dtype=POA_NODE_DTYPE); edges = np.ascontiguousarray(edges, dtype=POA_EDGE_DTYPE)
        cands = np.ascontiguousarray(cands, dtype=POA_CAND_DTYPE); progs = np.ascontiguousarray(progs, dtype=POA_PROG_DTYPE)
        queries = np.ascontiguousarray(queries, dtype=np.uint8)
        bw = (par.rows.bandwidth + 15) // 16 * 16
        res = np.zeros(len(progs), dtype=POA_RESULT_DTYPE)
        ev = np.zeros(max(events_cap, 1), dtype=POA_EVENT_DTYPE)
        rows = np.zeros((len(nodes), bw), dtype=POA_CELL_DTYPE) if want_rows else None
        u0 = np.zeros(len(nodes), dtype=np.int32) if want_rows else None
        rc = L.bsa_poa_graph_host(self.h, nodes.ctypes.data, len(nodes), edges.ctypes.data, len(edges), cands.ctypes.data, len(cands),
                                  progs.ctypes.data, len(progs), queries.ctypes.data, queries.size, C.byref(par), res.ctypes.data, ev.ctypes.data, events_cap,
                                  rows.ctypes.data if want_rows else None, u0.ctypes.data if want_rows else None)
        self._chk(rc)
        return res, ev, rows, u0

    def diagdp_batch(self, planes, probs, matrix_bytes):
        """anti-diagonal u8 DP of the MSA refinement (bsa_diagdp_batch): planes = uint8 blob in the reference's layout,
        probs = DIAGDP_PROB_DTYPE array; returns the matrix buffer (uint8, rows outside 2 mbeg .. 2 mend - 1 zero)"""
        L = lib()
        planes = np.ascontiguousarray(planes, dtype=np.uint8)
        probs = np.ascontiguousarray(probs, dtype=DIAGDP_PROB_DTYPE)
        matrix = np.zeros(matrix_bytes, dtype=np.uint8)
        self._chk(L.bsa_diagdp_batch(self.h, planes.ctypes.data, planes.size, probs.ctypes.data, len(probs), matrix.ctypes.data, matrix.size))
        return matrix

    def diagdp_walk_batch(self, planes, probs):
        """the same DP followed by the traceback on the device (bsa_diagdp_walk_batch): no matrix comes back.
        -> (walk records, list of uint8 step arrays: 0 diagonal, 1 x - 1, 2 y - 1)"""
        L = lib()
        planes = np.ascontiguousarray(planes, dtype=np.uint8)
        probs = np.ascontiguousarray(probs, dtype=DIAGDP_PROB_DTYPE)
        n = len(probs)
        walks = np.zeros(n, dtype=DIAGDP_WALK_DTYPE)
        cap = int(sum((2 * (int(p["mend"]) - int(p["mbeg"])) + 2 + 15) // 16 for p in probs)) + 1
        words = np.zeros(cap, dtype=np.uint32)
        L.bsa_diagdp_walk_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
        self._chk(L.bsa_diagdp_walk_batch(self.h, planes.ctypes.data, planes.size, probs.ctypes.data, n, walks.ctypes.data, words.ctypes.data, cap))
        steps = []
        for k in range(n):
            ns, fw = int(walks[k]["nsteps"]), int(walks[k]["first_word"])
            w = words[fw:fw + (ns + 15) // 16]
            st = ((w[:, None] >> (2 * np.arange(16, dtype=np.uint32))[None, :]) & 3).astype(np.uint8).reshape(-1)[:ns]
            steps.append(st)
        return walks, steps

    def diagdp_last_ms(self):
        return float(lib().bsa_diagdp_last_ms(self.h))

    def edit_batch(self, pairs, mode=MODE_GLOBAL, bandwidth=0, cigar_cap=None):
        p = EditParams()
        p.mode, p.bandwidth = mode, bandwidth
        return self._batch(lib().bsa_edit_batch, pairs, p, cigar_cap)

    def kmer_edit_batch(self, pairs, ksz=13, threads=0, cigar_cap=None):
        """k-mer anchored edit alignment (the reference's kmer_striped_seqedit_pairwise, bsalign.h:1209) of a batch"""
        p = KmerParams()
        p.ksz, p.threads = ksz, threads
        return self._batch(lib().bsa_kmer_edit_batch, pairs, p, cigar_cap)


def synth_pairs_host(n, L, eps=0.10, seed=20240611, first_pair=0):
    """host (C) form of the synthetic generator: returns list of (q, t)"""
    stride = lib().bsa_synth_stride(L)
    seqs = np.zeros(2 * n * stride, dtype=np.uint8)
    qlen = np.zeros(n, dtype=np.uint32)
    rc = lib().bsa_synth_pairs_host(seed, first_pair, n, L, int(eps * 4294967296.0), _p(seqs), _p(qlen))
    if rc != 0:
        raise BsaError(rc)
    return [(seqs[(n + k) * stride:(n + k) * stride + qlen[k]].copy(), seqs[k * stride:k * stride + L].copy()) for k in range(n)]


class AlignPlan:
    """two-phase form (bsa_align_plan_create / bsa_align_run): host metadata once, device-resident data per run.
    Device buffers are torch tensors (plumbing only); the run is asynchronous on the context's stream."""

    def __init__(self, ctx, qoff, qlen, toff, tlen, par):
        self.ctx = ctx
        self.n = len(qlen)
        self.qoff, self.qlen = _np(qoff, np.uint64), _np(qlen, np.uint32)
        self.toff, self.tlen = _np(toff, np.uint64), _np(tlen, np.uint32)
        self.par = par
        h = C.c_void_p()
        ctx._chk(lib().bsa_align_plan_create(ctx.h, _p(self.qoff), _p(self.qlen), _p(self.toff), _p(self.tlen),
                                             self.n, C.byref(par), C.byref(h)))
        self.h = h

    def cells(self):
        return lib().bsa_align_plan_cells(self.h)

    def run(self, d_seqs, d_out, d_cigar=None, d_cigar_off=None, d_status=None):
        cap = d_cigar.numel() if d_cigar is not None else 0
        self.ctx._chk(lib().bsa_align_run(self.h, C.c_void_p(d_seqs.data_ptr()), C.c_void_p(d_out.data_ptr()),
                                          C.c_void_p(d_cigar.data_ptr() if d_cigar is not None else 0), cap,
                                          C.c_void_p(d_cigar_off.data_ptr() if d_cigar_off is not None else 0),
                                          C.c_void_p(d_status.data_ptr() if d_status is not None else 0)))

    def debug_slot(self, pair, nbytes):
        """raw traceback slot of `pair` after a single-chunk run (band offsets + row records, layout in csrc/bsa_common.h)"""
        buf = np.zeros(nbytes, dtype=np.uint8)
        rowb = C.c_uint32()
        self.ctx._chk(lib().bsa_align_debug_rows(self.h, pair, _p(buf), nbytes, C.byref(rowb)))
        return buf

    def close(self):
        if self.h:
            lib().bsa_align_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EditPlan:
    """bsa_edit_plan_create / bsa_edit_run (striped_seqedit_pairwise on the device)"""

    def __init__(self, ctx, qoff, qlen, toff, tlen, mode=MODE_GLOBAL, bandwidth=0):
        self.ctx = ctx
        self.n = len(qlen)
        self.qoff, self.qlen = _np(qoff, np.uint64), _np(qlen, np.uint32)
        self.toff, self.tlen = _np(toff, np.uint64), _np(tlen, np.uint32)
        self.par = EditParams()
        self.par.mode, self.par.bandwidth = mode, bandwidth
        h = C.c_void_p()
        ctx._chk(lib().bsa_edit_plan_create(ctx.h, _p(self.qoff), _p(self.qlen), _p(self.toff), _p(self.tlen),
                                            self.n, C.byref(self.par), C.byref(h)))
        self.h = h

    def cells(self):
        return lib().bsa_edit_plan_cells(self.h)

    def run(self, d_seqs, d_out, d_cigar=None, d_cigar_off=None, d_status=None):
        cap = d_cigar.numel() if d_cigar is not None else 0
        self.ctx._chk(lib().bsa_edit_run(self.h, C.c_void_p(d_seqs.data_ptr()), C.c_void_p(d_out.data_ptr()),
                                         C.c_void_p(d_cigar.data_ptr() if d_cigar is not None else 0), cap,
                                         C.c_void_p(d_cigar_off.data_ptr() if d_cigar_off is not None else 0),
                                         C.c_void_p(d_status.data_ptr() if d_status is not None else 0)))

    def close(self):
        if self.h:
            lib().bsa_edit_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
