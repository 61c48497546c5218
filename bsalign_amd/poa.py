"""ctypes mirror of include/bsalign_poa.h -- the library's own POA graph surface (bsa_pog_*) -- for tests and bench.py.
Nothing is computed here: every call lands in libbsalign_hip.so."""
import ctypes as C

import numpy as np

from . import BsaError, SweepParams, POA_NODE_DTYPE, POA_EDGE_DTYPE, POA_CAND_DTYPE, POA_EVENT_DTYPE, POA_RESULT_DTYPE, RESULT_DTYPE, lib

POG_NODE_DTYPE = np.dtype([("header", np.uint32), ("next", np.uint32), ("prev", np.uint32), ("pos", np.int32), ("cpos", np.int32),
                           ("rid", np.uint16), ("cov", np.uint16), ("base", np.uint8), ("flags", np.uint8), ("reserved", np.uint16)])
F_BLESS, F_RDC, F_RDD, F_REF = 1, 2, 4, 8


class PogParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("alnmode", "bandwidth", "bwtrigger", "nrec", "seqcore", "M", "X", "O", "E", "Q", "P", "T", "refbonus")]


class PogSnapshot(C.Structure):
    _fields_ = [("nnodes", C.c_uint32), ("nreads", C.c_uint32), ("head", C.c_uint32), ("tail", C.c_uint32), ("nodes", C.c_void_p), ("ndoff", C.c_void_p), ("rdlen", C.c_void_p),
                ("out_off", C.c_void_p), ("out_to", C.c_void_p), ("out_cov", C.c_void_p), ("in_off", C.c_void_p), ("in_from", C.c_void_p)]


class PogGuide(C.Structure):
    _fields_ = [("reflen", C.c_uint32), ("have", C.c_int32), ("qb", C.c_int32), ("qe", C.c_int32), ("tb", C.c_int32), ("te", C.c_int32), ("cigar", C.c_void_p), ("ncigar", C.c_uint32)]


class PogRead(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("nhead", "ntail", "nsel", "bandwidth", "qlen", "slen", "qb", "qe")]


BACKEND_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32,
                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
_ready = False


def _L():
    global _ready
    L = lib()
    if not _ready:
        vp = C.c_void_p
        L.bsa_pog_create.argtypes = [C.POINTER(PogParams), C.POINTER(vp)]
        L.bsa_pog_destroy.argtypes = [vp]; L.bsa_pog_destroy.restype = None
        L.bsa_pog_clear.argtypes = [vp]; L.bsa_pog_clear.restype = None
        L.bsa_pog_add_read.argtypes = [vp, vp, C.c_uint32, C.POINTER(C.c_uint32)]
        L.bsa_pog_import.argtypes = [vp, C.POINTER(PogSnapshot), vp]
        L.bsa_pog_export.argtypes = [vp] + [C.POINTER(C.c_uint32)] * 5 + [vp] * 8
        L.bsa_pog_select.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(PogRead), C.POINTER(vp)]
        L.bsa_pog_needs_guide.argtypes = [vp, C.c_uint32]
        L.bsa_pog_place.argtypes = [vp, C.POINTER(PogGuide), vp, C.POINTER(PogRead)]
        L.bsa_pog_program.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp), C.POINTER(SweepParams)]
        L.bsa_pog_run.argtypes = [vp, vp, vp, vp, C.POINTER(vp)]
        L.bsa_pog_apply.argtypes = [vp, vp, vp]
        L.bsa_pog_aux_edges.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
        L.bsa_pog_abort.argtypes = [vp]
        L.bsa_pog_set_cpos.argtypes = [vp, vp, vp, C.c_size_t]
        L.bsa_pog_get_cpos.argtypes = [vp, vp, vp, C.c_size_t]
        L.bsa_pog_seconds.argtypes = [vp, vp]; L.bsa_pog_seconds.restype = None
        _ready = True
    return L


def _chk(rc):
    if rc != 0:
        raise BsaError(rc)


def _view(ptr, n, dt):
    if not n:
        return np.zeros(0, dt)
    buf = (C.c_uint8 * (n * np.dtype(dt).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dt).copy()


class Pog:
    """one POA window's graph inside the library"""

    def __init__(self, **par):
        d = dict(alnmode=1, bandwidth=128, bwtrigger=1, nrec=20, seqcore=40, M=2, X=-6, O=-3, E=-2, Q=-8, P=-1, T=20, refbonus=1)
        d.update({k: v for k, v in par.items() if k in d})
        self.par = PogParams(*[int(d[n]) for n, _ in PogParams._fields_])
        self.h = C.c_void_p()
        _chk(_L().bsa_pog_create(C.byref(self.par), C.byref(self.h)))
        self._keep = None

    def close(self):
        if self.h:
            _L().bsa_pog_destroy(self.h)
            self.h = C.c_void_p()

    def clear(self):
        _L().bsa_pog_clear(self.h)

    def add_read(self, bases):
        b = np.ascontiguousarray(bases, np.uint8)
        rid = C.c_uint32()
        _chk(_L().bsa_pog_add_read(self.h, b.ctypes.data, len(b), C.byref(rid)))
        return rid.value

    def import_graph(self, g):
        """g: dict(nodes POG_NODE_DTYPE, ndoff, rdlen, out_off, out_to, out_cov, in_off, in_from, head, tail)"""
        a = {k: np.ascontiguousarray(g[k], POG_NODE_DTYPE if k == "nodes" else np.uint32) for k in ("nodes", "ndoff", "rdlen", "out_off", "out_to", "out_cov", "in_off", "in_from")}
        s = PogSnapshot(len(a["nodes"]), len(a["ndoff"]), int(g["head"]), int(g["tail"]), *[a[k].ctypes.data for k in ("nodes", "ndoff", "rdlen", "out_off", "out_to", "out_cov", "in_off", "in_from")])
        _chk(_L().bsa_pog_import(self.h, C.byref(s), None))

    def export_graph(self):
        n, nr, ne, hd, tl = (C.c_uint32() for _ in range(5))
        _chk(_L().bsa_pog_export(self.h, C.byref(n), C.byref(nr), C.byref(ne), C.byref(hd), C.byref(tl), *([None] * 8)))
        g = dict(nodes=np.zeros(n.value, POG_NODE_DTYPE), ndoff=np.zeros(nr.value, np.uint32), rdlen=np.zeros(nr.value, np.uint32), out_off=np.zeros(n.value + 1, np.uint32),
                 out_to=np.zeros(ne.value, np.uint32), out_cov=np.zeros(ne.value, np.uint32), in_off=np.zeros(n.value + 1, np.uint32), in_from=np.zeros(ne.value, np.uint32))
        _chk(_L().bsa_pog_export(self.h, None, None, None, None, None, *[g[k].ctypes.data for k in ("nodes", "ndoff", "rdlen", "out_off", "out_to", "out_cov", "in_off", "in_from")]))
        g.update(head=hd.value, tail=tl.value)
        return g

    def select(self, rid, rbeg, rlen):
        rd, sel = PogRead(), C.c_void_p()
        _chk(_L().bsa_pog_select(self.h, rid, rbeg, rlen, C.byref(rd), C.byref(sel)))
        return rd, _view(sel.value, rd.nsel, np.uint32)

    def needs_guide(self, reflen):
        return bool(_L().bsa_pog_needs_guide(self.h, reflen))

    def place(self, reflen, guide=None, cigar=None, cpos=None):
        """guide = (qb, qe, tb, te) of the read against its consensus, cigar its words; cpos: columns of the selected nodes"""
        gd = PogGuide(reflen, 0, 0, 0, 0, 0, None, 0)
        cg = None
        if guide is not None:
            cg = np.ascontiguousarray(cigar, np.uint32)
            gd = PogGuide(reflen, 1, int(guide[0]), int(guide[1]), int(guide[2]), int(guide[3]), cg.ctypes.data, len(cg))
        cp = np.ascontiguousarray(cpos, np.int32) if cpos is not None else None
        rd = PogRead()
        _chk(_L().bsa_pog_place(self.h, C.byref(gd), cp.ctypes.data if cp is not None else None, C.byref(rd)))
        return rd

    def program(self):
        pn, pe, pc, pq = (C.c_void_p() for _ in range(4))
        nn, ne, nc = (C.c_size_t() for _ in range(3))
        sp = SweepParams()
        _chk(_L().bsa_pog_program(self.h, C.byref(pn), C.byref(nn), C.byref(pe), C.byref(ne), C.byref(pc), C.byref(nc), C.byref(pq), C.byref(sp)))
        return _view(pn.value, nn.value, POA_NODE_DTYPE), _view(pe.value, ne.value, POA_EDGE_DTYPE), _view(pc.value, nc.value, POA_CAND_DTYPE), pq.value, sp

    def aux_edges(self):
        p, n = C.c_void_p(), C.c_size_t()
        _chk(_L().bsa_pog_aux_edges(self.h, C.byref(p), C.byref(n)))
        return _view(p.value, n.value, np.uint64)

    def run(self, ctx=None, backend=None):
        """ctx: a bsalign_amd.Context (the device); backend: a python callable(nodes, edges, cands, query, slen) -> (result dict, events POA_EVENT_DTYPE) that
        stands in for it (fixture replay)"""
        res = np.zeros(1, POA_RESULT_DTYPE)
        ev = C.c_void_p()
        if backend is not None:
            def cb(user, n, nn, e, ne, c, nc, q, slen, par, r, events, cap):
                try:
                    out, evs = backend(_view(n, nn, POA_NODE_DTYPE), _view(e, ne, POA_EDGE_DTYPE), _view(c, nc, POA_CAND_DTYPE), _view(q, slen, np.uint8), slen)
                    if len(evs) > cap:
                        return -3
                    rr = np.zeros(1, POA_RESULT_DTYPE)
                    for k, v in out.items():
                        rr[0][k] = v
                    rr[0]["nevents"] = len(evs)
                    C.memmove(r, rr.ctypes.data, rr.nbytes)
                    if len(evs):
                        evs = np.ascontiguousarray(evs, POA_EVENT_DTYPE)
                        C.memmove(events, evs.ctypes.data, evs.nbytes)
                    return 0
                except Exception:          # noqa: a python exception must not cross the C frame
                    return -4
            self._keep = BACKEND_FN(cb)
            rc = _L().bsa_pog_run(self.h, C.cast(self._keep, C.c_void_p), None, res.ctypes.data, C.byref(ev))
        else:
            rc = _L().bsa_pog_run(self.h, None, ctx.h, res.ctypes.data, C.byref(ev))
        _chk(rc)
        return res[0], _view(ev.value, int(res[0]["nevents"]), POA_EVENT_DTYPE)

    def apply(self, nevents=0):
        rs = np.zeros(1, RESULT_DTYPE)
        gn = np.zeros(max(nevents, 1), np.uint32)
        _chk(_L().bsa_pog_apply(self.h, rs.ctypes.data, gn.ctypes.data))
        return rs[0], gn[:nevents]

    def abort(self):
        _chk(_L().bsa_pog_abort(self.h))

    def seconds(self):
        out = np.zeros(5)
        _L().bsa_pog_seconds(self.h, out.ctypes.data)
        return out
