"""Problems of the anti-diagonal u8 DP of the reference's MSA refinement (maxmat_dp_diag_rowcal, bspoa.h:3856-3896) in the
layout remsa_pedits_bspoa builds them (bspoa.h:4213-4233, 4236-4247, 4339-4345, 4424-4440), and the two CPU checkers:
the oracle's restatement (orc_diagdp_fill) and the real reference functions (oracle/_ref, ref_diagdp_fill)."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
import support as S

u8p = C.c_void_p


def make_window(rng, mlen, nreads, W, eps=0.1, saturate=False):
    """One POA window: an MSA of `nreads` reads over `mlen` columns -> the ten planes of every read's problem.
    Returns (planes blob, list of problem dicts with the logical-index-0 offsets, mbeg, mend)."""
    bw = 16 * W
    hw = bw // 2
    plane = (mlen + bw + 15) // 16 * 16
    cns = rng.integers(0, 4, size=mlen).astype(np.uint8)
    cns[rng.random(mlen) < 0.15] = 4                         # columns where the consensus has a gap
    cols = np.full((nreads, mlen), 4, dtype=np.uint8)        # msacols: base of read r in column pos, 4 = gap
    spans = []
    for r in range(nreads):
        b = int(rng.integers(0, max(1, mlen // 8)))
        e = mlen - int(rng.integers(0, max(1, mlen // 8)))
        row = cns[b:e].copy()
        mut = rng.random(e - b)
        row[mut < eps / 3] = rng.integers(0, 4, size=int((mut < eps / 3).sum()))
        row[(mut >= eps / 3) & (mut < 2 * eps / 3)] = 4
        ins = (mut >= 2 * eps / 3) & (mut < eps) & (row == 4)
        row[ins] = rng.integers(0, 4, size=int(ins.sum()))
        if (row < 4).sum() < 2:
            row[:2] = [0, 1]
        cols[r, b:e] = row
        spans.append((b, e))
    # the shared y side (bspoa.h:4236-4247, 4339-4345): consensus reversed, profile counts reversed
    seq1 = np.full(plane, 0, dtype=np.uint8)
    mats1 = np.zeros((4, plane), dtype=np.uint8)
    for pos in range(mlen):
        seq1[hw + mlen - 1 - pos] = cns[pos]
        for b in range(4):
            cnt = int((cols[:, pos] == b).sum())
            if saturate:
                cnt = min(255, cnt * int(rng.integers(1, 60)))
            mats1[b, hw + mlen - 1 - pos] = min(cnt, 255)
    blobs = [seq1] + [mats1[b] for b in range(4)]
    shared = {"seq1": hw, "mats1": [plane * (1 + b) + hw for b in range(4)]}
    off = plane * 5
    probs = []
    for r in range(nreads):
        # the x side of one read (bspoa.h:4424-4440): its bases at their MSA positions, gaps = 4; mats[0] = length of the
        # homopolymer run behind a base (walking from the read's end)
        seq0 = np.full(plane, 4, dtype=np.uint8)
        mats0 = np.zeros((4, plane), dtype=np.uint8)
        pos_list = [p for p in range(mlen) if cols[r, p] < 4]
        lc, cc = 4, 0
        for p in reversed(pos_list):
            b = int(cols[r, p])
            seq0[hw + p] = b
            if b == lc:
                cc = min(cc + 1, 255)
                mats0[b, hw + p] = cc if not saturate else min(255, cc * 97)
            else:
                lc, cc = b, 0
        blobs += [seq0] + [mats0[b] for b in range(4)]
        probs.append({"seq0": off + hw, "mats0": [off + plane * (1 + b) + hw for b in range(4)], "seq1": shared["seq1"],
                      "mats1": shared["mats1"], "mlen": mlen, "mbeg": pos_list[0], "mend": pos_list[-1] + 1, "W": W})
        off += plane * 5
    return np.concatenate(blobs), probs


def matrix_layout(probs):
    """output offsets: two planes of (2 mlen + 1) rows of 16 W + 2 bytes per problem -> (total bytes)"""
    off = 0
    for p in probs:
        sz = (2 * p["mlen"] + 1) * (16 * p["W"] + 2)
        sz = (sz + 15) // 16 * 16
        p["out0"], p["out1"] = off, off + sz
        off += 2 * sz
    return off


def _libs():
    orc = S.oracle()
    if not hasattr(orc, "_diag"):
        orc.orc_diagdp_fill.restype = None
        orc.orc_diagdp_fill.argtypes = [u8p, u8p, C.POINTER(u8p), C.POINTER(u8p), C.c_int, C.c_int, C.c_int, C.c_int, u8p, u8p]
        orc._diag = True
    return orc


def oracle_fill(planes, probs, matrix_bytes):
    orc = _libs()
    m = np.zeros(matrix_bytes, dtype=np.uint8)
    base = planes.ctypes.data
    for p in probs:
        a0 = (u8p * 4)(*[base + o for o in p["mats0"]])
        a1 = (u8p * 4)(*[base + o for o in p["mats1"]])
        orc.orc_diagdp_fill(base + p["seq0"], base + p["seq1"], a0, a1, p["mlen"], p["mbeg"], p["mend"], p["W"],
                            m.ctypes.data + p["out0"], m.ctypes.data + p["out1"])
    return m


def oracle_walk(planes, probs, matrix):
    """orc_diagdp_walk over the planes a fill left -> list of (steps uint8 array, score, xi, yi, status)"""
    orc = _libs()
    orc.orc_diagdp_walk.restype = C.c_int
    orc.orc_diagdp_walk.argtypes = [u8p, u8p, C.POINTER(u8p), C.POINTER(u8p), C.c_int, C.c_int, C.c_int, C.c_int, u8p, u8p, u8p,
                                    C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    base = planes.ctypes.data
    out = []
    for p in probs:
        a0 = (u8p * 4)(*[base + o for o in p["mats0"]])
        a1 = (u8p * 4)(*[base + o for o in p["mats1"]])
        st = np.zeros(2 * (p["mend"] - p["mbeg"]) + 4, np.uint8)
        n, sc, xe, ye = C.c_uint32(), C.c_int(), C.c_int(), C.c_int()
        rc = orc.orc_diagdp_walk(base + p["seq0"], base + p["seq1"], a0, a1, p["mlen"], p["mbeg"], p["mend"], p["W"],
                                 matrix.ctypes.data + p["out0"], matrix.ctypes.data + p["out1"], st.ctypes.data, C.byref(n), C.byref(sc), C.byref(xe), C.byref(ye))
        out.append((st[:n.value].copy(), sc.value, xe.value, ye.value, rc))
    return out


def ref_fill(planes, probs, matrix_bytes):
    ref = S.ref()
    if not hasattr(ref, "_diag"):
        ref.ref_diagdp_fill.restype = None
        ref.ref_diagdp_fill.argtypes = [u8p] * 10 + [C.c_int] * 4 + [u8p, u8p]
        ref._diag = True
    m = np.zeros(matrix_bytes, dtype=np.uint8)
    base = planes.ctypes.data
    for p in probs:
        ref.ref_diagdp_fill(base + p["seq0"], base + p["seq1"], *[base + o for o in p["mats0"]], *[base + o for o in p["mats1"]],
                            p["mlen"], p["mbeg"], p["mend"], p["W"], m.ctypes.data + p["out0"], m.ctypes.data + p["out1"])
    return m


def written_rows(m, probs):
    """the rows a fill writes (2 mbeg .. 2 mend - 1 of both planes), concatenated: what parity is claimed on"""
    out = []
    for p in probs:
        rl = 16 * p["W"] + 2
        for o in (p["out0"], p["out1"]):
            out.append(m[o + 2 * p["mbeg"] * rl: o + 2 * p["mend"] * rl])
    return np.concatenate(out) if out else np.zeros(0, np.uint8)


def to_struct(probs):
    import bsalign_amd as B
    a = np.zeros(len(probs), dtype=B.DIAGDP_PROB_DTYPE)
    for k, p in enumerate(probs):
        for f in ("seq0", "seq1", "out0", "out1", "mlen", "mbeg", "mend", "W"):
            a[k][f] = p[f]
        a[k]["mats0"] = p["mats0"]
        a[k]["mats1"] = p["mats1"]
    return a
