"""Synthetic POA sweep programs for measurement (bench.py --workload poa) and tests.

A real program comes from the reference's graph (include/bsalign_poa_adapter.h flattens align_rd_bspoacore,
bspoa.h:2515-2618).  The reference is not available where the benchmark runs, so this module builds programs with the
same structure and the op mix measured on the reference's real programs (30 reads x 3 kbp, eps 0.10, default POA
parameters, read 20 of 30): 1.9 graph nodes and 3.2 row updates per read base, 0.42 merges per update, band steps
0/1/2/3+ = 26/55/16/3 %, 55 % of the tasks continue from the row the previous task produced.

Graph model: position i of the window has a main node M_i and, with probability p_alt, an alternative node A_i (a
mismatch / insertion branch).  Edges: M_{i-1}->M_i always; M_{i-1}->A_i; A_{i-1}->M_i; A_{i-1}->A_i with probability
p_aa; a skip edge M_{i-2}->M_i with probability p_skip (a deletion branch).  The first in-edge of a node writes the
node's row block, every further one goes through block 1 and is merged (bspoa.h:2585-2593).  Band offset of position i
is clamp(rmap[i] - bw/2, 0, slen - bw) with rmap advancing 0/1/2 read bases per position, as prepare_rd_align_bspoa assigns it from the read-to-consensus map (bspoa.h:2170-2176).
"""
import numpy as np

TASK_DTYPE = np.dtype([("op", np.uint32), ("src", np.uint32), ("dst", np.uint32), ("qoff_src", np.uint32), ("qoff_dst", np.uint32),
                       ("toff", np.uint32), ("query", np.uint32), ("base", np.uint8), ("prof", np.uint8), ("reserved", np.uint16)])
OP_UPDATE, OP_MERGE, OP_INIT, OP_SCORE_TAIL, OP_SCORE_END = 0, 1, 2, 3, 4


def make_program(seed, npos, bw, slen=None, overlap=True, p_alt=0.6, p_aa=0.5, p_skip=0.5):
    """-> (tasks[TASK_DTYPE] with query index 0, number of row blocks, number of updates, number of merges, read length)"""
    rng = np.random.default_rng(seed)
    alt = rng.random(npos) < p_alt
    aa = rng.random(npos) < p_aa
    skip = rng.random(npos) < p_skip
    base_m = rng.integers(0, 4, size=npos)
    base_a = (base_m + rng.integers(1, 4, size=npos)) & 3
    bonus = rng.integers(0, 2, size=2 * npos)
    # read position each graph position maps to: advances by 0 / 1 / 2 bases (insertion branch / match / deletion)
    rmap = np.cumsum(rng.choice([0, 1, 1, 1, 1, 1, 1, 2, 1, 0], size=npos))
    slen = slen or int(rmap[-1]) + 1
    rpos = np.clip(rmap - bw // 2, 0, max(slen - bw, 0))
    blk_m = np.zeros(npos, dtype=np.int64)
    blk_a = np.zeros(npos, dtype=np.int64)
    nxt = 3                                       # 0 scratch, 1 merge temp, 2 head
    for i in range(npos):
        blk_m[i] = nxt
        nxt += 1
        if alt[i]:
            blk_a[i] = nxt
            nxt += 1
    tasks = [(OP_INIT, 0, 2, 0, 0, 0, 0, 0, 0, 0)]
    nupd = nmrg = 0

    def edge(src_blk, src_rpos, src_base, dst_blk, i, base, bon, first):
        nonlocal nupd, nmrg
        prof = (2 if base == src_base else 0) + int(bon)
        tasks.append((OP_UPDATE, src_blk, dst_blk if first else 1, src_rpos, int(rpos[i]), i, 0, base, prof, 0))
        nupd += 1
        if not first:
            tasks.append((OP_MERGE, 1, dst_blk, 0, 0, 0, 0, 0, 0, 0))
            nmrg += 1

    for i in range(npos):
        pm = (blk_m[i - 1], int(rpos[i - 1]), int(base_m[i - 1])) if i else (2, 0, 4)
        # the alternative node first, so that the position ends with M_i's row in registers for the next position
        if alt[i]:
            edge(pm[0], pm[1], pm[2], int(blk_a[i]), i, int(base_a[i]), bonus[2 * i + 1], True)
            if i and alt[i - 1] and aa[i]:
                edge(int(blk_a[i - 1]), int(rpos[i - 1]), int(base_a[i - 1]), int(blk_a[i]), i, int(base_a[i]), bonus[2 * i + 1], False)
        # in-edges of M_i: the previous main node, then the previous alternative, then the skip edge
        edge(pm[0], pm[1], pm[2], int(blk_m[i]), i, int(base_m[i]), bonus[2 * i], True)
        if i and alt[i - 1]:
            edge(int(blk_a[i - 1]), int(rpos[i - 1]), int(base_a[i - 1]), int(blk_m[i]), i, int(base_m[i]), bonus[2 * i], False)
        if i >= 2 and skip[i]:
            edge(int(blk_m[i - 2]), int(rpos[i - 2]), int(base_m[i - 2]), int(blk_m[i]), i, int(base_m[i]), bonus[2 * i], False)
        if overlap and rpos[i] + bw >= slen and i % 16 == 0:
            tasks.append((OP_SCORE_END, int(blk_m[i]), 0, int(rpos[i]), 0, int(blk_m[i]), 0, 0, 0, 0))
    tasks.append((OP_SCORE_TAIL, int(blk_m[npos - 1]), 0, int(rpos[npos - 1]), 0, int(blk_m[npos - 1]), 0, 0, 0, 0))
    return np.array(tasks, dtype=TASK_DTYPE), nxt, nupd, nmrg, slen
