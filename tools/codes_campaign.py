"""CPU campaign: the compact traceback's scalar statement (orc_align_pairwise_codes_mode) against the literal backcal restatement
on N random pairs, 2-piece gaps by default (usage: python tools/codes_campaign.py [N] [seed] [1|2 pieces])."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import support as S  # noqa: E402
from test_oracle_codes import SCORINGS, SCORINGS2, codes_align  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    sc_list = SCORINGS if (len(sys.argv) > 3 and sys.argv[3] == "1") else SCORINGS2
    rng = np.random.default_rng(seed)
    same = bad = hand = both = 0
    for it in range(n):
        L = int(rng.choice([1, 5, 15, 16, 17, 40, 100, 300, 800]))
        T = rng.integers(0, 4, size=L).astype(np.uint8)
        Q = S.mutate(rng, T, float(rng.choice([0.0, 0.02, 0.1, 0.2, 0.4])))
        r = float(rng.choice([1.0, 1.0, 1.0, 0.8, 1.25, 2.0, 0.5, 0.3]))
        if r != 1.0:
            Lq = max(1, int(len(Q) * r))
            Q = Q[:Lq] if Lq <= len(Q) else np.concatenate([Q, rng.integers(0, 4, size=Lq - len(Q)).astype(np.uint8)])
        if len(Q) == 0:
            Q = np.array([1], np.uint8)
        bw = int(rng.choice([0, 16, 32, 48, 64, 128, 256]))
        sc = sc_list[int(rng.integers(len(sc_list)))]
        mode = int(rng.integers(3))
        if os.environ.get("CAMPAIGN_MODE"):
            mode = int(os.environ["CAMPAIGN_MODE"])
        res, cig, k = S.oracle_align(Q, T, mode, bw, *sc)
        cres, ccig, ck = codes_align(Q, T, bw, *sc, mode=mode)
        if k == S.ORC_ERR_TRACE:
            both += ck == S.ORC_ERR_TRACE
            bad += ck != S.ORC_ERR_TRACE
        elif ck == S.ORC_ERR_TRACE:
            hand += 1
        elif ck == k and np.array_equal(res, cres) and np.array_equal(cig, ccig):
            same += 1
        else:
            bad += 1
            print("DIFF", mode, L, len(Q), bw, sc, flush=True)
    print("pairs %d identical %d hand-overs %d both-nonterminating %d differences %d" % (n, same, hand, both, bad), flush=True)


if __name__ == "__main__":
    main()
