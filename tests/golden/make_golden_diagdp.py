"""tests/golden/diagdp.npz: rows written by the REAL reference's anti-diagonal DP (oracle/_ref: ref_diagdp_fill calls
maxmat_dp_diag_rowcal_init / _prepare / maxmat_dp_diag_rowcal of /root/reference/bspoa.h) on the seeded windows of
tests/test_diagdp_cpu.py -- hashes of inputs and outputs plus the first 4 KiB of every output.  Run in the build container:
    python tests/golden/make_golden_diagdp.py"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import diag_support as D
from test_diagdp_cpu import CASES, _case

out = {}
for k in range(len(CASES)):
    planes, probs, nbytes = _case(k)
    rows = D.written_rows(D.ref_fill(planes, probs, nbytes), probs)
    out["planes_sha_%d" % k] = hashlib.sha256(planes.tobytes()).hexdigest()
    out["rows_sha_%d" % k] = hashlib.sha256(rows.tobytes()).hexdigest()
    out["rows_head_%d" % k] = rows[:4096]
np.savez_compressed(os.path.join(HERE, "diagdp.npz"), **out)
print("wrote diagdp.npz:", {k: (v if isinstance(v, str) else v.shape) for k, v in out.items() if k.startswith("rows_sha")})
