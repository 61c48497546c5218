"""The oracle's restatement of the anti-diagonal u8 DP of the MSA refinement against the real reference functions
(maxmat_dp_diag_rowcal_init / _prepare / maxmat_dp_diag_rowcal, compiled into oracle/_ref) and against the committed
fixture tests/golden/diagdp.npz (made by tests/golden/make_golden_diagdp.py from that build)."""
import hashlib
import os

import numpy as np
import pytest

import diag_support as D
import support as S

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "diagdp.npz")
CASES = [(1, 300, 6, 0.1, False), (2, 700, 9, 0.15, False), (4, 500, 5, 0.05, False), (2, 400, 40, 0.2, True), (1, 64, 3, 0.3, True)]


def _case(k):
    W, mlen, nreads, eps, sat = CASES[k]
    rng = np.random.default_rng(9000 + k)
    planes, probs = D.make_window(rng, mlen, nreads, W, eps, sat)
    return planes, probs, D.matrix_layout(probs)


@pytest.mark.skipif(not S.have_ref(), reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("k", range(len(CASES)))
def test_oracle_equals_the_reference_functions(k):
    planes, probs, nbytes = _case(k)
    a, b = D.oracle_fill(planes, probs, nbytes), D.ref_fill(planes, probs, nbytes)
    assert np.array_equal(D.written_rows(a, probs), D.written_rows(b, probs))


def test_oracle_reproduces_the_golden_fixture():
    g = np.load(GOLD)
    for k in range(len(CASES)):
        planes, probs, nbytes = _case(k)
        assert hashlib.sha256(planes.tobytes()).hexdigest() == str(g["planes_sha_%d" % k]), "the generator changed: regenerate the fixture"
        got = D.written_rows(D.oracle_fill(planes, probs, nbytes), probs)
        assert hashlib.sha256(got.tobytes()).hexdigest() == str(g["rows_sha_%d" % k])
        assert np.array_equal(got[:4096], g["rows_head_%d" % k])
