"""CPU: the C oracle against the golden vectors and the NumPy oracle."""
import numpy as np
import pytest

from oracle import oracle_c, oracle_np as onp
from tests.golden import cases


@pytest.mark.parametrize("case", cases.PROBER_CASES[:3], ids=lambda c: c["name"])
def test_c_prober_and_gate_match_reference(golden, case):
    states = [cases.synth_state(case["wseed"] + l, case["d"]) for l in range(case["L"])]
    x = cases.case_x(case)
    got = np.stack([oracle_c.prober_forward(s, x[l]) for l, s in enumerate(states)])
    np.testing.assert_allclose(got, golden[f"{case['name']}/logits"], atol=1e-5, rtol=0)
    for ab in (0, 2, 5):
        s, dec = oracle_c.gate(golden[f"{case['name']}/logits"], ab, 1.0)
        np.testing.assert_allclose(s, golden[f"{case['name']}/probsum_ab{ab}"], atol=1e-6, rtol=0)
        ref = golden[f"{case['name']}/decision_ab{ab}_th1.0"]
        margin = np.abs(s[:, 0] + np.float32(1.0) - s[:, 1])
        assert not ((dec != ref) & (margin > 1e-5)).any()


@pytest.mark.parametrize("metric", [onp.METRIC_L2, onp.METRIC_IP, onp.METRIC_COS])
def test_c_flat_search_equals_numpy(metric):
    X = onp.synth_rows(42, 0, 2000, 96)
    X[1500] = X[3]            # exact duplicate -> tie broken by id
    Q = onp.synth_rows(7, 0, 9, 96)
    xs = onp.normalize_rows(X) if metric == onp.METRIC_COS else X
    D0, I0 = onp.flat_search(xs, Q, 6, metric)
    D1, I1 = oracle_c.flat_search(xs, Q, 6, metric)
    assert np.array_equal(I0, I1)
    np.testing.assert_allclose(D0, D1, rtol=1e-6, atol=1e-6)
    D2, I2 = oracle_c.flat_search(xs[:4], Q, 6, metric)
    assert (I2[:, 4:] == -1).all()
