"""The committed golden vectors (tests/golden/*.npz, written by tests/golden/make_golden.py from the
oracle) against the oracle as it is now: any change of the restated algorithm shows up here."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden  # noqa: E402


def compare(expected_file, got):
    with np.load(expected_file) as z:
        keys = [k for k in z.files if k not in ("couplings", "temperatures")]
        assert set(keys) == set(got), sorted(set(keys) ^ set(got))
        for key in keys:
            np.testing.assert_array_equal(z[key], got[key], err_msg=key)


@pytest.mark.parametrize("name", sorted(make_golden.CASES))
def test_oracle_reproduces_golden_vectors(oracle, name):
    case = make_golden.CASES[name]
    path = os.path.join(make_golden.HERE, f"{name}.npz")
    with np.load(path) as z:
        J, temps = z["couplings"], z["temperatures"]
    np.testing.assert_array_equal(J, make_golden.couplings_of(case))  # the inputs are part of the fixture
    sim = oracle.OracleSimulation(list(case["shape"]), J, temps, case["n_rep"], None, case["seed"])
    compare(path, make_golden.run(sim, case))
