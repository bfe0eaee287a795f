"""The HIP engine against the committed golden vectors (tests/golden/*.npz): inputs and expected
outputs only, no oracle involved at run time.  Bit-exact."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden  # noqa: E402
from test_golden import compare  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(make_golden.CASES))
def test_engine_reproduces_golden_vectors(name):
    from peapods_amd import IsingSimulation

    case = make_golden.CASES[name]
    path = os.path.join(make_golden.HERE, f"{name}.npz")
    with np.load(path) as z:
        J, temps = z["couplings"], z["temperatures"]
    sim = IsingSimulation(list(case["shape"]), J, temps, case["n_rep"], None, case["seed"])
    compare(path, make_golden.run(sim, case))
