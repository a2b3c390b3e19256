import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import OracleLib, build
    build(with_ref=False)
    return OracleLib()


@pytest.fixture(scope="session")
def reflib():
    """The reference's own EmSolver (oracle/_ref); only where it has been built."""
    from oracle import RefLib, have_ref
    if not have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference; `make -C oracle ref`)")
    return RefLib()


def load_golden(name):
    from strawberry_amd.synth import LocusBatch
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    b = LocusBatch(z["row_off"], z["iso_off"], z["f_off"], z["count"], z["F"], z["length"], name)
    return b, z["ref_theta"], z["ref_flags"]


@pytest.fixture(scope="session")
def golden():
    return load_golden


def ref_flags_to_status(flags, oracle_status):
    """EmSolver exposes two bools; MAXITER is not observable through it, so an
    (init=1, run=1) golden matches both OK and MAXITER."""
    out = []
    for f, s in zip(flags, oracle_status):
        if not (f & 1):
            out.append(1)
        elif not (f & 2):
            out.append(2)
        else:
            out.append(int(s) if int(s) in (0, 3) else 0)
    return np.array(out, np.int32)


def rel_err(a, b, floor=1e-300):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), floor)
