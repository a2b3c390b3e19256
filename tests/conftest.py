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


REF_DIR = os.path.join(ROOT, "oracle", "_ref")
REF_PROGRAMS = ("libstrawberry_ref.so", "strawberry_ref", "strawberry_sbgpu", "strawberry_sbgpu_batched",
                "strawberry_sbgpu_chain", "strawberry_sbgpu_front", "strawberry_dump", "sam2bam")


def _ref_half_shipped():
    """libstrawberry_ref.so is there (the tree carries a built oracle/_ref) but some of its programs are not."""
    return os.path.exists(os.path.join(REF_DIR, "libstrawberry_ref.so")) and not all(
        os.path.exists(os.path.join(REF_DIR, f)) for f in REF_PROGRAMS)


# A tree that carries oracle/_ref must carry ALL of it: where the reference library is present at collection time the
# tests that need any of its files FAIL instead of skipping (a half-shipped _ref would otherwise lose tests quietly).
# A clean checkout (no _ref at all) still skips them, and says so in the header.
if os.path.exists(os.path.join(REF_DIR, "libstrawberry_ref.so")):
    os.environ.setdefault("SBGPU_REQUIRE_REF", "1")


def pytest_report_header(config):
    """Says up front whether the compiled reference (oracle/_ref: git-ignored, built only where /root/reference exists,
    carried to the GPU box with the tree) is there -- the tests that link or run it skip without it, and a run on a
    clean checkout should not lose them quietly.  SBGPU_REQUIRE_REF=1 turns those skips into failures."""
    have = [f for f in REF_PROGRAMS if os.path.exists(os.path.join(REF_DIR, f))]
    miss = [f for f in REF_PROGRAMS if f not in have]
    line = "oracle/_ref (the reference compiled from its own sources): %d of %d files present" % (len(have), len(REF_PROGRAMS))
    if miss:
        line += "; MISSING " + ", ".join(miss) + (" -> the tests that need them will FAIL (SBGPU_REQUIRE_REF=1: the reference library "
                                                  "is here, so the rest of oracle/_ref must be)" if os.environ.get("SBGPU_REQUIRE_REF") == "1" else
                                                  " -> the tests against the reference itself will SKIP (goldens and the oracle still run)")
    return [line]


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    skipped = [r for r in terminalreporter.stats.get("skipped", []) if "oracle/_ref" in str(getattr(r, "longrepr", ""))]
    if skipped:
        terminalreporter.write_line("NOTE: %d test(s) skipped because oracle/_ref is not built here (needs /root/reference; "
                                    "`make -C oracle ref`)" % len(skipped), yellow=True)


def need_ref(what="oracle/_ref"):
    """skip (or, with SBGPU_REQUIRE_REF=1, fail) a test that needs the compiled reference"""
    msg = "%s not built (oracle/_ref needs /root/reference; `make -C oracle ref`)" % what
    if os.environ.get("SBGPU_REQUIRE_REF") == "1":
        pytest.fail(msg)
    pytest.skip(msg)


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
        need_ref("oracle/_ref/libstrawberry_ref.so")
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
