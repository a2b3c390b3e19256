"""CPU tests: the oracle (plain-C restatement) against the reference's golden
vectors and known-answer tests.  No GPU."""
import numpy as np
import pytest

from conftest import ref_flags_to_status, rel_err

# Known-answer tests captured from the reference's EmSolver during the survey
# ("Known-answer tests captured from the oracle", %.12g).
KATS = [
    ("toy", [100, 50, 30], [[.002, .001], [.003, 0], [0, .004]], 0, [136.1126370333, 43.8873629667]),
    ("denom_zero", [0, 5], [[.1, 0], [0, .1]], 2, [2.5, 2.5]),
    ("all_dropped", [3, 4], [[1e-5, 1e-6], [0, 1e-5]], 1, [3.5, 3.5]),
    ("zero_col", [10, 20], [[.2, .1, 0], [.05, .3, 0]], 0, [4.58515682744, 25.4148431726, 0]),
    ("row_dropped", [7, 10, 20], [[1e-6, 1e-6], [.2, .1], [.05, .3]], 0, [4.58515682744, 25.4148431726]),
    ("single_iso", [10, 20], [[.2], [.05]], 0, [30]),
    ("single_row", [9], [[.2, .1, .4]], 0, [2.57142857143, 1.28571428571, 5.14285714286]),
]


@pytest.mark.parametrize("name,n,F,status,theta", KATS, ids=[k[0] for k in KATS])
def test_oracle_known_answers(oracle, name, n, F, status, theta):
    th, st, it = oracle.em_locus(n, np.array(F, np.float64))
    assert st == status
    np.testing.assert_allclose(th, theta, rtol=2e-11, atol=1e-11)
    if status == 1:
        assert it == 0


@pytest.mark.parametrize("name", ["em_edge", "em_random_256", "em_c2_64", "em_c3_400", "em_c4_assembled"])
def test_oracle_matches_reference_goldens(oracle, golden, name):
    """theta from the restatement == theta from the reference's EmSolver (1e-12 rel,
    the agreement BASELINE.md asks for before the restatement is trusted)."""
    b, ref_theta, ref_flags = golden(name)
    theta, status, iters = oracle.em_batch(b.row_off, b.iso_off, b.f_off, b.count, b.F)
    np.testing.assert_array_equal(status, ref_flags_to_status(ref_flags, status))
    # absolute floor: theta below 1e-9 fragments is noise of the last iterations
    err = np.abs(theta - ref_theta) / np.maximum(np.abs(ref_theta), 1e-9)
    assert err.max() < 1e-10, (name, err.max(), int(err.argmax()))
    assert (iters[status == 1] == 0).all()
    assert (iters[status == 3] == 1000).all()
    assert (iters[status == 0] >= 1).all() and (iters[status == 0] <= 1000).all()


def test_oracle_vs_live_reference(oracle, reflib):
    """Where oracle/_ref is built: fresh random loci through both."""
    from strawberry_amd import synth
    b = synth.make_random(128, max_nrow=48, max_niso=10, seed=99)
    theta, status, _ = oracle.em_batch(b.row_off, b.iso_off, b.f_off, b.count, b.F)
    rtheta, rflags = reflib.em_batch(b.row_off, b.iso_off, b.f_off, b.count, b.F)
    np.testing.assert_array_equal(status, ref_flags_to_status(rflags, status))
    err = np.abs(theta - rtheta) / np.maximum(np.abs(rtheta), 1e-9)
    assert err.max() < 1e-10


def test_oracle_threads_equal_serial(oracle, golden):
    b, _, _ = golden("em_random_256")
    t1 = oracle.em_batch(b.row_off, b.iso_off, b.f_off, b.count, b.F, threads=1)
    t4 = oracle.em_batch(b.row_off, b.iso_off, b.f_off, b.count, b.F, threads=4)
    for x, y in zip(t1, t4):
        np.testing.assert_array_equal(x, y)


def test_em_preserves_mass(oracle, golden):
    """Size-independent property: after the first iteration sum(theta) = sum of the
    kept rows' counts."""
    b, _, _ = golden("em_c2_64")
    theta, status, iters = oracle.em_batch(b.row_off, b.iso_off, b.f_off, b.count, b.F)
    for l in range(b.n_loci):
        if status[l] in (0, 3) and iters[l] > 1:
            n, F = b.locus(l)
            kept = (F > 1e-5).any(axis=1)
            assert abs(theta[b.iso_off[l]:b.iso_off[l + 1]].sum() - n[kept].sum()) < 1e-6 * max(1, n.sum())


def test_abundance_and_tpm_oracle(oracle):
    theta = np.array([10.0, 30.0, 0.0, 60.0])
    length = np.array([1000, 2000, 500, 4000], np.int32)
    fpkm, frac, keep, s = oracle.abundance_locus(theta, length, 1_000_000)
    np.testing.assert_allclose(fpkm, theta * 1.0 * (1e3 / length), rtol=1e-15)
    np.testing.assert_allclose(frac, fpkm / fpkm.sum(), rtol=1e-15)
    assert list(keep) == [1, 1, 0, 1]  # Frac < 0.01 erased (estimate.cpp:346-355)
    fpkm2, frac2, keep2, _ = oracle.abundance_locus(theta, length, 1_000_000, min_isoform_frac=0.0)
    assert list(keep2) == [1, 1, 1, 1]  # -r: kMinIsoformFrac = 0 keeps everything
    tpm, tot = oracle.tpm(fpkm, keep)
    assert abs(tpm.sum() - 1e6) < 1e-6
    assert tpm[2] == 0.0
    # effective_len_norm: length - mean < 0 -> "NA" (estimate.cpp:317-324)
    f3, fr3, k3, _ = oracle.abundance_locus(theta, length, 1_000_000, effective_len_norm=True, insert_mean=600.0,
                                            min_isoform_frac=0.0)
    assert k3[2] == 2 and f3[2] == 0.0
    np.testing.assert_allclose(f3[0], 10.0 * 1.0 * (1e3 / 400.0), rtol=1e-15)
