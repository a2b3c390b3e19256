"""CPU tests of the bin-weight restatement (the "next" row):
known-answer table from the reference's -f context TSV, and live comparison with
the reference's own ExonBin::effective_len / InsertSize::emp_dist_pdf where
oracle/_ref is built."""
import numpy as np
import pytest

# SURVEY.md bin-weight KAT: -i 200/20, read length 50, T1 = [1001-1300],[1501-1800],[2001-2400]
# (L=1000), T2 = [1001-1300],[2001-2400] (L=700); values from the reference's ctx.tsv.
T1, T2 = 1000, 700
KAT = [
    # (segment lengths under the isoform, implicit idx, iso_len, expected F)
    ([300], [], T1, 0.12554653416),
    ([300], [], T2, 0.200318325892),
    ([300, 300], [], T1, 0.249219236244),
    ([300, 400], [], T2, 0.399442931239),
    ([300], [], T1, 0.12554653416),            # [1501-1800] under T1
    ([300, 400], [], T1, 0.249219237416),
    ([400], [], T1, 0.250468456847),
    ([400], [], T2, 0.400238742868),
]


@pytest.mark.parametrize("segs,imp,L,expect", KAT)
def test_bin_weight_known_answers(oracle, segs, imp, L, expect):
    ins = oracle.make_insert(200.0, 20.0)
    w = oracle.bin_weight(segs, imp, L, 50, ins)
    assert abs(w - expect) / expect < 5e-12 * 10  # TSV prints 12 significant digits


def test_no_gap_and_gap_closed_forms(oracle):
    L = oracle.L
    # include/isoform.h:105-115
    assert L.sbo_no_gap_ef(100, 100, 0, 1) == 0          # fl < l_int + 2
    assert L.sbo_no_gap_ef(100, 100, 0, 201) == 0        # fl > total
    assert L.sbo_no_gap_ef(100, 100, 0, 2) == 1
    assert L.sbo_no_gap_ef(100, 100, 0, 101) == 100
    assert L.sbo_no_gap_ef(100, 100, 0, 200) == 1
    assert L.sbo_no_gap_ef(100, 50, 30, 100) == min(100, 69) + min(50, 69) - 69
    # include/isoform.h:117-129
    assert L.sbo_gap_ef(100, 100, 50, 50, 10) == 0       # 2rl+gap < ...? 110 >= 52 ok; start=max(50,139)=139,end=min(100,190) -> 0
    assert L.sbo_gap_ef(200, 200, 50, 50, 150) == max(0, min(200, 200 + 200 + 50 - 150 - 50) - max(50, 200 + 50 - 150 - 1))


def test_effective_len_vs_reference(oracle, reflib):
    rng = np.random.Generator(np.random.PCG64(5))
    n = 0
    for _ in range(3000):
        nseg = int(rng.integers(1, 9))
        segs = rng.integers(5, 300, nseg)
        if nseg <= 2:
            imp = []
        elif nseg == 3:
            imp = [1] if rng.random() < .5 else []
        elif nseg == 4:
            imp = [[], [1], [2], [1, 2]][int(rng.integers(0, 4))]
        else:
            imp = sorted(rng.choice(np.arange(1, nseg - 1), int(rng.integers(0, nseg - 1)), replace=False).tolist())
        rl = int(rng.integers(25, 101))
        inner = int(segs[1:-1].sum()) if nseg > 2 else 0
        lo = max(rl, inner) if nseg > 2 else rl
        hi = int(segs.sum())
        if lo > hi:
            continue
        for fl in rng.integers(lo, hi + 1, 4):
            a = oracle.effective_len(segs, imp, int(fl), rl)
            b = reflib.effective_len(segs, imp, int(fl), rl)
            assert a == b, (segs, imp, fl, rl, a, b)
            n += 1
    assert n > 5000


def test_insert_pdf_vs_reference(oracle, reflib):
    rng = np.random.Generator(np.random.PCG64(6))
    # Gaussian
    ref, _ = reflib.insert_pdf(250.0, 30.0, None, 1, 800)
    ins = oracle.make_insert(250.0, 30.0)
    mine = np.array([oracle.insert_pdf(ins, fl) for fl in range(1, 801)])
    np.testing.assert_allclose(mine, ref, rtol=1e-14, atol=0)
    # empirical with holes in the histogram (falls back to the Gaussian there)
    fl = np.rint(rng.normal(220, 25, 500)).astype(np.int32)
    ref, info = reflib.insert_pdf(0, 0, fl, 100, 400)
    ins = oracle.make_insert(info[0], info[1], fl)
    mine = np.array([oracle.insert_pdf(ins, x) for x in range(100, 401)])
    np.testing.assert_allclose(mine, ref, rtol=1e-14, atol=0)


def test_bin_weight_vs_reference(oracle, reflib):
    rng = np.random.Generator(np.random.PCG64(7))
    worst = 0.0
    for _ in range(300):
        nseg = int(rng.integers(1, 8))
        segs = rng.integers(20, 400, nseg)
        if nseg <= 2:
            imp = []
        elif nseg == 3:
            imp = [1] if rng.random() < .5 else []
        elif nseg == 4:
            imp = [[], [1], [2], [1, 2]][int(rng.integers(0, 4))]
        else:
            imp = sorted(rng.choice(np.arange(1, nseg - 1), int(rng.integers(0, nseg - 1)), replace=False).tolist())
        L = int(segs.sum() + rng.integers(0, 2000))
        rl = int(rng.integers(36, 101))
        a = oracle.bin_weight(segs, imp, L, rl, oracle.make_insert(230.0, 35.0))
        b = reflib.bin_weight(segs, imp, L, rl, 230.0, 35.0)
        if b != 0:
            worst = max(worst, abs(a - b) / abs(b))
        else:
            assert a == 0
    assert worst < 1e-12
