"""oracle/collapse_oracle.c (the C restatement of HitCluster::collapseAndFilterHits, /root/reference/src/alignments.cpp:658-703)
pinned to the reference's own HitCluster -- every read through addOpenHit, then the reference's collapse
(oracle/ref_shim.cpp: ref_collapse_cluster; where oracle/_ref is built) -- and the product's host form
(sbgpu_collapse_pairs_host) checked against the oracle.  CPU only."""
import numpy as np
import pytest


def random_cluster(rng, n, base=100000, tie_free=True):
    """n read pairs: duplicates (same fragment several times), NH 1-4, single reads, spliced mates, one far outlier.
    tie_free: no two DIFFERENT fragments share (left end, right end) -- the reference's std::sort leaves their order
    to its implementation."""
    left, right, nh, seen = [], [], [], {}
    while len(left) < n:
        s = int(rng.integers(base, base + 600))
        lb = [(s, s + 74)]
        if rng.random() < 0.3:
            cut, gap = int(rng.integers(10, 60)), int(rng.choice([200, 350]))
            lb = [(s, s + cut - 1), (s + cut + gap, s + gap + 74)]
        if rng.random() < 0.1:
            rb = []
        else:
            ins = int(rng.choice([180, 200, 230, 260]))
            rs = lb[-1][1] + 1 + ins - 75
            rb = [(rs, rs + 74)]
            if rng.random() < 0.2:
                rb = [(rs, rs + 30), (rs + 31 + 150, rs + 74 + 150)]
        ends = (lb[0][0], (rb or lb)[-1][1])
        sig = (tuple(lb), tuple(rb))
        if tie_free and seen.get(ends, sig) != sig:
            continue
        seen[ends] = sig
        for _ in range(int(rng.choice([1, 1, 1, 2, 3]))):       # PCR duplicates
            left.append(lb)
            right.append(rb)
            nh.append(int(rng.choice([1, 1, 2, 3, 4])))
    if n > 20:   # a read whose span is far off the cluster's (filtered)
        left.append([(base + 100, base + 130), (base + 40000, base + 40043)])
        right.append([(base + 40300, base + 40374)])
        nh.append(1)
    perm = rng.permutation(len(left))
    return [left[i] for i in perm], [right[i] for i in perm], [nh[i] for i in perm]


def test_phi_known_values(oracle):
    assert abs(oracle.L.sbo_phi(0.0) - 0.5) < 1e-9
    assert abs(oracle.L.sbo_phi(3.0902) - 0.999) < 2e-6 and oracle.L.sbo_phi(-1.0) < 0.16


def test_collapse_oracle_small_known_cluster(oracle):
    left = [[(100, 174)], [(100, 174)], [(120, 194)], [(100, 174)]]
    right = [[(300, 374)], [(300, 374)], [], [(300, 340), (400, 433)]]
    up, um, cm, nf = oracle.collapse_cluster(left, right, [1, 2, 1, 1])
    # order: (100, 374) twice [equal: collapsed, masses 1 + 1/2], (100, 433), (120, 194)
    assert list(up) == [0, 3, 2] and nf == 0
    np.testing.assert_allclose(um, [1.5, 1.0, 1.0])
    assert cm == 3.5


def test_collapse_oracle_equals_reference_hitcluster(oracle, reflib):
    rng = np.random.default_rng(77)
    for trial in range(60):
        n = int(rng.integers(1, 400 if trial % 10 else 3000))
        left, right, nh = random_cluster(rng, n)
        up, um, cm, nf = oracle.collapse_cluster(left, right, nh)
        rp, rm, rcm, _ = reflib.collapse_cluster(left, right, nh)
        assert len(up) == len(rp)
        # which duplicate represents a unique hit is the sort's business: compare the fragments
        assert [(left[a], right[a]) for a in up] == [(left[b], right[b]) for b in rp]
        # a duplicate group's masses (1, 1/2, 1/3, 1/4 ...) are added in the sorted order, and std::sort leaves the order of
        # equal pairs to its implementation: the sums agree to the last bit or two, not bit for bit
        np.testing.assert_allclose(um, rm, rtol=4e-16, atol=0)
        assert abs(cm - rcm) <= 1e-12 * max(1.0, rcm)
        if n > 200:     # (the far read is an outlier of 5 sd only once the cluster has a few hundred reads)
            assert nf >= 1


def test_product_host_collapse_equals_oracle(oracle):
    """sbgpu_collapse_pairs_host (csrc/locus_bins.cpp, host code of libsbgpu.so) against the oracle: unique hits in
    order, float masses, cluster masses, filtered counts; several loci, shuffled input, ties included."""
    from strawberry_amd import exonbin as eb
    rng = np.random.default_rng(9)
    loc, mass, left, right, nhs = [], [], [], [], []
    n_loci = 7
    for l in range(n_loci):
        a, b, c = random_cluster(rng, int(rng.integers(0, 500)) if l != 3 else 0, base=100000 * (l + 1), tie_free=False) if l != 3 else ([], [], [])
        loc += [l] * len(a)
        left += a
        right += b
        nhs += c
    mass = [1.0 / k for k in nhs]
    perm = rng.permutation(len(loc))
    loc, mass, left, right, nhs = ([x[i] for i in perm] for x in (loc, mass, left, right, nhs))
    hits, cmass, info = eb.collapse_pairs(n_loci, loc, mass, left, right)
    at = 0
    filtered = 0
    for l in range(n_loci):
        idx = [i for i in range(len(loc)) if loc[i] == l]
        up, um, cm, nf = oracle.collapse_cluster([left[i] for i in idx], [right[i] for i in idx], [nhs[i] for i in idx])
        filtered += nf
        assert abs(cmass[l] - cm) < 1e-12
        for a, m in zip(up, um):
            f = eb.hit_features(left[idx[a]], right[idx[a]])
            if f is None:
                continue
            assert hits.hit_locus[at] == l
            s = slice(hits.feat_off[at], hits.feat_off[at + 1])
            assert (list(hits.feat_code[s]), list(hits.feat_left[s]), list(hits.feat_right[s])) == (f[0], f[1], f[2])
            assert hits.mass[at] == np.float32(m)
            at += 1
    assert at == hits.n_hits and info["filtered"] == filtered
