"""Mate pairing (HitCluster::addOpenHit + addHit, /root/reference/src/alignments.cpp:423-655), CPU:
  * oracle/matepair_oracle.c pinned to the reference's own HitCluster (oracle/ref_shim.cpp: ref_cluster_from_records,
    where oracle/_ref is built): records -> [reference: addOpenHit ... collapseAndFilterHits] must give the unique
    hits of records -> [oracle pairing] -> [oracle collapse];
  * the product's host form (sbgpu_pair_mates_host) against the oracle, pair by pair."""
import numpy as np
import pytest

import matepair_util as MU


def uniq_via_oracle(oracle, recs):
    ids, blocks, ppos, flags, nh = MU.arrays(recs)
    lr, rr, m, cnt = oracle.pair_mates(ids, blocks, ppos, flags, nh)
    left = [blocks[i] if i >= 0 else [] for i in lr]
    right = [blocks[i] if i >= 0 else [] for i in rr]
    # the collapse oracle takes one NH per pair: the tests' pairs have equal NH on both mates
    pnh = [nh[i if i >= 0 else j] for i, j in zip(lr, rr)]
    up, um, cm, nf = oracle.collapse_cluster(left, right, pnh)
    return [(left[a], right[a]) for a in up], um, cm, len(lr), cnt


def test_oracle_pairing_known_cluster(oracle):
    recs = [
        {"id": 5, "blocks": [(100, 174)], "ppos": 300, "flags": 4, "nh": 1},          # left mate waits
        {"id": 9, "blocks": [(120, 194)], "ppos": 0, "flags": 1, "nh": 2},            # single read, reverse: a right mate
        {"id": 5, "blocks": [(300, 374)], "ppos": 100, "flags": 5, "nh": 1},          # completes pair (0, 2)
        {"id": 7, "blocks": [(310, 384)], "ppos": 150, "flags": 1, "nh": 1},          # its mate never came: waits as a right mate
        {"id": 8, "blocks": [(400, 474)], "ppos": 400, "flags": 0, "nh": 1},          # partner at its own start: refused
    ]
    ids, blocks, ppos, flags, nh = MU.arrays(recs)
    lr, rr, m, cnt = oracle.pair_mates(ids, blocks, ppos, flags, nh)
    assert list(lr) == [-1, 0] and list(rr) == [1, 2]          # completion order: the single read first
    np.testing.assert_allclose(m, [0.5, 1.0])
    assert cnt == {"complete": 1, "single": 1, "refused": 1, "orphan": 1}


def test_oracle_pairing_equals_reference_hitcluster(oracle, reflib):
    rng = np.random.default_rng(2024)
    for trial in range(40):
        recs = MU.random_cluster(rng, int(rng.integers(1, 300)))
        ids, blocks, ppos, flags, nh = MU.arrays(recs)
        rl, rr, rm, rcm, n_hits = reflib.cluster_from_records(ids, blocks, ppos, flags, nh)
        want = [(blocks[i] if i >= 0 else [], blocks[j] if j >= 0 else []) for i, j in zip(rl, rr)]
        got, um, cm, n_pairs, cnt = uniq_via_oracle(oracle, recs)
        assert n_pairs == n_hits                                   # pairs + single reads before the collapse
        assert got == want
        np.testing.assert_allclose(um, rm, rtol=4e-16, atol=0)     # (a duplicate group's sum follows std::sort's tie order)
        assert abs(cm - rcm) <= 1e-12 * max(1.0, rcm)


def test_product_host_pairing_equals_oracle(oracle):
    from strawberry_amd import exonbin as eb
    rng = np.random.default_rng(7)
    n_loci = 9
    clusters = [MU.random_cluster(rng, int(rng.integers(0, 250)) if l != 4 else 0, base=200000 * (l + 1)) for l in range(n_loci)]
    loc = [l for l, c in enumerate(clusters) for _ in c]
    flat = [r for c in clusters for r in c]
    reads = eb.Reads(loc, *MU.arrays(flat))
    got = eb.pair_mates(n_loci, reads)
    at = 0
    tot = {"complete": 0, "single": 0, "refused": 0, "orphan": 0}
    for l, c in enumerate(clusters):
        ids, blocks, ppos, flags, nh = MU.arrays(c)
        lr, rr, m, cnt = oracle.pair_mates(ids, blocks, ppos, flags, nh)
        for k in tot:
            tot[k] += cnt[k]
        assert got["pair_off"][l + 1] - got["pair_off"][l] == len(lr)
        for i, j, mm in zip(lr, rr, m):
            for side, rec in (("left", i), ("right", j)):
                s = slice(int(got[side + "_off"][at]), int(got[side + "_off"][at + 1]))
                code, fl, fr = (x[s] for x in got[side])
                want = eb.mate_features(blocks[rec]) if rec >= 0 else ([], [], [])
                assert (list(code), list(fl), list(fr)) == tuple(list(x) for x in want)
            assert got["mass"][at] == mm
            at += 1
    assert at == got["info"]["pairs"]
    assert {k: got["info"][k] for k in tot} == tot


def stream_case(rng, n_clusters, n_reads, refs=2):
    """Genes (some overlapping, some on another reference, either strand) and position-sorted records around them."""
    c = sorted((int(rng.integers(0, refs)), int(rng.integers(1000, 60000))) for _ in range(n_clusters))
    c_ref = [x[0] for x in c]
    c_left = [x[1] for x in c]
    c_right = [l + int(rng.integers(200, 6000)) for l in c_left]
    c_strand = [int(rng.choice([1, 2, 1, 2, 0])) for _ in c]
    r = sorted((int(rng.integers(0, refs + 1)), int(rng.integers(1, 70000))) for _ in range(n_reads))
    r_ref = [x[0] for x in r]
    r_left = [x[1] for x in r]
    r_right = [l + int(rng.choice([74, 74, 300, 2500])) for l in r_left]
    r_xs = [int(rng.choice([1, 2, 0, 0])) for _ in r]
    return c_ref, c_left, c_right, c_strand, r_ref, r_left, r_right, r_xs


def test_oracle_read_stream_known_case(oracle):
    # two overlapping genes on one reference: a record is offered to the first cluster whose end it does not lie behind
    got, off = oracle.assign_reads([0, 0], [100, 300], [500, 900], [1, 1],
                                   [0, 0, 0, 0, 0, 0], [10, 90, 350, 501, 700, 950], [84, 164, 424, 575, 774, 1024], [0, 1, 2, 0, 1, 0])
    #          before gene 1 | overlaps 1 | strand - in a + gene | behind gene 1: gene 2 | gene 2 | behind everything
    assert list(got) == [-1, 0, -1, 1, 1, -1] and list(off) == [0, 3, 5]


def test_product_host_read_stream_equals_oracle(oracle):
    from strawberry_amd import exonbin as eb
    rng = np.random.default_rng(3)
    for trial in range(30):
        case = stream_case(rng, int(rng.integers(0, 40)), int(rng.integers(0, 3000)))
        want, woff = oracle.assign_reads(*case)
        flags = [x << 2 for x in case[7]]
        got, off, fl = eb.assign_reads(*case[:7], flags)
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(off, woff)
        np.testing.assert_array_equal((fl & 16) != 0, want < 0)
