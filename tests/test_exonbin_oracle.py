"""A5 (exon-bin assignment) on the CPU: the oracle's restatement against the reference's answers
(tests/golden/exonbin_cases.npz, made by tools/make_exonbin_goldens.py from oracle/_ref; and live
against oracle/_ref where it is built), and the library's HOST bookkeeping (csrc/locus_bins.cpp:
segments, hit features, bins) against the reference binary's -f table (tests/golden/e2e_toy*).
No GPU work here: the kernel's words are supplied by the oracle."""
import os
import types

import numpy as np
import pytest

import e2e_util as U
import exonbin_util as XU
from strawberry_amd import exonbin as eb
from strawberry_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "exonbin_cases.npz")


@pytest.fixture(scope="module")
def cases():
    z = dict(np.load(GOLD))  # materialise: an NpzFile re-reads the archive on every access
    annot = types.SimpleNamespace(**{k: z[k] for k in ("iso_off", "exon_off", "exon_left", "exon_right")})
    annot.n_loci = len(annot.iso_off) - 1
    return z, annot


def annotation_from_arrays(z):
    loci = []
    for l in range(len(z["iso_off"]) - 1):
        isos = []
        for i in range(z["iso_off"][l], z["iso_off"][l + 1]):
            e = slice(z["exon_off"][i], z["exon_off"][i + 1])
            isos.append(list(zip(z["exon_left"][e].tolist(), z["exon_right"][e].tolist())))
        loci.append(isos)
    return eb.Annotation(loci), loci


def test_oracle_reproduces_reference_words(cases, oracle):
    z, _ = cases
    annot, _ = annotation_from_arrays(z)
    hits = eb.Hits.from_arrays(z["hit_locus"], z["feat_off"], z["feat_code"], z["feat_left"], z["feat_right"])
    assert annot.compat_words == z["compat"].shape[1] == 2 and annot.key_words == z["key"].shape[1] == 2
    compat, key = oracle.exonbin_batch(annot, hits)
    np.testing.assert_array_equal(compat, z["compat"])
    np.testing.assert_array_equal(key, z["key"])
    # the fixture is not trivial: compatible and incompatible hits, long hits, both words in use
    assert 0.5 < (z["compat"] != 0).any(1).mean() < 0.99
    assert np.diff(z["feat_off"]).max() > 8 and (z["compat"][:, 1] != 0).any() and (z["key"][:, 1] != 0).any()
    # and the per-call form agrees with the batch form
    c2, k2 = XU.oracle_words(oracle, annot, hits)
    np.testing.assert_array_equal(c2, compat)
    np.testing.assert_array_equal(k2, key)


def test_hit_features_reproduce_reference_contigs(cases):
    """sbgpu_hit_features == Contig::Contig(PairedHit) on every pair of the fixture, rejected pairs included."""
    z, _ = cases
    h = 0
    rejected = 0
    for p in range(len(z["pair_locus"])):
        lb = list(zip(z["left_l"][z["left_off"][p]:z["left_off"][p + 1]].tolist(),
                      z["left_r"][z["left_off"][p]:z["left_off"][p + 1]].tolist()))
        rb = list(zip(z["right_l"][z["right_off"][p]:z["right_off"][p + 1]].tolist(),
                      z["right_r"][z["right_off"][p]:z["right_off"][p + 1]].tolist()))
        f = eb.hit_features(lb, rb)
        if z["pair_n_feat"][p] == 0:
            assert f is None
            rejected += 1
            continue
        s = slice(z["feat_off"][h], z["feat_off"][h + 1])
        assert f == (z["feat_code"][s].tolist(), z["feat_left"][s].tolist(), z["feat_right"][s].tolist()), p
        h += 1
    assert h == len(z["hit_locus"]) and rejected > 20


def test_oracle_matches_reference_live(reflib, oracle):
    """Fresh random cases straight through the reference's code (only where oracle/_ref is built)."""
    loci = synth.make_gene_models(12, seed=31337)
    hit_locus, pairs = synth.make_fragments(loci, 25, seed=4, noise=0.4, mean=170.0, sd=40.0)
    annot = eb.Annotation(loci)
    n = 0
    for loc, (lb, rb) in zip(hit_locus, pairs):
        f = reflib.pairedhit_features(lb, rb)
        assert f == eb.hit_features(lb, rb)
        if f is None:
            continue
        for iso in loci[loc]:
            xl, xr = [a for a, _ in iso], [b for _, b in iso]
            assert oracle.is_compatible(f[0], f[1], f[2], xl, xr) == reflib.is_compatible(f[0], f[1], f[2], xl, xr)
        segs = annot.segments(loc)
        sl, sr = [a for a, _ in segs], [b for _, b in segs]
        np.testing.assert_array_equal(oracle.overlap_key(f[0], f[1], f[2], sl, sr), reflib.overlap_key(f[0], f[1], f[2], sl, sr))
        n += 1
    assert n > 200


def test_segments_host_is_the_disjoint_cut():
    loci = synth.make_gene_models(30, seed=5)
    annot = eb.Annotation(loci)
    for l, isos in enumerate(loci):
        assert annot.segments(l) == U.disjoint_segments([e for iso in isos for e in iso])
    # known answer: nested / abutting / identical exons
    a = eb.Annotation([[[(10, 20), (30, 40)], [(10, 20), (35, 50)], [(15, 18), (41, 45)], [(21, 25)]]])
    assert a.segments(0) == [(10, 14), (15, 18), (19, 20), (21, 25), (30, 34), (35, 40), (41, 45), (46, 50)]
    assert eb.Annotation([]).n_loci == 0


@pytest.mark.parametrize("which", ["E2E", "E2E_LONG", "E2E_MASS", "E2E_EMP", "E2E_SINGLE", "E2E_LONGREAD"])
def test_bins_reproduce_reference_context_table(which, oracle):
    """The reference binary's -f table lists every exon bin (its segments) with the number of
    fragments in it; hits -> (oracle words) -> sbgpu_bins_create must give the same bins and counts."""
    d = getattr(U, which)
    ordered, rows, _, _ = U.load(d)
    annot, hits, names, rejected = XU.e2e_inputs(d, ordered)
    assert rejected == open(os.path.join(d, "theta_log.txt")).read().count("not compatible")
    compat, key = oracle.exonbin_batch(annot, hits)
    bins = eb.LocusBins(annot, hits, compat, key)
    assert bins.n_bins == len(rows)
    assert hits.total_mapped == rows[0]["total_mapped"]
    from strawberry_amd.binseq import bin_segments     # the bins' segment lists as flat arrays == bin_coords
    off, sl, sr = bin_segments(bins)
    flat = [c for l in range(len(names)) for c in bins.bin_coords(l)]
    assert off.tolist() == np.concatenate([[0], np.cumsum([len(c) for c in flat])]).tolist()
    assert list(zip(sl.tolist(), sr.tolist())) == [s for c in flat for s in c]
    hits_in_bin = np.bincount(bins.hit_bin[bins.hit_bin >= 0], minlength=bins.n_bins)
    for l, g in enumerate(names):
        # the table's path_count is the number of unique hits of the bin; with unit masses and no
        # duplicates (the two plain toys) that is also the bin's count n_i
        ref = sorted((tuple(r["coords"]), r["count"]) for r in rows if r["gene"] == g)
        got = sorted(zip([tuple(c) for c in bins.bin_coords(l)], hits_in_bin[bins.row_off[l]:bins.row_off[l + 1]].tolist()))
        assert got == ref, g
        if which != "E2E_MASS":
            assert hits_in_bin[bins.row_off[l]:bins.row_off[l + 1]].tolist() == bins.count[bins.row_off[l]:bins.row_off[l + 1]].tolist()
        # a nonzero weight in the table is a (bin, isoform) pair of ours (the table prints, per bin, the
        # weights of the isoforms its LAST fragment is compatible with, alignments.cpp:1556-1563)
        by_coords = {tuple(c): b for b, c in zip(range(bins.row_off[l], bins.row_off[l + 1]), bins.bin_coords(l))}
        for r in rows:
            if r["gene"] == g:
                word = int(bins.bin_compat[by_coords[tuple(r["coords"])], 0])
                assert all((word >> j) & 1 for j, f in enumerate(r["F"]) if f != 0.0)
    assert bins.n_pairs >= sum(1 for r in rows for f in r["F"] if f != 0.0)
    # the batch arrays line up
    assert bins.f_off[-1] == bins.n_elem == int((np.diff(bins.row_off) * np.diff(bins.iso_off)).sum())
    assert bins.pair_out_index.max() < bins.n_elem and len(np.unique(bins.pair_out_index)) == bins.n_pairs
    assert (bins.hit_bin >= 0).sum() == bins.n_hits_used


def test_fractional_masses_reproduce_reference_theta(oracle):
    """PCR duplicates and multi-mapped pairs (tests/golden/e2e_toy_mass): hit masses are sums of 1, 1/2
    and 1/3; the bin counts n_i = (int) float-sum of them (ExonBin::read_count) feed the EM, whose theta
    the reference logged.  Our bins + the oracle's weights + the oracle's EM must land on that theta."""
    d = U.E2E_MASS
    ordered, rows, gtf, theta_log = U.load(d)
    annot, hits, names, _ = XU.e2e_inputs(d, ordered)
    assert len(np.unique(hits.mass)) > 10 and np.abs(hits.mass - np.rint(hits.mass)).max() > 0.3
    compat, key = oracle.exonbin_batch(annot, hits)
    bins = eb.LocusBins(annot, hits, compat, key)
    # counts differ from the plain number of hits in most bins -- the masses matter
    hits_in_bin = np.bincount(bins.hit_bin[bins.hit_bin >= 0], minlength=bins.n_bins)
    assert (hits_in_bin != bins.count).mean() > 0.5
    ins = oracle.make_insert(250.0, 30.0)
    F = np.zeros(bins.n_elem)
    for p in range(bins.n_pairs):
        s = slice(bins.pair_seg_off[p], bins.pair_seg_off[p + 1])
        imp = [k for k in range(32) if (int(bins.pair_implicit_mask[p]) >> k) & 1]
        F[bins.pair_out_index[p]] = oracle.bin_weight(bins.pair_seg_lens[s], imp, int(bins.pair_iso_len[p]), 75, ins)
    theta, status, iters = oracle.em_batch(bins.row_off, bins.iso_off, bins.f_off, bins.count, F)
    for l, ref_theta in enumerate(theta_log):
        th = theta[bins.iso_off[l]:bins.iso_off[l + 1]]
        assert np.abs(th - np.array(ref_theta)).max() < 1e-6, (names[l], th, ref_theta)
    # and FPKM through the oracle's epilogue with the reference's mapped-read total
    k = 0
    for l, g in enumerate(names):
        th = theta[bins.iso_off[l]:bins.iso_off[l + 1]]
        fpkm, frac, keep, _ = oracle.abundance_locus(th, bins.iso_len[bins.iso_off[l]:bins.iso_off[l + 1]],
                                                     hits.total_mapped, min_isoform_frac=0.0)
        for (t, _), f, fr in zip(ordered[g], fpkm, frac):
            assert abs(f - float(gtf[t][0])) <= 1e-6 * max(1.0, f), t
            assert abs(fr - float(gtf[t][1])) < 2e-6, t


@pytest.mark.parametrize("which", ["E2E", "E2E_MASS", "E2E_SINGLE", "E2E_LONGREAD"])
def test_collapse_pairs_gives_the_unique_hits(which):
    """sbgpu_collapse_pairs_host on every sequenced copy (shuffled) == the unique hits the other tests build by
    hand: same order, features, float masses, and the reference's mapped-read total (the -f table's column 2)."""
    d = getattr(U, which)
    ordered, rows, _, _ = U.load(d)
    annot, hits, names, rejected = XU.e2e_inputs(d, ordered)
    copies = XU.load_read_copies(d)
    rng = np.random.default_rng(5)
    copies = [copies[i] for i in rng.permutation(len(copies))]
    got, cluster_mass, info = eb.collapse_pairs(len(names), [c[0] for c in copies], [c[3] for c in copies],
                                                [c[1] for c in copies], [c[2] for c in copies])
    assert info["total_mapped"] == rows[0]["total_mapped"] == hits.total_mapped
    assert info["rejected"] == rejected and info["filtered"] == 0
    for a in ("hit_locus", "feat_off", "feat_code", "feat_left", "feat_right", "mass"):
        np.testing.assert_array_equal(getattr(got, a), getattr(hits, a), err_msg=a)
    if which == "E2E_MASS":
        assert len(copies) > got.n_hits * 1.3 and len(np.unique(got.mass)) > 10


def test_collapse_pairs_span_filter_and_equality():
    """The span filter (a mate far longer than the cluster's mates is skipped, its mass not counted) and
    what "equal" means when collapsing (both mates: same start and same blocks)."""
    left = [[(1000 + 3 * k, 1074 + 3 * k)] for k in range(200)]
    right = [[(1300 + 3 * k, 1374 + 3 * k)] for k in range(200)]
    left.append([(1500, 1574)]); right.append([(1700, 1710), (30000, 30063)])      # a mate spanning 28 kb
    left.append([(1003, 1077)]); right.append([(1303, 1377)])                      # a second copy of pair 1
    left.append([(1003, 1077)]); right.append([(1303, 1340), (1400, 1436)])        # same ends... no: spliced, not equal
    hits, cm, info = eb.collapse_pairs(1, [0] * 203, [1.0] * 201 + [0.5, 1.0], left, right)
    assert info["filtered"] == 1 and info["rejected"] == 0
    assert hits.n_hits == 201 and abs(cm[0] - 201.5) < 1e-12 and info["total_mapped"] == 201
    k = int(np.nonzero(hits.feat_left[hits.feat_off[:-1]] == 1003)[0][0])
    assert hits.mass[k] == 1.5 or hits.mass[k + 1] == 1.5   # the copy was collapsed into its twin
    # single reads (no right mate) and a pair whose mates abut (rejected, but counted in the cluster mass)
    hits, cm, info = eb.collapse_pairs(2, [0, 0, 1], [1.0, 1.0, 1.0], [[(10, 84)], [(10, 84)], [(500, 574)]],
                                       [[], [], [(575, 649)]])
    assert hits.n_hits == 1 and hits.mass[0] == 2.0 and info["rejected"] == 1 and cm.tolist() == [2.0, 1.0]


def test_bins_bookkeeping_rules():
    """set_maps / read_count details: first-appearance order, duplicate fragments count once,
    float masses truncate, hits without a compatible isoform are dropped."""
    annot = eb.Annotation([[[(100, 199), (300, 399)], [(100, 199), (500, 599)]]])
    assert annot.segments(0) == [(100, 199), (300, 399), (500, 599)]
    m = lambda l, r: ([0], [l], [r])  # noqa: E731
    feats = [m(300, 350), m(110, 150), m(300, 350), m(120, 160), m(510, 520), m(250, 260)]
    hits = eb.Hits([0] * 6, feats, mass=[1.5, 1.0, 7.0, 0.75, 2.0, 9.0])
    compat = np.array([[1], [3], [1], [3], [2], [0]], np.uint32)
    key = np.array([[2], [1], [2], [1], [4], [0]], np.uint32)
    b = eb.LocusBins(annot, hits, compat, key)
    assert b.bin_coords(0) == [[(300, 399)], [(100, 199)], [(500, 599)]]  # order of first appearance
    assert b.count.tolist() == [1, 1, 2]   # int(1.5) (duplicate ignored), int(1.0 + 0.75), int(2.0)
    assert b.hit_bin.tolist() == [0, 1, 0, 1, 2, -1]
    assert b.bin_compat[:, 0].tolist() == [1, 3, 2]
    assert b.n_pairs == 4 and b.pair_out_index.tolist() == [0, 2, 3, 5]
    assert b.pair_iso_len.tolist() == [200, 200, 200, 200]
    assert b.pair_seg_lens.tolist() == [100, 100, 100, 100] and b.pair_implicit_mask.tolist() == [0, 0, 0, 0]
    # a fragment spanning exon 1 and 3 of isoform A..C through a mate gap: segment 2 is implicit
    annot = eb.Annotation([[[(100, 199), (300, 399), (500, 599)]]])
    hits = eb.Hits([0], [([0, 2, 0], [150, 200, 520], [199, 519, 560])])
    b = eb.LocusBins(annot, hits, np.array([[1]], np.uint32), np.array([[5]], np.uint32))
    assert b.pair_seg_lens.tolist() == [100, 100, 100] and b.pair_implicit_mask.tolist() == [2]
    # word counts that do not cover the locus are refused
    from strawberry_amd._lib import SbgpuError
    wide = eb.Annotation([[[(k * 100, k * 100 + 50)] for k in range(1, 40)]])
    with pytest.raises(SbgpuError):
        eb.LocusBins(wide, eb.Hits([0], [m(100, 120)]), np.zeros((1, 1), np.uint32), np.zeros((1, 2), np.uint32))


def test_empirical_insert_size_reproduces_reference_weights(oracle):
    """No -i (tests/golden/e2e_toy_emp): the reference builds the empirical insert-size distribution from
    every unique hit that fits exactly one transcript (fragLenDist) and the bin weights use it.
    sbgpu_frag_lens_host + InsertSize(frag_lens) + the oracle's weight model must reproduce every nonzero
    weight of the reference's -f table (12 digits) and, through the oracle's EM, the logged theta."""
    from strawberry_amd.binweight import InsertSize
    d = U.E2E_EMP
    ordered, rows, gtf, theta_log = U.load(d)
    annot, hits, names, _ = XU.e2e_inputs(d, ordered)
    compat, key = oracle.exonbin_batch(annot, hits)
    fl = eb.frag_lens(annot, hits, compat)
    popc = np.array([bin(int(w)).count("1") for w in compat[:, 0]])
    assert len(fl) == (popc == 1).sum() > 1000
    ins = InsertSize.from_frag_lens(fl)
    assert ins.use_emp and ins.start_offset == fl.min() and ins.end_offset == fl.max()
    oi = oracle.make_insert(ins.mean, ins.sd, frag_lens=fl)
    bins = eb.LocusBins(annot, hits, compat, key)
    F = np.zeros(bins.n_elem)
    for p in range(bins.n_pairs):
        s = slice(bins.pair_seg_off[p], bins.pair_seg_off[p + 1])
        imp = [k for k in range(32) if (int(bins.pair_implicit_mask[p]) >> k) & 1]
        F[bins.pair_out_index[p]] = oracle.bin_weight(bins.pair_seg_lens[s], imp, int(bins.pair_iso_len[p]), 75, oi)
    n = 0
    for l, g in enumerate(names):
        coords = [tuple(c) for c in bins.bin_coords(l)]
        niso = int(bins.iso_off[l + 1] - bins.iso_off[l])
        Fl = F[bins.f_off[l]:bins.f_off[l + 1]].reshape(len(coords), niso)
        for r in rows:
            if r["gene"] == g:
                b = coords.index(tuple(r["coords"]))
                for j, f in enumerate(r["F"]):
                    if f != 0.0:
                        assert abs(Fl[b, j] - f) <= 5e-11 * f, (g, r["coords"], j, Fl[b, j], f)
                        n += 1
    assert n > 150
    theta, status, iters = oracle.em_batch(bins.row_off, bins.iso_off, bins.f_off, bins.count, F)
    for l, ref_theta in enumerate(theta_log):
        th = theta[bins.iso_off[l]:bins.iso_off[l + 1]]
        assert np.abs(th - np.array(ref_theta)).max() < 1e-6, (names[l], th, ref_theta)
