"""A5 on the GPU: the exon-bin kernel (sbgpu_exonbin_host / _device) against the reference's
answers (tests/golden/exonbin_cases.npz), against the oracle on larger seeded inputs, and the
whole hits -> abundances chain against the reference binary's outputs (tests/golden/e2e_toy*).
Integer work: every comparison is bit-exact."""
import os

import numpy as np
import pytest

import e2e_util as U
import exonbin_util as XU
from test_exonbin_oracle import GOLD, annotation_from_arrays

pytestmark = pytest.mark.gpu

RL, MEAN, SD = 75, 250.0, 30.0


@pytest.fixture(scope="module")
def ctx():
    from strawberry_amd import em
    return em.default_context(0)


def test_kernel_reproduces_reference_words(ctx):
    from strawberry_amd import exonbin as eb
    z = dict(np.load(GOLD))
    annot, _ = annotation_from_arrays(z)
    hits = eb.Hits.from_arrays(z["hit_locus"], z["feat_off"], z["feat_code"], z["feat_left"], z["feat_right"])
    compat, key = eb.compat_and_keys(annot, hits, ctx)
    np.testing.assert_array_equal(compat, z["compat"])
    np.testing.assert_array_equal(key, z["key"])


def synth_case(n_loci, per_locus, seed, **kw):
    from strawberry_amd import exonbin as eb
    from strawberry_amd import synth
    loci = synth.make_gene_models(n_loci, seed=seed, **{k: v for k, v in kw.items() if k.startswith(("max_", "ex_", "in_"))})
    hl, pairs = synth.make_fragments(loci, per_locus, seed=seed + 1, **{k: v for k, v in kw.items() if not k.startswith(("max_", "ex_", "in_"))})
    feats, loc = [], []
    for l, (lb, rb) in zip(hl, pairs):
        f = eb.hit_features(lb, rb)
        if f is not None:
            feats.append(f)
            loc.append(l)
    return loci, eb.Annotation(loci), eb.Hits(loc, feats)


@pytest.mark.parametrize("name,args", [
    ("typical", dict(n_loci=300, per_locus=150, seed=1)),
    ("noisy_overlapping_mates", dict(n_loci=200, per_locus=100, seed=2, noise=0.5, mean=150.0, sd=15.0)),
    ("many_blocks", dict(n_loci=60, per_locus=200, seed=3, max_exons=40, ex_lo=15, ex_hi=50, read_len=150, mean=400.0, sd=50.0)),
    ("wide_loci", dict(n_loci=12, per_locus=300, seed=4, max_exons=60, max_isoforms=50, ex_lo=70, ex_hi=200)),
    ("single_exon_genes", dict(n_loci=100, per_locus=50, seed=5, max_exons=1, max_isoforms=1, ex_lo=300, ex_hi=900)),
])
def test_kernel_matches_oracle(ctx, oracle, name, args):
    from strawberry_amd import exonbin as eb
    loci, annot, hits = synth_case(**args)
    compat, key = eb.compat_and_keys(annot, hits, ctx)
    o_compat, o_key = oracle.exonbin_batch(annot, hits)
    np.testing.assert_array_equal(compat, o_compat)
    np.testing.assert_array_equal(key, o_key)
    assert hits.n_hits > 1000 and (compat != 0).any()
    if name == "many_blocks":
        assert np.diff(hits.feat_off).max() > 8      # the from-memory path of the kernel
    if name == "wide_loci":
        assert annot.compat_words > 1 or annot.key_words > 1


def test_kernel_edge_cases(ctx, oracle):
    from strawberry_amd import exonbin as eb
    from strawberry_amd._lib import SbgpuError
    annot = eb.Annotation([[[(100, 199), (300, 399)], [(100, 199), (500, 599)], []],
                           [[(1000, 1099)]]])
    m = lambda l, r: ([0], [l], [r])  # noqa: E731
    feats = [m(100, 199),                       # exactly an exon
             m(99, 150),                        # one base outside
             ([0, 1, 0], [150, 200, 300], [199, 299, 350]),   # junction of isoform 0 only
             ([0, 1, 0], [150, 200, 500], [199, 499, 550]),   # junction of isoform 1 only
             ([0, 2, 0], [150, 200, 500], [199, 499, 550]),   # same span through a mate gap
             ([0, 1, 0], [150, 200, 301], [199, 300, 350]),   # intron off by one
             ([], [], []),                      # a hit without features
             m(350, 520),                       # block over two segments of different isoforms
             m(1000, 1099)]
    hits = eb.Hits([0, 0, 0, 0, 0, 0, 0, 0, 1], feats)
    compat, key = eb.compat_and_keys(annot, hits, ctx)
    assert compat[:, 0].tolist() == [3, 0, 1, 2, 2, 0, 0, 0, 1]
    assert key[:, 0].tolist() == [1, 1, 3, 5, 5, 3, 0, 6, 1]
    o = oracle.exonbin_batch(annot, hits)
    np.testing.assert_array_equal(compat, o[0])
    np.testing.assert_array_equal(key, o[1])
    # no hits: nothing to do; too few words: refused
    eb.compat_and_keys(annot, eb.Hits([], []), ctx)
    wide = eb.Annotation([[[(k * 100, k * 100 + 50)] for k in range(1, 40)]])
    with pytest.raises(SbgpuError):
        eb.compat_and_keys(wide, eb.Hits([0], [m(100, 120)]), ctx, compat_words=1)
    with pytest.raises(SbgpuError):
        eb.compat_and_keys(annot, eb.Hits([7], [m(100, 120)]), ctx)   # hit_locus out of range


def test_irregular_feature_lists(ctx, oracle):
    """Feature lists that are not MATCH (connector MATCH)*: the kernel's per-lane walk must take them
    and agree with the oracle bit for bit, mixed into waves of regular hits."""
    from strawberry_amd import exonbin as eb
    from strawberry_amd import synth
    loci = synth.make_gene_models(20, seed=8)
    hl, pairs = synth.make_fragments(loci, 120, seed=9)
    rng = np.random.default_rng(10)
    feats, loc = [], []
    n_irregular = 0
    for l, (lb, rb) in zip(hl, pairs):
        f = eb.hit_features(lb, rb)
        if f is None:
            continue
        c, le, ri = [list(x) for x in f]
        u = rng.random()
        if u < 0.10 and len(c) >= 3:        # drop a connector: two MATCH blocks in a row
            k = 1 + 2 * int(rng.integers(0, len(c) // 2))
            del c[k], le[k], ri[k]
            n_irregular += 1
        elif u < 0.20 and len(c) >= 3:      # double connector: GAP then INTRON
            k = 1 + 2 * int(rng.integers(0, len(c) // 2))
            mid = (le[k] + ri[k]) // 2
            c[k:k + 1] = [2, 1]
            le[k:k + 1] = [le[k], mid + 1]
            ri[k:k + 1] = [mid, ri[k]]
            n_irregular += 1
        elif u < 0.25:                      # a connector first
            c, le, ri = [1] + c, [le[0] - 50] + le, [le[0] - 1] + ri
            n_irregular += 1
        elif u < 0.30 and len(c) >= 3:      # an INTRON turned MATCH
            c[1] = 0
            n_irregular += 1
        feats.append((c, le, ri))
        loc.append(l)
    annot, hits = eb.Annotation(loci), eb.Hits(loc, feats)
    assert n_irregular > 300
    compat, key = eb.compat_and_keys(annot, hits, ctx)
    o_compat, o_key = oracle.exonbin_batch(annot, hits)
    np.testing.assert_array_equal(compat, o_compat)
    np.testing.assert_array_equal(key, o_key)


def test_segment_basis_form_on_coinciding_boundaries(ctx, oracle):
    """exonbin_kernel decides compatibility in the SEGMENT basis (bit masks per isoform, exonbin_device.h) for loci of up
    to 64 segments.  Its equivalence with Contig::is_compatible rests on how exons, segments and blocks meet at their
    ends, so everything here lives on a coarse grid: exon ends on multiples of 10 -- abutting exons inside one isoform,
    alternative 5' / 3' ends, retained introns, exons inside other isoforms' exons -- and read blocks on multiples of 5:
    blocks that end exactly with an exon, stick out by half a cell, span abutting exons, sit in introns; INTRON
    connectors that match an isoform's intron, match another isoform's, or nothing; connectors that do not touch their
    blocks, and blocks out of order (the per-lane walk takes those).  Plus a locus of more than 64 segments (the exon
    walk).  Words equal the oracle's, bit for bit."""
    from strawberry_amd import exonbin as eb
    rng = np.random.default_rng(2024)
    loci = []
    for l in range(60):
        base = 1000 * (l + 1)
        n_cells = int(rng.integers(6, 40)) if l else 90          # locus 0: many segments
        isoforms = []
        for _ in range(int(rng.integers(1, 9))):
            exons, pos = [], int(rng.integers(0, 4))
            while pos < n_cells - 1:
                length = int(rng.integers(1, 5))
                end = min(pos + length, n_cells)
                exons.append((base + 10 * pos, base + 10 * end - 1))
                pos = end + int(rng.choice([0, 0, 1, 2, 3]))     # 0: the next exon abuts this one
            if l == 0:
                exons = [(base + 10 * k, base + 10 * k + 6) for k in range(0, 90, 1 + int(rng.integers(0, 2)))]
            if exons:
                isoforms.append(exons)
        if l % 7 == 3:
            isoforms.append([])                                   # an isoform without exons
        loci.append(isoforms)
    annot = eb.Annotation(loci)
    assert annot.key_words >= 3                                    # locus 0 is beyond the mask form
    nseg = np.diff(annot.seg_off)
    assert (nseg[1:] <= 64).all() and nseg[0] > 64
    loc, feats = [], []
    for l, isoforms in enumerate(loci):
        base = 1000 * (l + 1)
        span = max((e[-1][1] for e in isoforms if e), default=base + 50) - base
        for _ in range(400):
            nb = int(rng.integers(1, 5))
            starts = np.sort(rng.integers(0, max(2, span // 5), nb * 2)) * 5 + base
            c, le, ri = [], [], []
            for j in range(nb):
                a, b = int(starts[2 * j]), int(starts[2 * j + 1])
                if j:
                    kind = int(rng.choice([1, 1, 2]))
                    cl, cr = ri[-1] + 1, a - 1
                    u = rng.random()
                    if u < 0.04:
                        cl += 5                                  # a connector that does not touch its block
                    c.append(kind), le.append(cl), ri.append(max(cr, cl))
                c.append(0), le.append(a), ri.append(max(b - 1, a))
            if rng.random() < 0.02 and nb >= 2:                  # blocks out of order
                le[0], le[-1] = le[-1], le[0]
                ri[0], ri[-1] = ri[-1], ri[0]
            # half of the hits: snap the blocks to an isoform's exon ends, so that introns and exon ends are hit exactly
            if isoforms and isoforms[0] and rng.random() < 0.5:
                ex = isoforms[int(rng.integers(0, len(isoforms)))] or isoforms[0]
                k = int(rng.integers(0, len(ex)))
                c, le, ri = [0], [int(rng.integers(ex[k][0], ex[k][1] + 1))], [ex[k][1]]
                for k2 in range(k + 1, min(k + nb, len(ex))):
                    c += [int(rng.choice([1, 1, 1, 2])), 0]
                    le += [ri[-1] + 1, ex[k2][0]]
                    ri += [ex[k2][0] - 1, ex[k2][1] if k2 + 1 < min(k + nb, len(ex)) else int(rng.integers(ex[k2][0], ex[k2][1] + 1))]
                    if ri[-2] < le[-2]:                         # abutting exons: no room for a connector
                        del c[-2:], le[-2:], ri[-2:]
                        break
            loc.append(l), feats.append((c, le, ri))
    hits = eb.Hits(loc, feats)
    compat, key = eb.compat_and_keys(annot, hits, ctx)
    o_compat, o_key = oracle.exonbin_batch(annot, hits)
    np.testing.assert_array_equal(key, o_key)
    np.testing.assert_array_equal(compat, o_compat)
    frac = (compat != 0).any(axis=1).mean()
    assert 0.1 < frac < 0.9, frac                                 # both answers are common


def bins_arrays(b):
    return [b.row_off, b.f_off, b.count, b.bin_key, b.bin_compat, np.asarray(b.hit_bin), b.pair_seg_off, b.pair_seg_lens,
            b.pair_implicit_mask, b.pair_iso_len, b.pair_out_index]


def test_device_grouping_equals_host_grouping(ctx, oracle):
    """sbgpu_bins_create_device vs sbgpu_bins_create on the same words: every array identical.  The input has
    duplicate fragments (equal feature sequences: counted once, the first one's mass), masses above 1, hits
    without a compatible isoform, empty loci."""
    from strawberry_amd import exonbin as eb
    from strawberry_amd import synth
    from strawberry_amd.quantify import InsertSize, LocusQuantifier
    loci = synth.make_gene_models(150, seed=41)
    hl, pairs = synth.make_fragments(loci, 120, seed=42, noise=0.3)
    rng = np.random.default_rng(43)
    rows = []
    for l, (lb, rb) in zip(hl, pairs):
        if l % 17 == 3:
            continue                                  # some loci get no hits at all
        f = eb.hit_features(lb, rb)
        if f is None:
            continue
        rows.append((l, f[1][0], f[2][-1], f, float(rng.integers(1, 5))))
        if rng.random() < 0.15:                       # an equal fragment later in the same (left, right) run
            rows.append((l, f[1][0], f[2][-1], f, float(rng.integers(1, 9))))
        if rng.random() < 0.05:                       # same coordinates, GAP turned INTRON: another Contig code-wise,
            c2 = [1 if c == 2 else c for c in f[0]]   # the SAME std::set element (compared by offset and length only)
            rows.append((l, f[1][0], f[2][-1], (c2, f[1], f[2]), 3.0))
    rows.sort(key=lambda r: (r[0], r[1], r[2]))
    annot = eb.Annotation(loci)
    hits = eb.Hits([r[0] for r in rows], [r[3] for r in rows], mass=[r[4] for r in rows])
    qd = LocusQuantifier(annot, hits, InsertSize(250.0, 30.0), 75, ctx=ctx, device_bins=True)
    qh = LocusQuantifier(annot, hits, InsertSize(250.0, 30.0), 75, ctx=ctx, device_bins=False)
    bd, bh = qd.assign_bins(), qh.assign_bins()
    assert qd.bins_on_device and not qh.bins_on_device
    assert bh.n_bins > 2000 and bh.n_hits_used < hits.n_hits and (np.diff(bh.row_off) == 0).any()
    assert bd.n_hits_used == bh.n_hits_used
    for x, y in zip(bins_arrays(bd), bins_arrays(bh)):
        np.testing.assert_array_equal(x, y)
    # the duplicates mattered: counting every hit's mass would give other counts
    naive = np.bincount(bh.hit_bin[bh.hit_bin >= 0], weights=hits.mass[bh.hit_bin >= 0], minlength=bh.n_bins)
    assert (naive != bh.count).sum() > 50


def test_device_pairs_for_narrow_and_wide_loci_in_one_call(ctx):
    """The (bin, isoform) pairs come from two kernels: one wave per locus for loci of up to 64 isoforms, one thread per
    isoform for the others.  A batch with both kinds -- and loci of more than 64 bins, which the wave takes in several
    rounds -- gives the host code's pair arrays, entry for entry."""
    from strawberry_amd import exonbin as eb
    from strawberry_amd import synth
    from strawberry_amd.quantify import InsertSize, LocusQuantifier
    loci = synth.make_gene_models(40, seed=51, max_exons=30, max_isoforms=90, ex_lo=60, ex_hi=160)
    n_iso = np.array([len(l) for l in loci])
    assert (n_iso > 64).any() and (n_iso <= 64).any()
    hl, pairs = synth.make_fragments(loci, 900, seed=52)
    rows = []
    for l, (lb, rb) in zip(hl, pairs):
        f = eb.hit_features(lb, rb)
        if f is not None:
            rows.append((l, f[1][0], f[2][-1], f))
    rows.sort(key=lambda r: (r[0], r[1], r[2]))
    annot = eb.Annotation(loci)
    hits = eb.Hits([r[0] for r in rows], [r[3] for r in rows])
    qd = LocusQuantifier(annot, hits, InsertSize(250.0, 30.0), 75, ctx=ctx, device_bins=True)
    qh = LocusQuantifier(annot, hits, InsertSize(250.0, 30.0), 75, ctx=ctx, device_bins=False)
    bd, bh = qd.assign_bins(), qh.assign_bins()
    assert qd.bins_on_device and not qh.bins_on_device
    assert (np.diff(bh.row_off) > 64).any() and bh.n_pairs > 10000
    for x, y in zip(bins_arrays(bd), bins_arrays(bh)):
        np.testing.assert_array_equal(x, y)


def group_both_ways(ctx, annot, hits, compat, key):
    """(device bins or None, host bins) for arbitrary word arrays (they need not come from the kernel)."""
    import ctypes as C
    import torch
    from strawberry_amd import exonbin as eb
    dev = torch.device("cuda", ctx.device)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    d = {k: up(getattr(hits, k).view(np.int32) if getattr(hits, k).dtype == np.uint32 else getattr(hits, k))
         for k in ("hit_locus", "feat_off", "feat_code", "feat_left", "feat_right")}
    d_mass, d_compat, d_key = up(hits.mass), up(compat.view(np.int32)), up(key.view(np.int32))
    d_hit_bin = torch.zeros(hits.n_hits, dtype=torch.int64, device=dev)
    from strawberry_amd import _lib
    ht = _lib.sbgpu_hits_t(hits.n_hits, d["hit_locus"].data_ptr(), d["feat_off"].data_ptr(), d["feat_code"].data_ptr(),
                           d["feat_left"].data_ptr(), d["feat_right"].data_ptr())
    bd = eb.LocusBins.on_device(ctx, annot, hits, ht, d_mass.data_ptr(), compat.shape[1], key.shape[1], d_compat.data_ptr(),
                                d_key.data_ptr(), d_hit_bin.data_ptr(), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if bd is not None:
        bd.hit_bin = d_hit_bin.cpu().numpy()
    return bd, eb.LocusBins(annot, hits, compat, key)


def test_device_grouping_multiword_keys_and_table_limit(ctx):
    """Keys of several words (loci with more than 32 segments), compat of several words, many small loci in
    one call -- and a locus with more bins than the LDS table holds, which the device form must decline."""
    from strawberry_amd import exonbin as eb
    rng = np.random.default_rng(77)
    # 30 loci of 71 isoforms that all hold the same 130 exons (= 130 segments): any set of segments is under any
    # isoform, so arbitrary key / compat words are consistent input.  3 compat words, 5 key words.
    loci = [[[(1000 * (l + 1) * 200 + 100 * k, 1000 * (l + 1) * 200 + 100 * k + 49) for k in range(130)]] * 71 for l in range(30)]
    annot = eb.Annotation(loci)
    assert annot.compat_words == 3 and annot.key_words == 5
    def make(n_keys_per_locus, n_hits_per_locus, n_loci=30):
        loc, feats, masses, compat, key = [], [], [], [], []
        for l in range(n_loci):
            base = 1000 * (l + 1) * 200
            keys = rng.integers(1, 2 ** 32, (n_keys_per_locus, 5), dtype=np.uint64).astype(np.uint32)
            keys[:, 4] &= 3                      # 130 segments: two bits in the last word
            comps = rng.integers(0, 2 ** 32, (n_keys_per_locus, 3), dtype=np.uint64).astype(np.uint32)
            comps[:, 2] &= (1 << 7) - 1          # 71 isoforms
            pick = np.sort(rng.integers(0, n_keys_per_locus, n_hits_per_locus))
            lefts = np.sort(rng.integers(base, base + 12000, n_hits_per_locus))
            for q in range(n_hits_per_locus):
                loc.append(l)
                feats.append(([0], [int(lefts[q])], [int(lefts[q]) + 74]))
                masses.append(float(rng.integers(1, 4)))
                key.append(keys[pick[q]])
                c = comps[pick[q]].copy()
                if rng.random() < 0.1:
                    c[:] = 0                      # a hit without a compatible isoform
                compat.append(c)
        return eb.Hits(loc, feats, mass=masses), np.array(compat, np.uint32), np.array(key, np.uint32)
    hits, compat, key = make(40, 400)
    bd, bh = group_both_ways(ctx, annot, hits, compat, key)
    assert bd is not None and bh.n_bins > 1000
    for x, y in zip(bins_arrays_no_pairs(bd), bins_arrays_no_pairs(bh)):
        np.testing.assert_array_equal(x, y)
    # a locus with ~4000 bins goes through the big table
    hits, compat, key = make(5000, 12000, n_loci=1)
    annot1 = eb.Annotation(loci[:1])
    bd, bh = group_both_ways(ctx, annot1, hits, compat, key)
    assert bd is not None and bh.n_bins > 3000
    for x, y in zip(bins_arrays_no_pairs(bd), bins_arrays_no_pairs(bh)):
        np.testing.assert_array_equal(x, y)
    # ~12000 different keys in one locus: more than the (big) table holds -> declined, the host form does it
    hits, compat, key = make(14000, 36000, n_loci=1)
    annot1 = eb.Annotation(loci[:1])
    bd, bh = group_both_ways(ctx, annot1, hits, compat, key)
    assert bd is None and bh.n_bins > 8192


def test_device_grouping_single_pass_with_a_locus_for_the_big_table(ctx):
    """One-word compat, two-word keys: the single-pass kernels group the loci; a locus with more bins than their table
    holds (here ~3000 > 1400) is redone by the big-table kernel, whose entries are zeroed for it alone, while its
    neighbours' bins stand.  The middle kernels, launched before the bin counts were back, run again over the result."""
    from strawberry_amd import exonbin as eb
    rng = np.random.default_rng(78)
    loci = [[[(1000 * (l + 1) * 200 + 100 * k, 1000 * (l + 1) * 200 + 100 * k + 49) for k in range(40)]] * 20 for l in range(3)]
    annot = eb.Annotation(loci)
    assert annot.compat_words == 1 and annot.key_words == 2
    loc, feats, masses, compat, key = [], [], [], [], []
    for l, (n_keys, n_hits) in enumerate([(300, 2500), (3000, 9000), (200, 1500)]):
        base = 1000 * (l + 1) * 200
        keys = rng.integers(1, 2 ** 32, (n_keys, 2), dtype=np.uint64).astype(np.uint32)
        keys[:, 1] &= 255                   # 40 segments
        comps = rng.integers(1, 2 ** 20, (n_keys, 1), dtype=np.uint64).astype(np.uint32)   # 20 isoforms
        pick = np.sort(rng.integers(0, n_keys, n_hits))
        # distinct fragments: the single-pass kernels rest on "equal fragments have equal words" (true of words made from
        # the fragments; not of the arbitrary words of this test)
        lefts = np.sort(rng.choice(np.arange(base, base + 12000), n_hits, replace=False))
        for q in range(n_hits):
            loc.append(l)
            feats.append(([0], [int(lefts[q])], [int(lefts[q]) + 74]))
            masses.append(float(rng.integers(1, 4)))
            key.append(keys[pick[q]])
            compat.append(comps[pick[q]])
    hits, compat, key = eb.Hits(loc, feats, mass=masses), np.array(compat, np.uint32), np.array(key, np.uint32)
    bd, bh = group_both_ways(ctx, annot, hits, compat, key)
    assert bd is not None
    nb = np.diff(bh.row_off)
    assert nb[1] > 1400 and nb[0] < 1400 and nb[2] < 1400
    for x, y in zip(bins_arrays_no_pairs(bd), bins_arrays_no_pairs(bh)):
        np.testing.assert_array_equal(x, y)


def bins_arrays_no_pairs(b):
    return [b.row_off, b.f_off, b.count, b.bin_key, b.bin_compat, np.asarray(b.hit_bin)]


def test_device_grouping_declines_what_it_cannot_do_exactly(ctx):
    from strawberry_amd import exonbin as eb
    from strawberry_amd import synth
    from strawberry_amd.quantify import InsertSize, LocusQuantifier
    loci = synth.make_gene_models(20, seed=51)
    hl, pairs = synth.make_fragments(loci, 60, seed=52)
    rows = [(l, eb.hit_features(lb, rb)) for l, (lb, rb) in zip(hl, pairs)]
    rows = [(l, f) for l, f in rows if f is not None]
    annot = eb.Annotation(loci)
    ok = eb.Hits([l for l, _ in rows], [f for _, f in rows])
    q = LocusQuantifier(annot, ok, InsertSize(250.0, 30.0), 75, ctx=ctx)
    q.assign_bins()
    assert q.bins_on_device
    ref = bins_arrays(q.bins)
    # fractional masses (multi-mapped reads): the float accumulation order matters -- the device form sums every
    # bin again in the order of the reference's std::set<Contig> and must agree with the host form bit for bit
    rng = np.random.default_rng(7)
    for trial in range(3):
        # sums of 1, 1/2, 1/3 (NH 1-3, collapsed duplicates), many fragments starting at the same position
        m = (rng.choice([1.0, 0.5, 1.0 / 3.0], len(rows)) * rng.integers(1, 4, len(rows))).astype(np.float32)
        frac = eb.Hits([l for l, _ in rows], [f for _, f in rows], mass=m)
        q = LocusQuantifier(annot, frac, InsertSize(250.0, 30.0), 75, ctx=ctx)
        q.assign_bins()
        assert q.bins_on_device
        qh = LocusQuantifier(annot, frac, InsertSize(250.0, 30.0), 75, ctx=ctx, device_bins=False)
        qh.assign_bins()
        for x, y in zip(bins_arrays(q.bins), bins_arrays(qh.bins)):
            np.testing.assert_array_equal(x, y)
    # hits of a locus not in (left, right) order: equal fragments need not be neighbours -> host, same bins as sets
    rev = list(reversed(rows))
    rev.sort(key=lambda r: r[0])
    unsorted = eb.Hits([l for l, _ in rev], [f for _, f in rev])
    q = LocusQuantifier(annot, unsorted, InsertSize(250.0, 30.0), 75, ctx=ctx)
    b = q.assign_bins()
    assert not q.bins_on_device and b.n_bins == q.bins.n_bins == len(ref[2])
    assert sorted(b.count.tolist()) == sorted(ref[2].tolist())


def test_one_call_chain_equals_the_staged_chain(ctx):
    """sbgpu_quantify_host vs LocusQuantifier (the same kernels driven stage by stage): identical theta, F, bins --
    with hits grouped by locus (device grouping), with hits of different loci interleaved (host grouping),
    with the empirical insert size, and with no hits at all."""
    from strawberry_amd import exonbin as eb
    from strawberry_amd import synth
    from strawberry_amd.quantify import InsertSize, LocusQuantifier, quantify_host
    loci = synth.make_gene_models(60, seed=61)
    hl, pairs = synth.make_fragments(loci, 90, seed=62, noise=0.2)
    rows = [(l, eb.hit_features(lb, rb)) for l, (lb, rb) in zip(hl, pairs)]
    rows = [(l, f) for l, f in rows if f is not None]
    annot = eb.Annotation(loci)
    hits = eb.Hits([l for l, _ in rows], [f for _, f in rows])
    q = LocusQuantifier(annot, hits, InsertSize(250.0, 30.0), 75, ctx=ctx)
    ref = q.run(hits.n_hits, min_isoform_frac=0.0)
    one = quantify_host(annot, hits, InsertSize(250.0, 30.0), 75, ctx=ctx)
    np.testing.assert_array_equal(one["theta"], ref["theta"])
    np.testing.assert_array_equal(one["iters"], ref["iters"])
    np.testing.assert_array_equal(one["F"], q.d_F.cpu().numpy()[:len(one["F"])])
    for x, y in zip(bins_arrays(one["bins"]), bins_arrays(q.bins)):
        np.testing.assert_array_equal(x, y)
    # hits of different loci interleaved: grouping falls back to the host, results are the same per locus
    perm = np.argsort(np.arange(hits.n_hits) % 7, kind="stable")
    mixed = eb.Hits([rows[i][0] for i in perm], [rows[i][1] for i in perm])
    two = quantify_host(annot, mixed, InsertSize(250.0, 30.0), 75, ctx=ctx)
    assert two["bins"].n_bins == one["bins"].n_bins
    assert abs(two["theta"].sum() - one["theta"].sum()) < 1e-6 * one["theta"].sum()
    # empirical insert size built by the library = built by hand from the same words
    emp = quantify_host(annot, hits, None, 75, ctx=ctx)
    fl = eb.frag_lens(annot, hits, one["compat"])
    byhand = InsertSize.from_frag_lens(fl)
    assert emp["insert"]["use_emp"] and emp["insert"]["total_reads"] == len(fl)
    assert emp["insert"]["mean"] == byhand.mean and emp["insert"]["sd"] == byhand.sd
    np.testing.assert_array_equal(emp["insert"]["emp_hist"], byhand.emp_hist)
    q2 = LocusQuantifier(annot, hits, byhand, 75, ctx=ctx)
    np.testing.assert_array_equal(emp["theta"], q2.run(hits.n_hits, min_isoform_frac=0.0)["theta"])
    # no hits: every locus is INIT_EMPTY, nothing breaks
    none = quantify_host(annot, eb.Hits([], []), InsertSize(250.0, 30.0), 75, ctx=ctx)
    assert none["bins"].n_bins == 0 and (none["status"] == 1).all()


def test_large_batch_properties(ctx, oracle):
    """BASELINE-scale hit counts through size-independent properties: tiling the loci along the genome
    must tile the answers; a fragment sampled from an isoform without noise is compatible with it."""
    from strawberry_amd import exonbin as eb
    from strawberry_amd import synth
    loci = synth.make_gene_models(400, seed=21)
    hl, pairs = synth.make_fragments(loci, 100, seed=22, noise=0.0, single=0.0)
    feats, loc = [], []
    for l, (lb, rb) in zip(hl, pairs):
        f = eb.hit_features(lb, rb)
        if f is not None:
            feats.append(f)
            loc.append(l)
    annot, hits = eb.Annotation(loci), eb.Hits(loc, feats)
    compat, key = eb.compat_and_keys(annot, hits, ctx)
    assert (compat != 0).any(1).all() and (key != 0).any(1).all()
    big_annot, big_hits = XU.tile(annot, hits, 64)
    assert big_hits.n_hits == 64 * hits.n_hits > 2_000_000
    c2, k2 = eb.compat_and_keys(big_annot, big_hits, ctx)
    np.testing.assert_array_equal(c2.reshape(64, hits.n_hits, -1), np.broadcast_to(compat, (64,) + compat.shape))
    np.testing.assert_array_equal(k2.reshape(64, hits.n_hits, -1), np.broadcast_to(key, (64,) + key.shape))


@pytest.mark.parametrize("which", ["E2E", "E2E_LONG", "E2E_MASS", "E2E_FILTER", "E2E_EMP", "E2E_SINGLE", "E2E_LONGREAD", "E2E_MINUS", "E2E_CHROMS"])
def test_chain_from_fragments_reproduces_reference_run(ctx, oracle, which):
    """Our simulated fragments (reads.npz, the BAM the reference binary was run on) through the whole
    device chain: bins and counts == the -f table; every nonzero weight of the table == F (12 digits);
    theta == the reference's log; FPKM / Frac / TPM == its GTF."""
    from strawberry_amd.quantify import InsertSize, LocusQuantifier
    d = getattr(U, which)
    ordered, rows, gtf, theta_log = U.load(d)
    annot, hits, names, _ = XU.e2e_inputs(d, ordered)
    # single-end library: the reference forces the insert size to N(200, 80) (Strawberry.cpp:329-333)
    # long reads (more than ten read lengths above 1000, Strawberry.cpp:292-303): every weight is 1/L_j
    unpaired = which in ("E2E_SINGLE", "E2E_LONGREAD")
    q = LocusQuantifier(annot, hits, InsertSize(200.0, 80.0) if unpaired else InsertSize(MEAN, SD), RL,
                        long_read=(which == "E2E_LONGREAD"), ctx=ctx)
    if unpaired:
        assert (hits.feat_code != 2).all()      # no mate gaps: every hit is a single read
    bins = q.assign_bins()
    # grouped on the device wherever the order of the float accumulation cannot matter (whole-number masses)
    assert q.bins_on_device   # all ten runs, the fractional-mass one included, take the device grouping
    o_compat, o_key = oracle.exonbin_batch(annot, hits)
    np.testing.assert_array_equal(q.d_compat.cpu().numpy().view(np.uint32)[:hits.n_hits], o_compat)
    np.testing.assert_array_equal(q.d_key.cpu().numpy().view(np.uint32)[:hits.n_hits], o_key)
    if which == "E2E_EMP":   # no -i: the empirical insert-size distribution, built from the hits (pass 1)
        from strawberry_amd import exonbin as eb
        q.insert = InsertSize.from_frag_lens(eb.frag_lens(annot, hits, o_compat))
        assert q.insert.use_emp and q.insert.total_reads > 1000 and abs(q.insert.mean - MEAN) < 5
    F = q.bin_weights().cpu().numpy()
    n_checked = 0
    for l, g in enumerate(names):
        coords = [tuple(c) for c in bins.bin_coords(l)]
        niso = int(bins.iso_off[l + 1] - bins.iso_off[l])
        Fl = F[bins.f_off[l]:bins.f_off[l + 1]].reshape(len(coords), niso)
        ref_rows = [r for r in rows if r["gene"] == g]
        if which == "E2E_FILTER":   # the table lost the erased isoforms' columns (and some bins): compared as a file below
            assert set(tuple(r["coords"]) for r in ref_rows) <= set(coords)
            n_checked += len(ref_rows)
            continue
        assert sorted(coords) == sorted(tuple(r["coords"]) for r in ref_rows)
        for r in ref_rows:
            b = coords.index(tuple(r["coords"]))
            if which != "E2E_MASS":     # unit masses: n_i == the table's number of unique hits
                assert bins.count[bins.row_off[l] + b] == r["count"]
            for j, f in enumerate(r["F"]):
                if f != 0.0:
                    assert abs(Fl[b, j] - f) <= 5e-11 * f, (g, r["coords"], j)
                    n_checked += 1
    assert n_checked > (50 if which == "E2E_FILTER" else 100)
    assert hits.total_mapped == rows[0]["total_mapped"]
    min_frac = 0.05 if which == "E2E_FILTER" else 0.0   # -r: kMinIsoformFrac = 0; -r -e 0.05: 0.05 (Strawberry.cpp:158-177)
    res = q.solve(hits.total_mapped, min_isoform_frac=min_frac)
    assert (res["keep"] == 0).sum() == (5 if which == "E2E_FILTER" else 0)
    for l, ref_theta in enumerate(theta_log):
        th = res["theta"][bins.iso_off[l]:bins.iso_off[l + 1]]
        assert np.abs(th - np.array(ref_theta)).max() < 1e-6, (names[l], th, ref_theta)
    tx_names = [t for g in names for t, _ in ordered[g]]
    assert set(gtf) == set(t for t, k in zip(tx_names, res["keep"]) if k)   # exactly the survivors are written
    for t, f, fr, tp, k in zip(tx_names, res["fpkm"], res["frac"], res["tpm"], res["keep"]):
        if not k:
            continue
        assert abs(f - float(gtf[t][0])) <= 1e-5 * max(1.0, f), t       # FPKM/TPM: 1e-4 rel is the bar; we are tighter
        assert abs(fr - float(gtf[t][1])) < 2e-6, t
        assert abs(tp - float(gtf[t][2])) <= 1e-5 * max(1.0, tp), t
    # and the EM input we built equals what the oracle gets from the same F
    o_theta, o_status, o_iters = oracle.em_batch(bins.row_off, bins.iso_off, bins.f_off, bins.count, F)
    np.testing.assert_array_equal(res["iters"], o_iters)
    np.testing.assert_array_equal(res["status"], o_status)

    # the reference's two output FILES, byte for byte, from the fragments alone
    from strawberry_amd.output import context_table, gtf_transcript
    compat = q.d_compat.cpu().numpy().view(np.uint32)[:hits.n_hits]
    table = context_table("toy", rows[0]["total_mapped"], names, [[t for t, _ in ordered[g]] for g in names], bins,
                          compat, F, res["fpkm"], res["frac"], keep=res["keep"])
    assert table == open(os.path.join(d, "ctx.tsv")).read()
    gtf_text = []
    k = 0
    strands, chroms = U.gene_strands(d), U.gene_chroms(d)
    for g in names:
        for t, ex in ordered[g]:
            if res["keep"][k]:
                gtf_text.append(gtf_transcript(chroms[g], strands[g], g, t, ex, res["fpkm"][k], res["frac"][k], res["tpm"][k],
                                               ref_gene_id=g, ref_gene_name=g))
            k += 1
    ref_gtf = open(os.path.join(d, "out.gtf")).read().split("\n", 2)
    assert ref_gtf[0].startswith("#") and ref_gtf[1].startswith("#")   # command line + rule: not data
    assert "".join(gtf_text) == ref_gtf[2]


def test_quantify_device_equals_quantify_host_on_the_chain_sample(oracle):
    """The chain workload's sample (strawberry_amd/chain.py: distinct gene models, read pairs drawn on the device) is a
    well-formed input -- sorted by (locus, left end, right end), every hit M (x M)* with one GAP -- and
    sbgpu_quantify_device on the hits in HBM gives the same theta, status and iteration counts as sbgpu_quantify_host on
    the same hits brought to the host; the words of every hit equal the oracle's."""
    from strawberry_amd import chain, em
    from strawberry_amd.quantify import InsertSize, quantify_host
    ctx = em.default_context(0)
    q = chain.ChainQuantifier(ctx, n_loci=400, n_frags=400 * 300, seed=5)
    q.step()
    hits = q.hits.host_hits(q.n_loci)
    assert hits.n_hits == q.n_hits > 50000 and q.n_frags == int(hits.mass.sum()) >= q.n_hits
    assert (hits.mass >= 1).all() and (hits.mass > 1).any()          # equal fragments were merged into unique hits
    # the order HitCluster::collapseAndFilterHits leaves: (locus, left end, right end)
    first, last = hits.feat_left[hits.feat_off[:-1]].astype(np.int64), hits.feat_right[hits.feat_off[1:] - 1].astype(np.int64)
    key = (hits.hit_locus.astype(np.int64) << 40) * 0 + first * (1 << 31) + last
    assert (np.diff(hits.hit_locus) >= 0).all() and (np.diff(key) >= 0).all()
    assert (hits.feat_code[hits.feat_off[:-1]] == 0).all() and (hits.feat_code[hits.feat_off[1:] - 1] == 0).all()
    assert (np.bincount(np.repeat(np.arange(hits.n_hits), np.diff(hits.feat_off)), weights=(hits.feat_code == 2)) == 1).all()
    r = quantify_host(q.annot, hits, InsertSize(250.0, 30.0), 75, ctx=ctx)
    np.testing.assert_array_equal(q.status[:q.n_loci], r["status"])
    np.testing.assert_array_equal(q.iters[:q.n_loci], r["iters"])
    np.testing.assert_array_equal(q.theta[:q.n_iso], r["theta"])
    oc, _ = oracle.exonbin_batch(q.annot, hits)
    np.testing.assert_array_equal(r["compat"], oc)
    # pairs that fit no isoform exist (shifted / unspliced mates), and most pairs fit one
    frac = float((oc != 0).any(axis=1).mean())
    assert 0.85 < frac < 0.999
    # a rank's share of the same sample (strong scaling): its loci are exactly its residue class, same gene models
    q2 = chain.ChainQuantifier(ctx, n_loci=400, n_frags=400 * 300, seed=5, loci_subset=(1, 2))
    assert q2.n_loci == 200
    np.testing.assert_array_equal(q2.annot.exon_left[:int(q2.annot.exon_off[int(q2.annot.iso_off[1])])],
                                  q.annot.exon_left[int(q.annot.exon_off[int(q.annot.iso_off[1])]):int(q.annot.exon_off[int(q.annot.iso_off[2])])])
    q.close()       # (the pin is keyed on the annotation arrays: it goes before they do)
    q2.close()


@pytest.mark.gpu
def test_pinned_annotation_gives_the_unpinned_results():
    """sbgpu_annotation_pin keeps the annotation's device copies and tables with the context; calls that are given the same
    arrays use them.  Same theta / status / iterations, bit for bit, as the call that uploads everything itself; an
    annotation that is NOT the pinned one takes the ordinary path; unpin restores it for the pinned one too."""
    from strawberry_amd import chain, em
    ctx = em.default_context(0)
    a = chain.ChainQuantifier(ctx, n_loci=300, n_frags=300 * 200, seed=9, pin=False)
    a.step()
    want = (a.theta[:a.n_iso].copy(), a.status[:a.n_loci].copy(), a.iters[:a.n_loci].copy())
    b = chain.ChainQuantifier(ctx, n_loci=300, n_frags=300 * 200, seed=9, pin=True)
    for _ in range(2):
        b.theta[:] = -1
        b.step()
        for got, w in zip((b.theta[:b.n_iso], b.status[:b.n_loci], b.iters[:b.n_loci]), want):
            np.testing.assert_array_equal(got, w)
    # another annotation while b's is pinned: not the resident one, so it is uploaded as ever
    c = chain.ChainQuantifier(ctx, n_loci=200, n_frags=200 * 150, seed=10, pin=False)
    c.step()
    c_first = c.theta.copy()
    b.unpin()
    c.step()
    np.testing.assert_array_equal(c.theta, c_first)
    b.theta[:] = -1
    b.step()
    np.testing.assert_array_equal(b.theta[:b.n_iso], want[0])


@pytest.mark.gpu
def test_a_quantifier_releases_only_its_own_pin():
    """One pin per context: a second quantifier's pin replaces the first one's.  The first owner's release (close / unpin /
    garbage collection) must then leave the second pin alone -- sbgpu_annotation_unpin would drop it, and every later step
    would quietly go back to uploading the annotation."""
    import ctypes as C
    from strawberry_amd import chain, em
    ctx = em.default_context(0)
    a = chain.ChainQuantifier(ctx, n_loci=120, n_frags=120 * 100, seed=3, pin=True)
    b = chain.ChainQuantifier(ctx, n_loci=150, n_frags=150 * 100, seed=4, pin=True)      # replaces a's pin
    b.step()
    want = b.theta.copy()
    a.close()                                                                              # must not touch b's pin
    released = C.c_int32(-1)
    assert ctx.L.sbgpu_annotation_unpin_matching(ctx.h, C.byref(a._an), C.byref(released)) == 0 and released.value == 0
    b.step()
    np.testing.assert_array_equal(b.theta, want)
    assert ctx.L.sbgpu_annotation_unpin_matching(ctx.h, C.byref(b._an), C.byref(released)) == 0 and released.value == 1   # it was still there
    b.pinned = False
    b.step()
    np.testing.assert_array_equal(b.theta, want)


@pytest.mark.gpu
def test_idle_arenas_are_kept_for_the_next_call_and_can_be_released():
    """The arenas of released handles go back to a process-wide pool (up to half of the device's memory: a sample-sized
    call's arenas are tens of GB, and allocating them per call costs seconds); sbgpu_release_idle_memory hands them back
    to the driver, and the next call allocates again -- same results either way."""
    from strawberry_amd import chain, em
    ctx = em.default_context(0)
    with chain.ChainQuantifier(ctx, n_loci=400, n_frags=400 * 300, seed=12, pin=False) as q:
        q.step()
        want = q.theta.copy()
        q.step()                                       # (the first step's handle arenas are in the pool by now)
        np.testing.assert_array_equal(q.theta, want)
        released = ctx.L.sbgpu_release_idle_memory()
        assert released > 0
        assert ctx.L.sbgpu_release_idle_memory() == 0  # nothing idle any more
        q.step()
        np.testing.assert_array_equal(q.theta, want)


def test_segment_basis_128_bit_form(ctx, oracle):
    """Loci of 65-128 segments or isoforms take the segment basis with 128-bit masks in a kernel of their own
    (exonbin_seg128_kernel, launched behind exonbin_kernel when the annotation's words go beyond two); beyond 128 the exon
    walk.  Same grid construction as above -- exon ends on multiples of 10, blocks on multiples of 5, snapped hits that meet
    exon ends and introns exactly -- on loci of ~70, ~100, 128 and 129+ segments, 3 to 120 isoforms, next to small loci
    (a wave that spans a small and a big locus).  Words equal the oracle's, bit for bit."""
    from strawberry_amd import exonbin as eb
    rng = np.random.default_rng(4096)
    shapes = [(70, 6), (12, 4), (100, 40), (128, 9), (140, 5), (30, 90), (64, 120), (9, 3), (127, 127), (200, 130)]   # (cells, isoforms)
    loci = []
    for l, (n_cells, n_iso) in enumerate(shapes):
        base = 5000 * (l + 1)
        isoforms = [[(base + 10 * k, base + 10 * k + 6) for k in range(n_cells)]]      # one isoform with every cell: n_cells segments at least
        for _ in range(n_iso - 1):
            exons, pos = [], int(rng.integers(0, 4))
            while pos < n_cells:
                if rng.random() < 0.6:
                    exons.append((base + 10 * pos, base + 10 * pos + 6))
                pos += 1 + int(rng.choice([0, 0, 1, 3]))
            isoforms.append(exons or [(base, base + 6)])
        loci.append(isoforms)
    annot = eb.Annotation(loci)
    nseg, niso = np.diff(annot.seg_off), np.diff(annot.iso_off)
    assert annot.key_words >= 5 and annot.compat_words >= 4
    assert ((nseg > 64) & (nseg <= 128)).sum() >= 4 and (nseg > 128).sum() >= 2 and ((niso > 64) & (niso <= 128)).sum() >= 3
    loc, feats = [], []
    for l, isoforms in enumerate(loci):
        n_cells = shapes[l][0]
        base = 5000 * (l + 1)
        for _ in range(500):
            ex = isoforms[int(rng.integers(0, len(isoforms)))]
            k = int(rng.integers(0, len(ex)))
            nb = int(rng.integers(1, 5))
            c, le, ri = [0], [int(rng.integers(ex[k][0], ex[k][1] + 1))], [ex[k][1]]
            for k2 in range(k + 1, min(k + nb, len(ex))):
                last = k2 + 1 == min(k + nb, len(ex))
                c += [int(rng.choice([1, 1, 1, 2])), 0]
                le += [ri[-1] + 1, ex[k2][0]]
                ri += [ex[k2][0] - 1, int(rng.integers(ex[k2][0], ex[k2][1] + 1)) if last else ex[k2][1]]
            u = rng.random()
            if u < 0.15:                                   # off the grid: a block that sticks out of its exon, or sits in an intron
                j = 2 * int(rng.integers(0, (len(c) + 1) // 2))
                ri[j] += int(rng.choice([1, 3, 5]))
                if j + 1 < len(c):
                    le[j + 1] = ri[j] + 1
                    ri[j + 1] = max(ri[j + 1], le[j + 1])
            elif u < 0.2:                                  # blocks out of order (the per-lane walk)
                le[0], le[-1], ri[0], ri[-1] = le[-1], le[0], ri[-1], ri[0]
            loc.append(l), feats.append((c, le, ri))
    order = np.argsort(np.asarray(loc), kind="stable")
    hits = eb.Hits([loc[i] for i in order], [feats[i] for i in order])
    compat, key = eb.compat_and_keys(annot, hits, ctx)
    o_compat, o_key = oracle.exonbin_batch(annot, hits)
    np.testing.assert_array_equal(key, o_key)
    np.testing.assert_array_equal(compat, o_compat)
    frac = (compat != 0).any(axis=1).mean()
    assert 0.3 < frac < 0.97, frac
