"""GPU tests of sbgpu_collapse_pairs_device (HitCluster::collapseAndFilterHits + Contig(PairedHit) on the GPU,
/root/reference/src/alignments.cpp:656-703, src/contig.cpp:216-267) against the ORACLE (oracle/collapse_oracle.c, pinned
to the reference's own HitCluster by tests/test_collapse_oracle.py; features by the reference's Contig(PairedHit) where
oracle/_ref is built) and against sbgpu_collapse_pairs_host: unique hits, features, float masses, cluster masses and
the int-truncated mapped-read total must be identical."""
import numpy as np
import pytest

import e2e_util as U
import exonbin_util as XU

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from strawberry_amd import em
    return em.default_context(0)


def same(got, ref, gcm, rcm, ginfo, rinfo):
    assert ginfo == rinfo
    for a in ("hit_locus", "feat_off", "feat_code", "feat_left", "feat_right", "mass"):
        np.testing.assert_array_equal(getattr(got, a), getattr(ref, a), err_msg=a)
    np.testing.assert_array_equal(gcm, rcm)


@pytest.mark.parametrize("which", ["E2E", "E2E_MASS", "E2E_SINGLE", "E2E_LONGREAD"])
def test_device_collapse_equals_host_on_the_reference_runs(ctx, which):
    """Every sequenced copy of the toy runs (shuffled): PCR duplicates, NH 2/3 masses, single-end and long reads."""
    from strawberry_amd import exonbin as eb
    d = getattr(U, which)
    ordered, rows, _, _ = U.load(d)
    annot, hits, names, rejected = XU.e2e_inputs(d, ordered)
    copies = XU.load_read_copies(d)
    rng = np.random.default_rng(5)
    copies = [copies[i] for i in rng.permutation(len(copies))]
    args = (len(names), [c[0] for c in copies], [c[3] for c in copies], [c[1] for c in copies], [c[2] for c in copies])
    ref, rcm, rinfo = eb.collapse_pairs(*args)
    if which == "E2E_LONGREAD":
        # reads of up to 11 blocks = 21 features per mate: inside the device form's 24
        pass
    got, gcm, ginfo = eb.collapse_pairs(*args, device=ctx)
    same(got, ref, gcm, rcm, ginfo, rinfo)
    assert ginfo["total_mapped"] == rows[0]["total_mapped"] == hits.total_mapped


def test_device_collapse_filter_equality_and_rejects(ctx):
    from strawberry_amd import exonbin as eb
    left = [[(1000 + 3 * k, 1074 + 3 * k)] for k in range(200)]
    right = [[(1300 + 3 * k, 1374 + 3 * k)] for k in range(200)]
    left.append([(1500, 1574)]); right.append([(1700, 1710), (30000, 30063)])      # a mate spanning 28 kb: filtered
    left.append([(1003, 1077)]); right.append([(1303, 1377)])                      # a second copy of pair 1
    left.append([(1003, 1077)]); right.append([(1303, 1340), (1400, 1436)])        # spliced: not equal
    args = (1, [0] * 203, [1.0] * 201 + [0.5, 1.0], left, right)
    g, r = eb.collapse_pairs(*args, device=ctx), eb.collapse_pairs(*args)
    same(g[0], r[0], g[1], r[1], g[2], r[2])
    assert g[2]["filtered"] == 1 and g[0].n_hits == 201
    # single reads, abutting mates (rejected but counted), overlapping mates (merged), an empty locus in between
    args = (4, [0, 0, 1, 3, 3], [1.0, 1.0, 1.0, 0.5, 1.0 / 3],
            [[(10, 84)], [(10, 84)], [(500, 574)], [(900, 974)], [(900, 950), (1000, 1023)]],
            [[], [], [(575, 649)], [(950, 1024)], [(1010, 1084)]])
    g, r = eb.collapse_pairs(*args, device=ctx), eb.collapse_pairs(*args)
    same(g[0], r[0], g[1], r[1], g[2], r[2])
    assert g[2]["rejected"] == 1 and g[0].n_hits == 3


def test_device_collapse_random_stress(ctx, oracle):
    """Random clusters: many duplicates, fractional masses (sums whose float / int truncation depends on the order),
    equal (left, right) ends with different blocks, single reads, spliced mates, up to 4000 pairs in a locus (bigger
    ones: test_device_collapse_big_loci)."""
    from strawberry_amd import exonbin as eb
    rng = np.random.default_rng(123)
    for trial in range(4):
        n_loci = int(rng.integers(3, 40))
        loc, mass, left, right = [], [], [], []
        for l in range(n_loci):
            n = int(rng.integers(0, 4000 if l == 0 else 300))
            base = 100000 * (l + 1)
            starts = rng.integers(base, base + 400, n)
            for k in range(n):
                s = int(starts[k])
                rl = 75
                lb = [(s, s + rl - 1)]
                if rng.random() < 0.3:                       # spliced left mate
                    cut = int(rng.integers(10, 60))
                    gap = int(rng.choice([200, 350]))
                    lb = [(s, s + cut - 1), (s + cut + gap, s + gap + rl - 1)]
                u = rng.random()
                if u < 0.1:
                    rb = []                                   # single read
                else:
                    ins = int(rng.choice([180, 200, 230, 75, 40]))   # apart, abutting (75) and overlapping (40) mates
                    rs = lb[-1][1] + 1 + ins - rl if ins > rl else lb[0][0] + ins
                    rb = [(rs, rs + rl - 1)]
                loc.append(l)
                mass.append(int(rng.choice([1, 2, 3, 4])))     # the pair's NH tag: mass 1 / NH
                left.append(lb)
                right.append(rb)
        perm = rng.permutation(len(loc))
        nh = [mass[i] for i in perm]
        args = (n_loci, [loc[i] for i in perm], [1.0 / k for k in nh], [left[i] for i in perm], [right[i] for i in perm])
        g, r = eb.collapse_pairs(*args, device=ctx), eb.collapse_pairs(*args)
        same(g[0], r[0], g[1], r[1], g[2], r[2])
        assert r[0].n_hits < len(loc) * 0.9      # the stress does collapse
        # ... and against the independent checker, cluster by cluster
        XU.check_collapse_against_oracle(oracle, n_loci, args[1], nh, args[3], args[4], g[0], g[1], g[2])


def test_device_collapse_one_sort_and_two_sorts(ctx, monkeypatch):
    """The pairs' order -- std::sort on (left end, right end), ties in input order -- comes from ONE device-wide sort whose
    key lays the clusters' ranges of left ends end to end, with the pair's length below; where that key does not fit 64
    bits (or SBGPU_COLLAPSE_TWO_SORTS=1) from round 4's two sorts.  Both routes on a sample with duplicates and ties, and a
    sample whose clusters reach across the whole 32-bit coordinate range with pairs as long (the key would need 65 bits: the
    library must take the two sorts by itself): the host form's unique hits either way."""
    from strawberry_amd import exonbin as eb
    rng = np.random.default_rng(41)
    loc, mass, left, right = [], [], [], []
    for l in range(12):
        base = 100000 * (l + 1)
        for s in rng.integers(base, base + 300, int(rng.integers(50, 1500))):
            s, d = int(s), int(rng.choice([150, 151, 180]))
            loc.append(l), mass.append(1.0), left.append([(s, s + 74)])
            right.append([(s + d, s + d + 74)] if rng.random() < 0.9 else [])
    args = (12, loc, mass, left, right)
    host = eb.collapse_pairs(*args)
    for env in ("0", "1"):
        monkeypatch.setenv("SBGPU_COLLAPSE_TWO_SORTS", env)
        g = eb.collapse_pairs(*args, device=ctx)
        same(g[0], host[0], g[1], host[1], g[2], host[2])
    monkeypatch.delenv("SBGPU_COLLAPSE_TWO_SORTS")
    # clusters across the whole coordinate range, pairs as long as the range
    far = 4200000000
    loc, mass, left, right = [], [], [], []
    for l in range(3):
        for k in range(40):
            a = 1000 + 7 * (k % 5) + l
            loc.append(l), mass.append(1.0), left.append([(a, a + 74)])
            right.append([(far + 3 * (k % 4), far + 3 * (k % 4) + 74)] if k % 2 else [(a + 200, a + 274)])
        loc.append(l), mass.append(1.0), left.append([(far - 500, far - 426)]), right.append([])
    args = (3, loc, mass, left, right)
    g, r = eb.collapse_pairs(*args, device=ctx), eb.collapse_pairs(*args)
    same(g[0], r[0], g[1], r[1], g[2], r[2])
    assert g[0].n_hits > 20


def test_device_collapse_many_small_clusters(ctx):
    """A pair's cluster is found by a search the WAVE makes 64 ways at a time (wave_range_of, csrc/device_common.h): cluster
    counts around the powers of 64, empty clusters in between, zero to three pairs each -- the host form's unique hits."""
    from strawberry_amd import exonbin as eb
    rng = np.random.default_rng(29)
    for n_loci in (1, 2, 63, 64, 65, 130, 4095, 4097, 9000):
        loc, mass, left, right = [], [], [], []
        for l in range(n_loci):
            for _ in range(int(rng.integers(0, 4)) if rng.random() < 0.8 else 0):
                a = 1000 + 5000 * l + int(rng.integers(0, 40))
                loc.append(l)
                mass.append(1.0 if rng.random() < 0.8 else 0.5)
                left.append([(a, a + 74)])
                right.append([(a + 200, a + 274)] if rng.random() < 0.8 else [])
        if not loc:
            loc, mass, left, right = [0], [1.0], [[(1000, 1074)]], [[]]
        args = (n_loci, loc, mass, left, right)
        g, r = eb.collapse_pairs(*args, device=ctx), eb.collapse_pairs(*args)
        same(g[0], r[0], g[1], r[1], g[2], r[2])


def test_device_collapse_mates_apart_abutting_and_overlapping(ctx):
    """The device form takes a short cut for mates that lie apart -- the hit's features are the left mate's, the GAP, the
    right mate's, as they are -- and merges the two lists otherwise (Contig(PairedHit), contig.cpp:216-267).  Spliced and
    unspliced mates at every distance around the boundary (a gap of 2, 1, 0 bases, abutting, overlapping by 1 ... 40): the
    host form's unique hits, its rejects included."""
    from strawberry_amd import exonbin as eb
    rng = np.random.default_rng(31)
    loc, mass, left, right = [], [], [], []
    for l in range(40):
        for k in range(int(rng.integers(20, 80))):
            a = 1000 + 20000 * l + int(rng.integers(0, 300))
            lm = [(a, a + 30), (a + 200, a + 243)] if rng.random() < 0.6 else [(a, a + 74)]
            d = int(rng.integers(-40, 200)) if k % 3 else int(rng.integers(-1, 4))     # (the right mate's start - the left mate's end)
            b = lm[-1][1] + d
            rm = [(b, b + 20), (b + 500, b + 553)] if rng.random() < 0.5 else ([(b, b + 74)] if rng.random() < 0.8 else [])
            loc.append(l), mass.append(1.0), left.append(lm), right.append(rm)
    args = (40, loc, mass, left, right)
    g, r = eb.collapse_pairs(*args, device=ctx), eb.collapse_pairs(*args)
    same(g[0], r[0], g[1], r[1], g[2], r[2])
    assert g[0].n_hits > 1000 and g[2]["rejected"] > 20


def test_device_collapse_big_loci(ctx, oracle):
    """Loci of more than 4096 pairs (the LDS sort's limit) take the same steps with their arrays in global memory
    (collapse_big_kernel): 4097 (just over), 20 000 and 70 000 pairs next to small loci, duplicates, NH masses whose sums
    depend on the order, single reads; identical to the host form, and the smaller big locus against the oracle."""
    from strawberry_amd import exonbin as eb
    rng = np.random.default_rng(321)
    sizes = [300, 4097, 50, 20000, 0, 70000, 700]
    loc, nh, left, right = [], [], [], []
    for l, n in enumerate(sizes):
        base = 1000000 * (l + 1)
        starts = rng.integers(base, base + max(400, n // 8), n)      # ~8 pairs per start: many duplicates
        for k in range(n):
            s0 = int(starts[k])
            lb = [(s0, s0 + 74)]
            if rng.random() < 0.2:
                cut = int(rng.integers(10, 60))
                lb = [(s0, s0 + cut - 1), (s0 + cut + 300, s0 + 300 + 74)]
            rb = [] if rng.random() < 0.1 else [(lb[-1][1] + 1 + int(rng.choice([120, 150])), lb[-1][1] + 75 + int(rng.choice([120, 150])))]
            if rb and rb[0][1] - rb[0][0] != 74:
                rb = [(rb[0][0], rb[0][0] + 74)]
            loc.append(l), nh.append(int(rng.choice([1, 2, 3]))), left.append(lb), right.append(rb)
    perm = rng.permutation(len(loc))
    args = (len(sizes), [loc[i] for i in perm], [1.0 / nh[i] for i in perm], [left[i] for i in perm], [right[i] for i in perm])
    g, r = eb.collapse_pairs(*args, device=ctx), eb.collapse_pairs(*args)
    same(g[0], r[0], g[1], r[1], g[2], r[2])
    assert r[0].n_hits < len(loc) * 0.8
    # the oracle on everything but the largest locus (it is plain C, but the Python around it is per pair)
    keep = [i for i in perm if loc[i] != 5]
    sub = (len(sizes), [loc[i] for i in keep], [1.0 / nh[i] for i in keep], [left[i] for i in keep], [right[i] for i in keep])
    g2 = eb.collapse_pairs(*sub, device=ctx)
    XU.check_collapse_against_oracle(oracle, len(sizes), sub[1], [nh[i] for i in keep], sub[3], sub[4], g2[0], g2[1], g2[2])


def test_device_collapse_declines_what_it_does_not_cover(ctx):
    """Round 4: a mate of 25 features (13 blocks) is served -- the limit of 24 was the per-cluster kernels'; the flat form's
    long-mate kernels go to 512 features, and only beyond that the call declines."""
    from strawberry_amd import _lib, exonbin as eb
    long_mate = [[(1000 + 100 * k, 1040 + 100 * k) for k in range(13)]]      # 25 features
    g, r = eb.collapse_pairs(1, [0], [1.0], long_mate, [[]], device=ctx), eb.collapse_pairs(1, [0], [1.0], long_mate, [[]])
    same(g[0], r[0], g[1], r[1], g[2], r[2])
    huge_mate = [[(1000 + 100 * k, 1040 + 100 * k) for k in range(300)]]     # 599 features
    with pytest.raises(_lib.SbgpuError, match="512 features"):
        eb.collapse_pairs(1, [0], [1.0], huge_mate, [[]], device=ctx)


def test_collapsed_hits_feed_the_chain_without_leaving_the_device(ctx):
    """pairs (device) -> sbgpu_collapse_pairs_device -> sbgpu_uniq_dev_hits -> sbgpu_quantify_device == the host chain on
    the host-collapsed hits."""
    import ctypes as C
    import torch
    from strawberry_amd import _lib, exonbin as eb
    from strawberry_amd.quantify import InsertSize, quantify_host
    d = U.E2E
    ordered, rows, _, _ = U.load(d)
    annot, hits, names, rejected = XU.e2e_inputs(d, ordered)
    copies = XU.load_read_copies(d)
    order = np.argsort([c[0] for c in copies], kind="stable")
    copies = [copies[i] for i in order]
    L = _lib.load()
    dev = torch.device("cuda", 0)
    def csr(blocks_list):
        off, c, l, r = [0], [], [], []
        for b in blocks_list:
            cc, ll, rr = eb.mate_features(b)
            c += cc; l += ll; r += rr
            off.append(len(c))
        return (np.asarray(off, np.int64), np.asarray(c, np.uint8), np.asarray(l, np.uint32), np.asarray(r, np.uint32))
    lo, lc, ll, lr = csr([c[1] for c in copies])
    ro, rc, rl, rr = csr([c[2] for c in copies])
    mass = np.asarray([c[3] for c in copies], np.float64)
    loc = np.asarray([c[0] for c in copies], np.int32)
    keep = [torch.from_numpy(x.view(np.int32) if x.dtype == np.uint32 else x).to(dev) if x.size else torch.zeros(1, dtype=torch.int64, device=dev)
            for x in (mass, lo, lc, ll, lr, ro, rc, rl, rr)]
    dp = _lib.sbgpu_pairs_t(len(loc), None, *[t.data_ptr() for t in keep])
    poff = np.searchsorted(loc, np.arange(len(names) + 1), side="left").astype(np.int64)
    h = C.c_void_p()
    _lib.check(L.sbgpu_collapse_pairs_device(ctx.h, len(names), C.byref(dp), poff.ctypes.data, None, C.byref(h)), "collapse")
    dh = _lib.sbgpu_hits_t()
    d_mass, hoff = C.c_void_p(), C.c_void_p()
    _lib.check(L.sbgpu_uniq_dev_hits(h, C.byref(dh), C.byref(d_mass), C.byref(hoff)), "uniq_dev_hits")
    n_iso = int(annot.iso_off[-1])
    theta = np.zeros(n_iso + 1); status = np.zeros(annot.n_loci + 1, np.int32); iters = np.zeros(annot.n_loci + 1, np.int32)
    ins = InsertSize(250.0, 30.0)._struct(75)
    an = annot._struct()
    bh = C.c_void_p()
    _lib.check(L.sbgpu_quantify_device(ctx.h, C.byref(an), C.byref(dh), d_mass, hoff, C.byref(ins), 75, 0, theta.ctypes.data,
                                       status.ctypes.data, iters.ctypes.data, C.byref(bh)), "sbgpu_quantify_device")
    L.sbgpu_bins_destroy(bh)
    L.sbgpu_uniq_dev_destroy(h)
    r = quantify_host(annot, hits, InsertSize(250.0, 30.0), 75, ctx=ctx)
    np.testing.assert_array_equal(theta[:n_iso], r["theta"])
    np.testing.assert_array_equal(iters[:annot.n_loci], r["iters"])


def test_flat_collapse_any_order_paths_equal_the_running_sums(ctx, oracle, monkeypatch):
    """The flat form decides the span filter from the spans' exact integer moments and, where every mass of a cluster is a
    multiple of 2^-20 (NH 1, 2, 4 ...), adds the masses in any order (hardware atomics); clusters where that is not safe
    take the reference's running sums.  Both routes, on the same input: masses that are dyadic in some clusters and not in
    others (NH 3), spans that are all equal in one cluster (sd 0: the filter sees 0 / 0), a cluster of 30 000 pairs with
    thousands of copies of one pair (one group's mass = thousands of additions) -- identical to the host form and to each
    other, and SBGPU_COLLAPSE_FORCE_SEQ=1 (every cluster on the running sums) gives the same bits."""
    from strawberry_amd import exonbin as eb
    rng = np.random.default_rng(99)
    sizes = [500, 30000, 0, 1200, 64, 65, 5000]
    nh_choices = [[1, 2, 4], [1], [1], [1, 3], [2], [1, 2, 3, 4], [1, 4]]
    loc, nh, left, right = [], [], [], []
    for l, n in enumerate(sizes):
        base = 1000000 * (l + 1)
        starts = rng.integers(base, base + max(50, n // 40), n)              # many copies per start
        for k in range(n):
            s0 = int(starts[k])
            lb = [(s0, s0 + 74)]
            if l != 4 and rng.random() < 0.25:                                # (locus 4: every span 75)
                cut = int(rng.integers(10, 60))
                gap = 30000 if rng.random() < 0.02 else 300                   # a few mates spanning 30 kb: the filter's outliers
                lb = [(s0, s0 + cut - 1), (s0 + cut + gap, s0 + cut + gap + 74 - cut)]
            rb = [(lb[-1][1] + 100, lb[-1][1] + 174)]
            loc.append(l), nh.append(int(rng.choice(nh_choices[l]))), left.append(lb), right.append(rb)
    perm = rng.permutation(len(loc))
    args = (len(sizes), [loc[i] for i in perm], [1.0 / nh[i] for i in perm], [left[i] for i in perm], [right[i] for i in perm])
    r = eb.collapse_pairs(*args)
    g = eb.collapse_pairs(*args, device=ctx)
    same(g[0], r[0], g[1], r[1], g[2], r[2])
    assert g[2]["filtered"] > 0 and r[0].n_hits < len(loc) * 0.5
    monkeypatch.setenv("SBGPU_COLLAPSE_FORCE_SEQ", "1")
    g2 = eb.collapse_pairs(*args, device=ctx)
    monkeypatch.delenv("SBGPU_COLLAPSE_FORCE_SEQ")
    same(g2[0], r[0], g2[1], r[1], g2[2], r[2])
    XU.check_collapse_against_oracle(oracle, len(sizes), args[1], [nh[i] for i in perm], args[3], args[4], g[0], g[1], g[2])


def test_device_collapse_long_mates(ctx, oracle):
    """Long reads: mates of 13 to 120 blocks (25 to 239 features; the main kernels keep a mate's 24 in registers), next to
    ordinary pairs, duplicates of the long ones included; overlapping long mates whose blocks merge, and one pair the merge
    rejects.  The device form (flat_heads_long_kernel / flat_fill_long_kernel) equals the host form and the oracle."""
    from strawberry_amd import exonbin as eb
    rng = np.random.default_rng(7)

    def blocks(start, n, ex=40, intron=100):
        return [(start + k * (ex + intron), start + k * (ex + intron) + ex - 1) for k in range(n)]

    loc, nh, left, right = [], [], [], []
    for l in range(3):
        base = 2_000_000 * (l + 1)
        for k in range(300):
            s0 = base + int(rng.integers(0, 60))
            kind = rng.random()
            if kind < 0.5:                                   # ordinary pair
                lb, rb = [(s0, s0 + 74)], [(s0 + 200, s0 + 274)]
            elif kind < 0.8:                                 # a long left mate, single read or with a short right mate far behind
                n = int(rng.choice([13, 30, 120]))
                lb = blocks(s0, n)
                rb = [] if rng.random() < 0.5 else [(lb[-1][1] + 500, lb[-1][1] + 574)]
            else:                                            # two long mates that overlap: the shared blocks merge
                n = int(rng.choice([20, 45]))
                lb = blocks(s0, n)
                rb = blocks(lb[n // 2][0], n)                # starts at one of the left mate's blocks: same exon grid
            loc.append(l), nh.append(int(rng.choice([1, 2]))), left.append(lb), right.append(rb)
    # overlapping long mates on DIFFERENT exon grids: two different introns meet -> Contig(PairedHit) rejects the pair
    loc.append(0), nh.append(1), left.append(blocks(2_000_000, 30)), right.append(blocks(2_000_000 + 17, 30, ex=40, intron=100)[:30])
    perm = rng.permutation(len(loc))
    args = (3, [loc[i] for i in perm], [1.0 / nh[i] for i in perm], [left[i] for i in perm], [right[i] for i in perm])
    r = eb.collapse_pairs(*args)
    g = eb.collapse_pairs(*args, device=ctx)
    same(g[0], r[0], g[1], r[1], g[2], r[2])
    nfeat = np.diff(g[0].feat_off)
    assert nfeat.max() > 200 and (nfeat > 48).sum() > 20 and g[2]["rejected"] >= 1
    XU.check_collapse_against_oracle(oracle, 3, args[1], [nh[i] for i in perm], args[3], args[4], g[0], g[1], g[2])
