"""GPU tests of sbgpu_pair_mates_device (HitCluster::addOpenHit + addHit on the GPU, /root/reference/src/alignments.cpp:
423-655, csrc/matepair_device.h) against the oracle (oracle/matepair_oracle.c, pinned to the reference's own HitCluster by
tests/test_matepair_oracle.py) -- pairs in completion order, mates' features, masses, counts -- and the front of the
path end to end ON THE DEVICE: alignment records -> pairs -> unique hits (sbgpu_collapse_pairs_device) must be the
unique hits of the reference binary's toy runs, bit for bit."""
import ctypes as C

import numpy as np
import pytest

import e2e_util as U
import exonbin_util as XU
import matepair_util as MU

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from strawberry_amd import em
    return em.default_context(0)


def check_against_oracle(oracle, clusters, got):
    from strawberry_amd import exonbin as eb
    at = 0
    tot = {"complete": 0, "single": 0, "refused": 0, "orphan": 0}
    for l, c in enumerate(clusters):
        ids, blocks, ppos, flags, nh = MU.arrays(c)
        lr, rr, m, cnt = oracle.pair_mates(ids, blocks, ppos, flags, nh)
        for k in tot:
            tot[k] += cnt[k]
        assert got["pair_off"][l + 1] - got["pair_off"][l] == len(lr), l
        for i, j, mm in zip(lr, rr, m):
            for side, rec in (("left", i), ("right", j)):
                s = slice(int(got[side + "_off"][at]), int(got[side + "_off"][at + 1]))
                code, fl, fr = (x[s] for x in got[side])
                want = eb.mate_features(blocks[rec]) if rec >= 0 else ([], [], [])
                assert ([int(x) for x in code], [int(x) for x in fl], [int(x) for x in fr]) == tuple(list(x) for x in want), (l, at, side)
            assert got["mass"][at] == mm
            at += 1
    assert at == got["info"]["pairs"]
    assert {k: got["info"][k] for k in tot} == tot


def test_device_pairing_equals_oracle_random_clusters(ctx, oracle):
    from strawberry_amd import exonbin as eb
    rng = np.random.default_rng(11)
    for trial in range(3):
        n_loci = int(rng.integers(2, 30))
        clusters = [MU.random_cluster(rng, int(rng.integers(0, 1500 if l == 1 else 200)), base=300000 * (l + 1)) for l in range(n_loci)]
        loc = [l for l, c in enumerate(clusters) for _ in c]
        reads = eb.Reads(loc, *MU.arrays([r for c in clusters for r in c]))
        got = eb.pair_mates(n_loci, reads, device=ctx)
        assert got["info"]["on_device"]
        check_against_oracle(oracle, clusters, got)
        host = eb.pair_mates(n_loci, reads)
        for k in ("pair_off", "mass", "left_off", "right_off"):
            np.testing.assert_array_equal(got[k], host[k], err_msg=k)
        for side in ("left", "right"):
            for x, y in zip(got[side], host[side]):
                np.testing.assert_array_equal(x, y)


def test_device_pairing_many_small_clusters(ctx):
    """The kernels find a record's / a pair's cluster with a search the WAVE makes 64 ways at a time (wave_range_of,
    csrc/device_common.h), and the pairing's sort key holds a GROUP of clusters in its upper bits: cluster counts around the
    powers of 64 and of two, with empty clusters in between, one record to a few pairs each -- the pairs must be the host form's
    (tests/test_collapse_gpu.py has the same for the unique hits)."""
    from strawberry_amd import exonbin as eb
    rng = np.random.default_rng(23)
    for n_loci in (1, 2, 63, 64, 65, 130, 4095, 4097, 9000):
        clusters = [MU.random_cluster(rng, int(rng.integers(0, 4)) if rng.random() < 0.8 else 0, base=1000 + 5000 * l, exotic=False)
                    for l in range(n_loci)]
        if not any(clusters):
            clusters[0] = MU.random_cluster(rng, 2, base=1000, exotic=False)
        loc = [l for l, c in enumerate(clusters) for _ in c]
        reads = eb.Reads(loc, *MU.arrays([r for c in clusters for r in c]))
        got, host = eb.pair_mates(n_loci, reads, device=ctx), eb.pair_mates(n_loci, reads)
        assert got["info"] == dict(host["info"], on_device=True), n_loci
        for k in ("pair_off", "mass", "left_off", "right_off"):
            np.testing.assert_array_equal(got[k], host[k], err_msg="%s %d" % (k, n_loci))
        for side in ("left", "right"):
            for x, y in zip(got[side], host[side]):
                np.testing.assert_array_equal(x, y)


def test_device_pairing_big_clusters(ctx, oracle):
    """Clusters of more than 8192 records (the LDS sort's limit) take the same steps with 1024 threads and the sort's
    arrays in global memory (matepair_big_kernel): 8193 single reads (just over), and random clusters of 6 000 and
    25 000 fragments (multi-mapped reads, orphans, refused records) next to small ones: identical to the host form; the
    6 000-fragment cluster (> 8192 records) also against the oracle."""
    from strawberry_amd import exonbin as eb
    n = 8193
    reads = eb.Reads([0] * n, list(range(1, n + 1)), [[(1000 + k, 1074 + k)] for k in range(n)], [0] * n, [0] * n, [1] * n)
    got, host = eb.pair_mates(1, reads, device=ctx), eb.pair_mates(1, reads)
    assert got["info"]["pairs"] == n and got["info"]["single"] == n
    np.testing.assert_array_equal(got["left_off"], host["left_off"])
    rng = np.random.default_rng(77)
    sizes = [150, 6000, 0, 25000, 40]
    clusters = [MU.random_cluster(rng, k, base=400000 * (l + 1)) for l, k in enumerate(sizes)]
    assert len(clusters[1]) > 8192 and len(clusters[3]) > 8192
    loc = [l for l, c in enumerate(clusters) for _ in c]
    reads = eb.Reads(loc, *MU.arrays([r for c in clusters for r in c]))
    got, host = eb.pair_mates(len(sizes), reads, device=ctx), eb.pair_mates(len(sizes), reads)
    assert got["info"]["on_device"]
    for k in ("pair_off", "mass", "left_off", "right_off"):
        np.testing.assert_array_equal(got[k], host[k], err_msg=k)
    for side in ("left", "right"):
        for x, y in zip(got[side], host[side]):
            np.testing.assert_array_equal(x, y)
    assert {k: got["info"][k] for k in ("complete", "single", "refused", "orphan")} == {k: host["info"][k] for k in ("complete", "single", "refused", "orphan")}
    sub = clusters[:3]
    loc = [l for l, c in enumerate(sub) for _ in c]
    got3 = eb.pair_mates(3, eb.Reads(loc, *MU.arrays([r for c in sub for r in c])), device=ctx)
    check_against_oracle(oracle, sub, got3)


@pytest.mark.parametrize("which", ["E2E", "E2E_MASS", "E2E_MINUS", "E2E_CHROMS"])
def test_records_to_unique_hits_on_the_device_equal_reference_runs(ctx, which):
    """Every sequenced copy of a toy run as two alignment records in BAM order -> sbgpu_pair_mates_device ->
    sbgpu_collapse_pairs_device: the unique hits (features, masses) and the mapped-read total of the reference run."""
    import torch
    from strawberry_amd import _lib, exonbin as eb
    d = getattr(U, which)
    ordered, rows, _, _ = U.load(d)
    annot, hits, names, rejected = XU.e2e_inputs(d, ordered)
    z = dict(np.load(__import__("os").path.join(d, "reads.npz")))
    genes = list(U.parse_annotation(__import__("os").path.join(d, "toy.gtf")))
    strands = U.gene_strands(d)
    locus_of = XU.locus_of_gene_index(d, names)
    recs, rid = [], 0
    for k in range(len(z["gene"])):
        gi = int(z["gene"][k])
        xs = 1 if strands[genes[gi]] == "+" else 2
        left = [(int(a), int(b)) for a, b in zip(z["left_l"][z["left_off"][k]:z["left_off"][k + 1]], z["left_r"][z["left_off"][k]:z["left_off"][k + 1]])]
        right = [(int(a), int(b)) for a, b in zip(z["right_l"][z["right_off"][k]:z["right_off"][k + 1]], z["right_r"][z["right_off"][k]:z["right_off"][k + 1]])]
        for nh in z["nh"][z["nh_off"][k]:z["nh_off"][k + 1]]:
            rid += 1
            recs.append((locus_of[gi], left[0][0], {"id": rid, "blocks": left, "ppos": right[0][0], "flags": xs << 2, "nh": int(nh)}))
            recs.append((locus_of[gi], right[0][0], {"id": rid, "blocks": right, "ppos": left[0][0], "flags": 1 | (xs << 2), "nh": int(nh)}))
    # the BAM's order: (chromosome, position), stable.  Which cluster a record joins is NOT taken from the simulation:
    # the device's read stream (sbgpu_assign_reads_device: Sample::nextClusterRefDemand's pass over the genes of the
    # annotation) must find it
    chroms = U.gene_chroms(d)
    chrom_id = {c: i for i, c in enumerate(sorted(set(chroms.values())))}
    gene_of_locus = names
    ref_of = [chrom_id[chroms[g]] for g in gene_of_locus]
    recs = [(ref_of[r[0]], r[1], r[0], r[2]) for r in recs]
    recs.sort(key=lambda r: (r[0], r[1]))
    c_left = [min(e[0] for _, ex in ordered[g] for e in ex) for g in gene_of_locus]
    c_right = [max(e[1] for _, ex in ordered[g] for e in ex) for g in gene_of_locus]
    c_strand = [1 if strands[g] == "+" else 2 for g in gene_of_locus]
    assert all((ref_of[k], c_left[k]) <= (ref_of[k + 1], c_left[k + 1]) for k in range(len(names) - 1))   # the reference's cluster order
    r_right = [r[3]["blocks"][-1][1] for r in recs]
    got_cluster, stream_off, stream_flags = eb.assign_reads(ref_of, c_left, c_right, c_strand, [r[0] for r in recs], [r[1] for r in recs],
                                                            r_right, [r[3]["flags"] for r in recs], device=ctx)
    np.testing.assert_array_equal(got_cluster, [r[2] for r in recs])          # every record reaches its gene's cluster
    for r, f in zip(recs, stream_flags):
        r[3]["flags"] = int(f)
    reads = eb.Reads(got_cluster, *MU.arrays([r[3] for r in recs]))
    L = _lib.load()
    dev = torch.device("cuda", 0)
    def up(x):
        x = x.view(np.int32) if x.dtype == np.uint32 else (x.view(np.int64) if x.dtype == np.uint64 else x)
        return torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    keep = [up(x) for x in (reads.read_id, reads.block_off, reads.block_left, reads.block_right, reads.partner_pos, reads.flags, reads.nh)]
    rs = _lib.sbgpu_reads_t(reads.n_reads, *[t.data_ptr() for t in keep])
    off = np.ascontiguousarray(stream_off, np.int64)       # the ranges the stream offered to the clusters
    assert off[-1] == reads.n_reads
    mh = C.c_void_p()
    _lib.check(L.sbgpu_pair_mates_device(ctx.h, len(names), C.byref(rs), off.ctypes.data, None, C.byref(mh)), "sbgpu_pair_mates_device")
    dp = _lib.sbgpu_pairs_t()
    poff = C.c_void_p()
    _lib.check(L.sbgpu_matepairs_pairs(mh, C.byref(dp), C.byref(poff)), "sbgpu_matepairs_pairs")
    uh = C.c_void_p()
    _lib.check(L.sbgpu_collapse_pairs_device(ctx.h, len(names), C.byref(dp), poff, None, C.byref(uh)), "sbgpu_collapse_pairs_device")
    info = (C.c_int64 * 8)()
    _lib.check(L.sbgpu_uniq_dev_info(uh, info), "sbgpu_uniq_dev_info")
    nh_, nf_ = int(info[0]), int(info[1])
    hl, fo = np.zeros(nh_, np.int32), np.zeros(nh_ + 1, np.int64)
    fc, fl, fr, ms = np.zeros(nf_, np.uint8), np.zeros(nf_, np.uint32), np.zeros(nf_, np.uint32), np.zeros(nh_, np.float32)
    cm = np.zeros(len(names))
    _lib.check(L.sbgpu_uniq_dev_export(uh, hl.ctypes.data, fo.ctypes.data, fc.ctypes.data, fl.ctypes.data, fr.ctypes.data, ms.ctypes.data,
                                       cm.ctypes.data), "sbgpu_uniq_dev_export")
    L.sbgpu_uniq_dev_destroy(uh)
    L.sbgpu_matepairs_destroy(mh)
    np.testing.assert_array_equal(hl, hits.hit_locus)
    np.testing.assert_array_equal(fo, hits.feat_off)
    np.testing.assert_array_equal(fc, hits.feat_code)
    np.testing.assert_array_equal(fl, hits.feat_left)
    np.testing.assert_array_equal(fr, hits.feat_right)
    np.testing.assert_array_equal(ms, hits.mass)
    assert int(info[4]) == rows[0]["total_mapped"] == hits.total_mapped


def test_device_read_stream_equals_oracle(ctx, oracle):
    from strawberry_amd import exonbin as eb
    from test_matepair_oracle import stream_case
    rng = np.random.default_rng(4)
    for trial in range(6):
        case = stream_case(rng, int(rng.integers(1, 400)), int(rng.integers(0, 200000)), refs=3)
        want, woff = oracle.assign_reads(*case)
        flags = [x << 2 for x in case[7]]
        got, off, fl = eb.assign_reads(*case[:7], flags, device=ctx)
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(off, woff)
        np.testing.assert_array_equal((fl & 16) != 0, want < 0)


def test_device_pairing_many_waiting_mates_and_long_groups(ctx, oracle):
    """A multi-mapped read with 12 and one with 40 left mates that ALL arrive before the first right mate (the
    per-cluster kernels kept at most 8 waiting mates of a read id and declined beyond; the flat form keeps a register mask
    for a group's first 64 records and state bytes behind it), and one read id with 150 alignments in the cluster (its
    group runs past the mask): every pair, in the reference's completion order -- equal to the oracle and the host form."""
    from strawberry_amd import exonbin as eb
    rng = np.random.default_rng(5)
    recs = []
    base = 500000
    for rid, n_aln in ((11, 12), (22, 40), (33, 150)):
        lefts = base + rid * 100000 + np.sort(rng.permutation(50000)[:n_aln])
        for s0 in lefts.tolist():
            recs.append({"id": rid, "blocks": [(s0, s0 + 74)], "ppos": s0 + 60000, "flags": (1 << 2), "nh": n_aln})
        for s0 in lefts.tolist():                          # the right mates arrive later, in the same order: oldest waiting mate first
            recs.append({"id": rid, "blocks": [(s0 + 60000, s0 + 60074)], "ppos": s0, "flags": MU.REVERSE | (1 << 2), "nh": n_aln})
    recs += MU.random_cluster(rng, 300, base=base + 50)
    order = np.argsort([r["blocks"][0][0] for r in recs], kind="stable")
    cluster = [recs[i] for i in order]
    reads = eb.Reads([0] * len(cluster), *MU.arrays(cluster))
    got, host = eb.pair_mates(1, reads, device=ctx), eb.pair_mates(1, reads)
    assert got["info"]["on_device"] and got["info"]["complete"] >= 12 + 40 + 150
    for k in ("pair_off", "mass", "left_off", "right_off"):
        np.testing.assert_array_equal(got[k], host[k], err_msg=k)
    for side in ("left", "right"):
        for x, y in zip(got[side], host[side]):
            np.testing.assert_array_equal(x, y)
    check_against_oracle(oracle, [cluster], got)


def _assert_same_pairs(got, host):
    for k in ("pair_off", "mass", "left_off", "right_off"):
        np.testing.assert_array_equal(got[k], host[k], err_msg=k)
    for side in ("left", "right"):
        for x, y in zip(got[side], host[side]):
            np.testing.assert_array_equal(x, y)
    assert {k: got["info"][k] for k in ("pairs", "complete", "single", "refused", "orphan")} == {k: host["info"][k] for k in ("pairs", "complete", "single", "refused", "orphan")}


def test_positional_and_sorted_forms_agree(ctx, oracle, monkeypatch):
    """Round 6: where a cluster's records ascend by position and no read id has two fitting mates, no sort is needed -- a closing
    mate searches its cluster for the records that start at its partner's position (the reference's rule is positional,
    alignments.cpp:612-615).  Random clusters (single reads, orphans, mates elsewhere, strands that disagree, partners at the
    read's own position, reads aligned at several PLACES under one id, PCR duplicates, a record beyond kMaxFragSpan) are served
    by the positional form; SBGPU_PAIR_FORCE_SORT=1 sends the same call through the sorted form: same pairs, same order, same
    counts, equal to the oracle and the host form."""
    from strawberry_amd import exonbin as eb
    rng = np.random.default_rng(101)
    for trial in range(3):
        n_loci = int(rng.integers(2, 40))
        clusters = [MU.random_cluster(rng, int(rng.integers(0, 3000 if l == 2 else 300)), base=400000 * (l + 1)) for l in range(n_loci)]
        loc = [l for l, c in enumerate(clusters) for _ in c]
        reads = eb.Reads(loc, *MU.arrays([r for c in clusters for r in c]))
        monkeypatch.delenv("SBGPU_PAIR_FORCE_SORT", raising=False)
        pos = eb.pair_mates(n_loci, reads, device=ctx)
        assert pos["positional"] and pos["why_sorted"] == 0
        monkeypatch.setenv("SBGPU_PAIR_FORCE_SORT", "1")
        srt = eb.pair_mates(n_loci, reads, device=ctx)
        monkeypatch.delenv("SBGPU_PAIR_FORCE_SORT")
        assert not srt["positional"] and srt["why_sorted"] == 0
        _assert_same_pairs(pos, srt)
        _assert_same_pairs(pos, eb.pair_mates(n_loci, reads))
        check_against_oracle(oracle, clusters, pos)


def test_positional_form_steps_aside(ctx, oracle):
    """What the positional form does not decide, it hands to the sorted form, and says why (sbgpu_matepairs_info[7]):
    a read id aligned twice at the SAME place (two openers fit one closing mate; the reference takes the oldest waiting one,
    alignments.cpp:593-641 -- the chain's order), one opener that two closing mates fit, records that do not ascend by position.
    Every case: the oracle's pairs, in its order.  5 000 records starting at ONE position are the positional form's own (a run
    of the arrival order is a hash table of its openers: a closing mate does not look through the run)."""
    from strawberry_amd import exonbin as eb
    rng = np.random.default_rng(202)

    def run(cluster, why):
        reads = eb.Reads([0] * len(cluster), *MU.arrays(cluster))
        got = eb.pair_mates(1, reads, device=ctx)
        assert not got["positional"] and got["why_sorted"] & why, (got["positional"], got["why_sorted"], why)
        _assert_same_pairs(got, eb.pair_mates(1, reads))
        check_against_oracle(oracle, [cluster], got)

    def sort_by_pos(recs):
        order = np.argsort([r["blocks"][0][0] for r in recs], kind="stable")
        return [recs[i] for i in order]

    bg = MU.random_cluster(rng, 200, base=700000)
    L = lambda rid, s, p, xs=1: {"id": rid, "blocks": [(s, s + 74)], "ppos": p, "flags": (xs << 2), "nh": 2}   # noqa: E731
    R = lambda rid, s, p, xs=1: {"id": rid, "blocks": [(s, s + 74)], "ppos": p, "flags": MU.REVERSE | (xs << 2), "nh": 2}   # noqa: E731
    # two alignments of read 5 at the same place: left, left, right, right
    run(sort_by_pos(bg + [L(5, 650000, 650200), L(5, 650000, 650200), R(5, 650200, 650000), R(5, 650200, 650000)]), 2)
    # one opener, two closing mates (the second finds it taken)
    run(sort_by_pos(bg + [L(6, 651000, 651200), R(6, 651200, 651000), R(6, 651200, 651000)]), 2)
    # two openers, one closing mate; the first opener's strand does not agree with the closer's, the second's does
    run(sort_by_pos(bg + [L(7, 652000, 652200, xs=2), L(7, 652000, 652200, xs=1), L(7, 652000, 652200, xs=1), R(7, 652200, 652000, xs=1)]), 2)
    # records out of position order
    c = sort_by_pos(list(bg))
    k = next(i for i in range(len(c) - 1) if c[i]["blocks"][0][0] != c[i + 1]["blocks"][0][0])
    c[k], c[k + 1] = c[k + 1], c[k]
    run(c, 1)
    # 5 000 records start at one position, their mates 300 bases on
    deep = []
    for i in range(5000):
        deep += [dict(L(1000 + i, 660000, 660300), nh=1), dict(R(1000 + i, 660300, 660000), nh=1)]
    cluster = sort_by_pos(bg + deep)
    reads = eb.Reads([0] * len(cluster), *MU.arrays(cluster))
    got = eb.pair_mates(1, reads, device=ctx)
    assert got["positional"] and got["info"]["complete"] >= 5000
    _assert_same_pairs(got, eb.pair_mates(1, reads))
    check_against_oracle(oracle, [cluster], got)
