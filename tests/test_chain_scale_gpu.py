"""The fragments -> abundances chain (sbgpu_quantify_device) at config scale against the ORACLE CHAIN, locus by locus.

6 000 loci / 2.2e7 read pairs of the chain workload's sample (strawberry_amd/chain.py::DeviceSample -- bench.py's
`c3-chain` at a tenth of its size) go through ONE sbgpu_quantify_device call; the same hits, brought to the host, go
through the oracle's restatements stage by stage:

  exonbin_oracle.c  (Contig::is_compatible / overlap_exons, contig.cpp:547-599, estimate.cpp:115-131)   -> words per hit
  a numpy restatement, in this file, of the grouping (assign_exon_bin, estimate.cpp:135-198: bins keyed by their segment
      set in first-appearance order, mass = sum over the bin's distinct fragments, truncated to int, estimate.cpp:288)
  e2e_util.bin_under_iso (isoform.h:363-411) + binweight_oracle.c (set_theory_bin_weight, estimate.cpp:201-234)   -> F
  em_oracle.c (EmSolver, estimate.cpp:366-488)                                                             -> theta

and every locus is compared: bins, keys, compat unions and counts EXACT; F <= 1e-12 relative; status and iteration
counts EXACT; theta <= 1e-9 relative.  Nothing on the checking side calls libsbgpu.so.  The host form of the product's
grouping (sbgpu_bins_create) is cross-checked against the same numpy grouping on the way.

Size: SB_CHAIN_TEST_LOCI / SB_CHAIN_TEST_FRAGS override the defaults (6 000 / 2.2e7; about 1.5 GB of host arrays)."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import e2e_util as U

pytestmark = pytest.mark.gpu

N_LOCI = int(os.environ.get("SB_CHAIN_TEST_LOCI", "6000"))
N_FRAGS = float(os.environ.get("SB_CHAIN_TEST_FRAGS", "2.2e7"))
RL, MEAN, SD = 75, 250.0, 30.0


def numpy_grouping(annot, hits, compat, key):
    """assign_exon_bin restated on the oracle's words (one compat / key word per hit: <= 32 isoforms and segments).
    -> row_off[n_loci + 1], bin_key[n_bins], bin_compat[n_bins], count[n_bins] (int32), hits in bins"""
    assert compat.shape[1] == 1 and key.shape[1] == 1
    compat, key = compat[:, 0], key[:, 0]
    # a fragment's identity: its feature list (ExonBin::_frags is a std::set<Contig>, isoform.h:133,267: equal feature
    # lists from different unique hits count once).  Two independent 64-bit sums over (position, code, left, right).
    with np.errstate(over="ignore"):
        pos = (np.arange(len(hits.feat_code), dtype=np.uint64) - np.repeat(hits.feat_off[:-1], np.diff(hits.feat_off)).astype(np.uint64))
        sig = []
        for a, b, c, d in ((0x9E3779B97F4A7C15, 0xC2B2AE3D27D4EB4F, 0x165667B19E3779F9, 0x27D4EB2F165667C5),
                           (0xD6E8FEB86659FD93, 0xA0761D6478BD642F, 0xE7037ED1A0B428DB, 0x8EBC6AF09C88C6E3)):
            v = (pos + np.uint64(1)) * np.uint64(a) ^ (hits.feat_code.astype(np.uint64) + np.uint64(7)) * np.uint64(b)
            v = (v ^ (v >> np.uint64(31))) * np.uint64(c) + hits.feat_left.astype(np.uint64) * np.uint64(d)
            v = (v ^ (v >> np.uint64(29))) * np.uint64(a) + hits.feat_right.astype(np.uint64) * np.uint64(b)
            v = v ^ (v >> np.uint64(32))
            sig.append(np.add.reduceat(v, hits.feat_off[:-1]) if hits.n_hits else np.zeros(0, np.uint64))
    off = np.searchsorted(hits.hit_locus, np.arange(annot.n_loci + 1), side="left")
    row_off = np.zeros(annot.n_loci + 1, np.int64)
    keys, compats, counts = [], [], []
    used = 0
    for l in range(annot.n_loci):
        c, k, m = compat[off[l]:off[l + 1]], key[off[l]:off[l + 1]], hits.mass[off[l]:off[l + 1]]
        ok = c != 0                                   # a hit that fits no isoform joins no bin (estimate.cpp:150-165)
        # of equal fragments the first one inserted stays (std::set::insert): equal fragments have equal words, hence one bin
        s2 = np.stack([sig[0][off[l]:off[l + 1]], sig[1][off[l]:off[l + 1]]], 1)
        _, first_of = np.unique(s2, axis=0, return_index=True)
        distinct = np.zeros(len(c), bool)
        distinct[first_of] = True
        used += int(ok.sum())
        ok &= distinct
        c, k, m = c[ok], k[ok], m[ok]
        uk, first, inv = np.unique(k, return_index=True, return_inverse=True)
        order = np.argsort(first, kind="stable")      # bins are numbered by first appearance (UniqPushAndReturnIdx, isoform.h:16-28)
        rank = np.empty(len(uk), np.int64)
        rank[order] = np.arange(len(uk))
        b = rank[inv]
        # the bin's mass: float sum over its distinct fragments (ExonBin::read_count, isoform.h:285-296), then (int)
        # (whole-number masses below 2^24: the float sum is exact in any order)
        mass = np.bincount(b, weights=m.astype(np.float64), minlength=len(uk))
        un = np.zeros(len(uk), np.uint32)
        np.bitwise_or.at(un, b, c)
        keys.append(uk[order])
        compats.append(un)
        counts.append(mass.astype(np.float32).astype(np.int32))
        row_off[l + 1] = row_off[l] + len(uk)
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)  # noqa: E731
    return row_off, cat(keys, np.uint32), cat(compats, np.uint32), cat(counts, np.int32), used


def oracle_weights(oracle, annot, row_off, bin_key, bin_compat, threads):
    """F of every locus from the bins: bin_under_iso + sbo_bin_weight per (bin, isoform) pair.  -> f_off, F"""
    ins = oracle.make_insert(MEAN, SD)
    niso = np.diff(annot.iso_off)
    f_off = np.concatenate([[0], np.cumsum(np.diff(row_off) * niso)]).astype(np.int64)
    F = np.zeros(int(f_off[-1]))
    n_pairs = [0]

    def locus(l):
        segs = [(int(a), int(b)) for a, b in annot.segments(l)]
        isos = []
        for j in range(int(annot.iso_off[l]), int(annot.iso_off[l + 1])):
            e0, e1 = int(annot.exon_off[j]), int(annot.exon_off[j + 1])
            ex = list(zip(annot.exon_left[e0:e1].tolist(), annot.exon_right[e0:e1].tolist()))
            isos.append((U.isoform_segments(segs, ex), sum(b - a + 1 for a, b in ex)))
        n = 0
        for r in range(int(row_off[l]), int(row_off[l + 1])):
            coords = [s for k, s in enumerate(segs) if (int(bin_key[r]) >> k) & 1]
            for j, (iso_segs, iso_len) in enumerate(isos):
                if not (int(bin_compat[r]) >> j) & 1:
                    continue
                got = U.bin_under_iso(coords, iso_segs)
                assert got is not None, (l, r, j)
                seg_lens, implicit = got
                F[f_off[l] + (r - row_off[l]) * len(isos) + j] = oracle.bin_weight(seg_lens, implicit, iso_len, RL, ins)
                n += 1
        return n

    with ThreadPoolExecutor(threads) as ex:
        n_pairs[0] = sum(ex.map(locus, range(annot.n_loci), chunksize=64))
    return f_off, F, n_pairs[0]


def test_chain_sample_every_locus_against_the_oracle_chain(oracle):
    from strawberry_amd import chain, em
    from strawberry_amd import exonbin as eb
    ctx = em.default_context(0)
    threads = min(32, os.cpu_count() or 1)
    from strawberry_amd.quantify import InsertSize, quantify_host
    with chain.ChainQuantifier(ctx, n_loci=N_LOCI, n_frags=N_FRAGS, seed=41, pin=True) as q:
        bins = q.step(keep=True)                      # sbgpu_quantify_device: hits resident in HBM
        theta, status, iters = q.theta[:q.n_iso].copy(), q.status[:q.n_loci].copy(), q.iters[:q.n_loci].copy()
        hits = q.hits.host_hits(q.n_loci)
        annot = q.annot
        n_frags = q.n_frags
        # the same hits through the host entry (same kernels; it also returns the words per hit and the weights, which the
        # device entry leaves in HBM): the two entries agree bit for bit
        r = quantify_host(annot, hits, InsertSize(MEAN, SD), RL, ctx=ctx)
    assert r["bins"].grouped_on_device, r["bins"].host_grouping_reason     # the host entry did not fall to the host grouping
    np.testing.assert_array_equal(r["theta"], theta)
    np.testing.assert_array_equal(r["status"], status)
    np.testing.assert_array_equal(r["iters"], iters)
    np.testing.assert_array_equal(r["bins"].count, bins.count)
    np.testing.assert_array_equal(r["bins"].bin_key, bins.bin_key)
    F = r["F"]
    assert annot.n_loci == N_LOCI >= 6000 or "SB_CHAIN_TEST_LOCI" in os.environ
    assert n_frags >= 2e7 or "SB_CHAIN_TEST_FRAGS" in os.environ
    assert annot.compat_words == 1 and annot.key_words == 1

    # ---- stage 1 + 2: words from the oracle, grouped by the numpy restatement; the device chain's bins must be these
    o_compat, o_key = oracle.exonbin_batch(annot, hits)
    np.testing.assert_array_equal(r["compat"], o_compat)          # every hit's compatibility word
    row_off, bin_key, bin_compat, count, used = numpy_grouping(annot, hits, o_compat, o_key)
    np.testing.assert_array_equal(bins.row_off, row_off)
    np.testing.assert_array_equal(bins.bin_key[:, 0], bin_key)
    np.testing.assert_array_equal(bins.bin_compat[:, 0], bin_compat)
    np.testing.assert_array_equal(bins.count, count)
    assert bins.n_hits_used == used and 0.85 < used / hits.n_hits < 0.999
    # (the product's host grouping on the oracle's words: the same bins -- it is what sbgpu_quantify_host falls back to)
    hb = eb.LocusBins(annot, hits, o_compat, o_key)
    np.testing.assert_array_equal(hb.row_off, row_off)
    np.testing.assert_array_equal(hb.bin_key[:, 0], bin_key)
    np.testing.assert_array_equal(hb.count, count)

    # ---- stage 3: weights
    f_off, o_F, n_pairs = oracle_weights(oracle, annot, row_off, bin_key, bin_compat, threads)
    np.testing.assert_array_equal(bins.f_off, f_off)
    assert n_pairs == bins.n_pairs
    assert ((F != 0) == (o_F != 0)).all()
    nz = o_F != 0
    err = np.abs(F[nz] - o_F[nz]) / o_F[nz]
    assert err.max() <= 1e-12, (err.max(), int(err.argmax()))

    # ---- stage 4: the EM on the oracle's own F
    o_theta, o_status, o_iters = oracle.em_batch(row_off, annot.iso_off, f_off, count, o_F, threads=threads)
    np.testing.assert_array_equal(status, o_status)
    np.testing.assert_array_equal(iters, o_iters)
    terr = np.abs(theta - o_theta) / np.maximum(np.abs(o_theta), 1e-9)
    assert terr.max() < 1e-9, (terr.max(), int(terr.argmax()))
    assert (o_status == 0).sum() > 0.5 * N_LOCI          # a real workload: most loci converge, some hit the cap, some are empty
    print("chain vs oracle chain: %d loci, %d read pairs in %d unique hits, %d bins, %d weights: F err %.2e, theta err %.2e, "
          "status/iters exact (%d capped, %d empty)" % (annot.n_loci, n_frags, hits.n_hits, len(count), n_pairs, err.max(),
                                                        terr.max(), int((o_status == 3).sum()), int((o_status == 1).sum())))


def test_stale_pin_is_not_served():
    """An annotation pinned with the context, then its arrays rewritten IN PLACE (what address recycling after a free looks
    like to the library): the call must notice (sampled fingerprint) and take the ordinary path -- same results as a
    context that never pinned -- instead of serving the stale device copies."""
    from strawberry_amd import chain, em
    from strawberry_amd.quantify import InsertSize, quantify_host
    ctx = em.default_context(0)
    with chain.ChainQuantifier(ctx, n_loci=300, n_frags=300 * 200, seed=9, pin=True) as q:
        q.step()
        hits = q.hits.host_hits(q.n_loci)
        a = q.annot
        want = quantify_host(a, hits, InsertSize(MEAN, SD), RL, ctx=ctx)["theta"].copy()     # (pinned annotation: resident path)
        np.testing.assert_array_equal(want, q.theta[:q.n_iso])
        # shift the genome by 1000 bases in place: annotation and hits move together, so the answer is unchanged
        for arr in (a.exon_left, a.exon_right, a.seg_left, a.seg_right, hits.feat_left, hits.feat_right):
            arr += 1000
        got = quantify_host(a, hits, InsertSize(MEAN, SD), RL, ctx=ctx)["theta"]
        np.testing.assert_array_equal(got, want)       # stale copies would fit no hit: theta would be all init-empty
        for arr in (a.exon_left, a.exon_right, a.seg_left, a.seg_right):
            arr -= 1000
