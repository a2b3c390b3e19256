#!/usr/bin/env python3
"""GPU box: random stress of the exon-bin path: kernel words vs the oracle, device grouping + pairs vs the host
code, over many seeds of gene models and read pairs (noise, single reads, overlapping mates, long reads)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from strawberry_amd import em, synth
from strawberry_amd import exonbin as eb
from strawberry_amd.quantify import InsertSize, LocusQuantifier
from oracle import OracleLib
o = OracleLib()
ctx = em.default_context(0)
bad = tot = 0
t0 = time.time()
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    rng = np.random.default_rng(seed)
    kw = dict(max_exons=int(rng.integers(2, 40)), max_isoforms=int(rng.integers(1, 45)), ex_lo=int(rng.integers(15, 80)),
              ex_hi=int(rng.integers(90, 500)))
    loci = synth.make_gene_models(int(rng.integers(20, 300)), seed=100 + seed, **kw)
    rl = int(rng.choice([50, 75, 150, 400]))
    hl, pairs = synth.make_fragments(loci, int(rng.integers(20, 300)), seed=200 + seed, read_len=rl, mean=float(rng.choice([160, 250, 400])),
                                     sd=float(rng.choice([15, 40, 80])), noise=float(rng.random() * 0.5), single=float(rng.random() * 0.3))
    rows = [(l, eb.hit_features(lb, rb)) for l, (lb, rb) in zip(hl, pairs)]
    rows = [(l, f) for l, f in rows if f is not None]
    annot = eb.Annotation(loci)
    hits = eb.Hits([l for l, _ in rows], [f for _, f in rows], mass=rng.integers(1, 4, len(rows)).astype(np.float32))
    qd = LocusQuantifier(annot, hits, InsertSize(250.0, 30.0), rl, ctx=ctx, device_bins=True)
    qh = LocusQuantifier(annot, hits, InsertSize(250.0, 30.0), rl, ctx=ctx, device_bins=False)
    bd, bh = qd.assign_bins(), qh.assign_bins()
    oc, ok = o.exonbin_batch(annot, hits)
    words = (qd.d_compat.cpu().numpy().view(np.uint32)[:hits.n_hits] == oc).all() and (qd.d_key.cpu().numpy().view(np.uint32)[:hits.n_hits] == ok).all()
    same = qd.bins_on_device
    if same:
        for x in ("row_off", "f_off", "count", "bin_key", "bin_compat", "pair_seg_off", "pair_seg_lens", "pair_implicit_mask",
                  "pair_iso_len", "pair_out_index"):
            same = same and np.array_equal(getattr(bd, x), getattr(bh, x))
        same = same and np.array_equal(np.asarray(bd.hit_bin), bh.hit_bin)
    print("seed %d: %d loci %d hits cw %d kw %d max feats %d  words %s  device grouping %s  bins %d pairs %d" % (
        seed, annot.n_loci, hits.n_hits, annot.compat_words, annot.key_words, int(np.diff(hits.feat_off).max()),
        "ok" if words else "MISMATCH", ("ok" if same else "MISMATCH") if qd.bins_on_device else "declined", bh.n_bins, bh.n_pairs), flush=True)
    tot += hits.n_hits
    bad += (not words) + (qd.bins_on_device and not same)
print("total %d hits, %d failures, %.1f s" % (tot, bad, time.time() - t0))
sys.exit(1 if bad else 0)
