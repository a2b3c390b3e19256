#!/usr/bin/env python3
"""tests/golden/exonbin_cases.npz: inputs of our own making (strawberry_amd.synth gene models and
read pairs) with the answers of the REFERENCE's own code, called through oracle/_ref
(oracle/ref_shim.cpp): Contig::Contig(PairedHit) for the feature lists, Contig::is_compatible for
every (hit, isoform), GenomicFeature::overlaps for every (hit, segment).  Run in the build
container (needs /root/reference); the .npz is the committed fixture."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import RefLib, build  # noqa: E402
from strawberry_amd import synth  # noqa: E402
from strawberry_amd.exonbin import Annotation  # noqa: E402  (host code only: segments)


def main():
    build(with_ref=True)
    ref = RefLib()
    loci = synth.make_gene_models(40, seed=2024, max_exons=14, max_isoforms=7)
    # one locus wider than a 32-bit word in both directions
    wide = synth.make_gene_models(1, seed=5, max_exons=45, max_isoforms=45, ex_lo=80, ex_hi=200)
    while len(wide[0]) < 34 or len({e for iso in wide[0] for e in iso}) < 34:
        wide = synth.make_gene_models(1, seed=int(np.random.default_rng(len(wide[0])).integers(1 << 30)), max_exons=45,
                                      max_isoforms=45, ex_lo=80, ex_hi=200)
    shift = loci[-1][-1][-1][1] + 10000
    loci.append([[(a + shift, b + shift) for a, b in iso] for iso in wide[0]])
    # short exons + long reads: fragments of many blocks (more than the kernel keeps in registers)
    n_norm = len(loci)
    shift = loci[-1][-1][-1][1] + 10000
    for isos in synth.make_gene_models(10, seed=99, max_exons=30, max_isoforms=5, ex_lo=20, ex_hi=60):
        loci.append([[(a + shift, b + shift) for a, b in iso] for iso in isos])
    hit_locus, pairs = synth.make_fragments(loci[:n_norm], 40, seed=77, noise=0.3, single=0.1)
    # mates that overlap or abut (merge_genomicFeats path; abutting pairs are rejected)
    hl2, p2 = synth.make_fragments(loci[:n_norm], 30, seed=78, mean=150.0, sd=12.0, noise=0.2, single=0.0)
    hl3, p3 = synth.make_fragments(loci[n_norm:], 60, seed=79, read_len=150, mean=420.0, sd=60.0, noise=0.2, single=0.1)
    hit_locus = hit_locus + hl2 + [n_norm + x for x in hl3]
    pairs = pairs + p2 + p3
    annot = Annotation(loci)

    feats, keep_locus, lb_off, lb, rb_off, rb, n_feat_ref = [], [], [0], [], [0], [], []
    for loc, (left, right) in zip(hit_locus, pairs):
        f = ref.pairedhit_features(left, right)
        lb += left
        lb_off.append(len(lb))
        rb += right
        rb_off.append(len(rb))
        n_feat_ref.append(0 if f is None else len(f[0]))
        if f is None:
            continue
        feats.append(f)
        keep_locus.append(loc)
    cw, kw = annot.compat_words, annot.key_words
    compat = np.zeros((len(feats), cw), np.uint32)
    key = np.zeros((len(feats), kw), np.uint32)
    for h, (loc, f) in enumerate(zip(keep_locus, feats)):
        for j, iso in enumerate(loci[loc]):
            if ref.is_compatible(f[0], f[1], f[2], [a for a, _ in iso], [b for _, b in iso]):
                compat[h, j >> 5] |= np.uint32(1 << (j & 31))
        segs = annot.segments(loc)
        k = ref.overlap_key(f[0], f[1], f[2], [a for a, _ in segs], [b for _, b in segs])
        for b in np.nonzero(k)[0]:
            key[h, int(b) >> 5] |= np.uint32(1 << (int(b) & 31))
    feat_off = np.concatenate([[0], np.cumsum([len(f[0]) for f in feats])]).astype(np.int64)
    out = os.path.join(ROOT, "tests", "golden", "exonbin_cases.npz")
    np.savez_compressed(
        out,
        iso_off=annot.iso_off, exon_off=annot.exon_off, exon_left=annot.exon_left, exon_right=annot.exon_right,
        # the raw pairs (for sbgpu_hit_features) and how many features the reference made of each (0 = rejected)
        pair_locus=np.asarray(hit_locus, np.int32), left_off=np.asarray(lb_off, np.int64),
        left_l=np.asarray([a for a, _ in lb], np.uint32), left_r=np.asarray([b for _, b in lb], np.uint32),
        right_off=np.asarray(rb_off, np.int64), right_l=np.asarray([a for a, _ in rb], np.uint32),
        right_r=np.asarray([b for _, b in rb], np.uint32), pair_n_feat=np.asarray(n_feat_ref, np.int32),
        # the reference's feature lists and answers
        hit_locus=np.asarray(keep_locus, np.int32), feat_off=feat_off,
        feat_code=np.asarray([c for f in feats for c in f[0]], np.uint8),
        feat_left=np.asarray([c for f in feats for c in f[1]], np.uint32),
        feat_right=np.asarray([c for f in feats for c in f[2]], np.uint32), compat=compat, key=key)
    print(out, os.path.getsize(out), "bytes;", len(feats), "hits of", len(pairs), "pairs;",
          "compat words", cw, "key words", kw, "; compatible with >=1 isoform:", int((compat != 0).any(1).sum()),
          "; max features", int(np.diff(feat_off).max()))


if __name__ == "__main__":
    main()
