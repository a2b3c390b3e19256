"""Test helpers for the exon-bin path (A5): build the kernel's inputs from the e2e fixtures and
restate, through the ORACLE (tests only), the words the kernel must produce."""
import os

import numpy as np

from strawberry_amd import exonbin as eb


def load_reads(directory):
    """reads.npz (our simulated fragments, tools/make_e2e_golden.py) -> [(gene index, left blocks, right blocks)]."""
    z = dict(np.load(os.path.join(directory, "reads.npz")))
    out = []
    nh_off, nh = z["nh_off"], z["nh"]
    for k in range(len(z["gene"])):
        lb = [(int(a), int(b)) for a, b in zip(z["left_l"][z["left_off"][k]:z["left_off"][k + 1]],
                                               z["left_r"][z["left_off"][k]:z["left_off"][k + 1]])]
        rb = [(int(a), int(b)) for a, b in zip(z["right_l"][z["right_off"][k]:z["right_off"][k + 1]],
                                               z["right_r"][z["right_off"][k]:z["right_off"][k + 1]])]
        # PairedHit mass = 0.5/NH per mate (src/read.cpp:49-53,734-741), collapsed over the copies of the
        # fragment (src/alignments.cpp:683-696), all in double
        mass = 0.0
        for n in nh[nh_off[k]:nh_off[k + 1]]:
            mass += 0.5 / int(n) + 0.5 / int(n)
        out.append((int(z["gene"][k]), lb, rb, mass))
    return out


def load_read_copies(directory):
    """Every sequenced copy of every fragment as its own pair: [(gene, left blocks, right blocks, raw mass)]
    -- what a driver reads from the BAM before HitCluster::collapseAndFilterHits."""
    z = dict(np.load(os.path.join(directory, "reads.npz")))
    out = []
    for k, (gi, lb, rb, _) in enumerate(load_reads(directory)):
        for n in z["nh"][z["nh_off"][k]:z["nh_off"][k + 1]]:
            out.append((gi, lb, rb, 0.5 / int(n) + 0.5 / int(n)))   # src/read.cpp:49-53; 1/NH for an unpaired read
    return out


def locus_of_gene_index(directory, names):
    """reads.npz numbers genes in annotation-file order; `names` is the reference's output order (chromosome by
    chromosome, the order of its -f table): annotation index -> locus index."""
    import e2e_util as U
    ann = list(U.parse_annotation(os.path.join(directory, "toy.gtf")))
    return [names.index(g) for g in ann]


def e2e_inputs(directory, ordered_genes):
    """-> (Annotation, Hits, gene names, rejected pair count).  Loci in gene order with the reference's
    isoform order; hits as HitCluster::collapseAndFilterHits leaves them: sorted by (left, right) of
    the pair (src/alignments.cpp:660, src/read.cpp:917-923), rejected pairs dropped.  Hits.mass is
    (float) collapse mass; Hits.total_mapped the reference's _total_mapped_reads: the int-truncated
    raw mass of every cluster, summed (src/alignments.cpp:1372)."""
    names = list(ordered_genes)
    annot = eb.Annotation([[ex for _, ex in ordered_genes[g]] for g in names])
    locus_of = locus_of_gene_index(directory, names)
    rows = []
    rejected = 0
    cluster_mass = [0.0] * len(names)
    for gi, lb, rb, mass in load_reads(directory):
        gi = locus_of[gi]
        cluster_mass[gi] += mass
        f = eb.hit_features(lb, rb)
        if f is None:
            rejected += 1
            continue
        rows.append((gi, lb[0][0], (rb or lb)[-1][1], f, mass))
    rows.sort(key=lambda r: (r[0], r[1], r[2]))
    hits = eb.Hits([r[0] for r in rows], [r[3] for r in rows], mass=[r[4] for r in rows])
    hits.total_mapped = sum(int(m) for m in cluster_mass)
    return annot, hits, names, rejected


def oracle_words(orc, annot, hits, compat_words=None, key_words=None):
    """(compat, key) bit words from the oracle's restatement, one (hit, isoform) at a time."""
    cw = compat_words or annot.compat_words
    kw = key_words or annot.key_words
    compat = np.zeros((hits.n_hits, cw), np.uint32)
    key = np.zeros((hits.n_hits, kw), np.uint32)
    for h in range(hits.n_hits):
        loc = int(hits.hit_locus[h])
        f = slice(hits.feat_off[h], hits.feat_off[h + 1])
        code, left, right = hits.feat_code[f], hits.feat_left[f], hits.feat_right[f]
        if len(code) == 0:
            continue
        for j, iso in enumerate(range(annot.iso_off[loc], annot.iso_off[loc + 1])):
            e = slice(annot.exon_off[iso], annot.exon_off[iso + 1])
            if e.stop > e.start and orc.is_compatible(code, left, right, annot.exon_left[e], annot.exon_right[e]):
                compat[h, j >> 5] |= np.uint32(1 << (j & 31))
        s = slice(annot.seg_off[loc], annot.seg_off[loc + 1])
        k = orc.overlap_key(code, left, right, annot.seg_left[s], annot.seg_right[s])
        for b in np.nonzero(k)[0]:
            key[h, b >> 5] |= np.uint32(1 << (int(b) & 31))
    return compat, key


def tile(annot, hits, copies, stride=None):
    """`copies` copies of the loci (and their hits) laid along the genome `stride` apart."""
    a, h = annot, hits
    stride = stride or int(max(a.exon_right.max(), h.feat_right.max()) + 100000)
    assert stride * copies < 2 ** 32
    big = eb.Annotation.__new__(eb.Annotation)
    n_iso, n_exon, n_seg = int(a.iso_off[-1]), int(a.exon_off[-1]), int(a.seg_off[-1])
    rep = lambda off, total: np.concatenate([[0]] + [off[1:] + k * total for k in range(copies)]).astype(np.int64)  # noqa: E731
    shift = lambda x: np.concatenate([x + np.uint32(k * stride) for k in range(copies)]).astype(np.uint32)  # noqa: E731
    big.n_loci = a.n_loci * copies
    big.iso_off, big.exon_off, big.seg_off = rep(a.iso_off, n_iso), rep(a.exon_off, n_exon), rep(a.seg_off, n_seg)
    big.exon_left, big.exon_right = shift(a.exon_left), shift(a.exon_right)
    big.seg_left, big.seg_right = shift(a.seg_left), shift(a.seg_right)
    big.compat_words, big.key_words = a.compat_words, a.key_words
    bh = eb.Hits.from_arrays(
        np.concatenate([h.hit_locus + k * a.n_loci for k in range(copies)]), rep(h.feat_off, int(h.feat_off[-1])),
        np.tile(h.feat_code, copies), shift(h.feat_left), shift(h.feat_right), np.tile(h.mass, copies))
    return big, bh


def check_collapse_against_oracle(oracle, n_loci, loc, nh, left, right, hits, cmass, info):
    """Unique hits of a collapse (Hits, cluster masses, info dict) against oracle/collapse_oracle.c, cluster by cluster:
    order, features (Contig(PairedHit) of the representative pair: the reference's own where oracle/_ref is built, else
    sbgpu_hit_features, which the goldens pin), float masses, cluster masses, filtered / rejected counts, the mapped-read
    total.  `nh`: the pairs' NH tags (pair mass 1 / NH)."""
    from oracle import RefLib, have_ref
    ref = RefLib() if have_ref() else None

    def features(lb, rb):
        if ref is not None and hasattr(ref.L, "ref_pairedhit_features"):
            return ref.pairedhit_features(lb, rb)
        return eb.hit_features(lb, rb)
    at = filtered = rejected = total = 0
    by_locus = [[] for _ in range(n_loci)]
    for i, l in enumerate(loc):
        by_locus[l].append(i)
    for l in range(n_loci):
        idx = by_locus[l]
        if not idx:
            assert cmass[l] == 0.0
            continue
        up, um, cm, nf = oracle.collapse_cluster([left[i] for i in idx], [right[i] for i in idx], [nh[i] for i in idx])
        filtered += nf
        assert abs(cmass[l] - cm) <= 1e-12 * max(1.0, cm), (l, cmass[l], cm)
        total += int(cmass[l])
        for a, m in zip(up, um):
            f = features(left[idx[a]], right[idx[a]])
            if not f:
                rejected += 1
                continue
            assert hits.hit_locus[at] == l
            s = slice(int(hits.feat_off[at]), int(hits.feat_off[at + 1]))
            got = ([int(x) for x in hits.feat_code[s]], [int(x) for x in hits.feat_left[s]], [int(x) for x in hits.feat_right[s]])
            assert got == ([int(x) for x in f[0]], [int(x) for x in f[1]], [int(x) for x in f[2]]), (l, at)
            assert hits.mass[at] == np.float32(m), (l, at, hits.mass[at], m)
            at += 1
    assert at == hits.n_hits
    assert info["filtered"] == filtered and info["rejected"] == rejected and info["total_mapped"] == total
