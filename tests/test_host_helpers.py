"""CPU tests of the host-side helpers round 6 added (no GPU, no library call that computes): the insert-size law from its
histogram alone, annotations cut and joined, the chunking of a record stream."""
import numpy as np
import pytest


def test_law_from_histogram_equals_law_from_lengths():
    """InsertSize.from_hist (what sbgpu_quantify_* builds from the device's pass 1: integer sums, one rounding each) ==
    InsertSize.from_frag_lens (the reference's InsertSize(frag_lens), src/read.cpp:238-262: running doubles, exact below 2^53):
    mean, sd, extremes, histogram -- bitwise -- and the oracle's pdf agrees on both."""
    from strawberry_amd.binweight import InsertSize
    rng = np.random.default_rng(5)
    for n in (1, 2, 17, 5000, 200000):
        fl = np.maximum(1, np.rint(rng.normal(250, 30, n))).astype(np.int64)
        if n > 100:
            fl[:7] = [151, 152, 2189, 900, 151, 640, 333]      # a thin tail: holes in the histogram
        a = InsertSize.from_frag_lens(fl)
        lo = int(fl.min())
        b = InsertSize.from_hist(lo, np.bincount(fl - lo))
        assert (a.mean, a.sd, a.start_offset, a.end_offset, a.total_reads) == (b.mean, b.sd, b.start_offset, b.end_offset, b.total_reads)
        np.testing.assert_array_equal(a.emp_hist, b.emp_hist)
    with pytest.raises(ValueError):
        InsertSize.from_hist(10, np.zeros(4))


def test_annotation_prefix_and_concat():
    from strawberry_amd import exonbin as eb
    from strawberry_amd import synth
    loci = synth.make_gene_models(40, seed=9)
    whole = eb.Annotation(loci)
    head, tail = eb.Annotation(loci[:15]), eb.Annotation(loci[15:])
    p = whole.prefix(15)
    for name in ("iso_off", "exon_off", "seg_off", "exon_left", "exon_right", "seg_left", "seg_right"):
        np.testing.assert_array_equal(getattr(p, name), getattr(head, name), err_msg=name)
    assert (p.n_loci, p.compat_words, p.key_words) == (head.n_loci, head.compat_words, head.key_words)
    j = eb.Annotation.concat([head, tail])
    for name in ("iso_off", "exon_off", "seg_off", "exon_left", "exon_right", "seg_left", "seg_right"):
        np.testing.assert_array_equal(getattr(j, name), getattr(whole, name), err_msg=name)
    assert (j.n_loci, j.compat_words, j.key_words) == (whole.n_loci, whole.compat_words, whole.key_words)
