"""GTF output exactly as the reference prints it.

Reference: Contig::print2gtf, /root/reference/src/contig.cpp:636-721; the FPKM / Frac / TPM
strings are std::to_string(double) cut to 11 characters.  Formatting is host work behind
sbgpu_format_value / sbgpu_format_gtf_transcript.
"""
import ctypes as C

import numpy as np

from . import _lib


def format_value(v):
    """first 11 characters of "%f" % v"""
    out = C.create_string_buffer(12)
    _lib.check(_lib.load().sbgpu_format_value(float(v), out), "sbgpu_format_value")
    return out.value.decode()


def gtf_transcript(chrom, strand, gene_id, transcript_id, exons, fpkm, frac, tpm, keep=1, ref_gene_id="",
                   ref_gene_name=""):
    """The transcript line + exon lines of one isoform, as print2gtf writes them."""
    L = _lib.load()
    left = np.ascontiguousarray([e[0] for e in exons], np.int32)
    right = np.ascontiguousarray([e[1] for e in exons], np.int32)
    cap = 4096 + 400 * len(exons)
    buf = C.create_string_buffer(cap)
    n = L.sbgpu_format_gtf_transcript(buf, cap, chrom.encode(), strand.encode(), gene_id.encode(),
                                      transcript_id.encode(), ref_gene_id.encode(), ref_gene_name.encode(), len(exons),
                                      left.ctypes.data, right.ctypes.data, float(fpkm), float(frac), float(tpm),
                                      int(keep))
    if n < 0:
        _lib.check(n, "sbgpu_format_gtf_transcript")
    return buf.value.decode()


CONTEXT_HEADER = "\t".join([
    "sample", "sample_frag_count", "gene_id", "gene_frag_count", "transcripts", "FPKMs", "conditional_probabilities",
    "class_probabilities", "path_symbol", "path_count", "path_gc_content", "path_hexmer_entropy", "gc_stretch_0.8_20",
    "gc_stretch_0.9_20", "gc_stretch_0.8_40", "gc_stretch_0.9_40"]) + "\n"   # alignments.cpp:1746-1752


def context_row(sample, sample_frag_count, gene_id, gene_frag_count, transcript_ids, fpkm, cond_prob, frac, segs,
                path_count, seq_stats=None):
    """One row of the `-f` table (Sample::printContext, alignments.cpp:1549-1639, columns 1-10).
    seq_stats = (gc, entropy, flags) of the bin (binseq.bin_sequence_stats) adds the six columns of a run
    with `-b genome.fa` (alignments.cpp:1622-1636)."""
    L = _lib.load()
    n = len(transcript_ids)
    names = (C.c_char_p * n)(*[t.encode() for t in transcript_ids])
    fpkm = np.ascontiguousarray(fpkm, np.float64)
    cond = np.ascontiguousarray(cond_prob, np.float64)
    frac = np.ascontiguousarray(frac, np.float64)
    sl = np.ascontiguousarray([s[0] for s in segs], np.uint32)
    sr = np.ascontiguousarray([s[1] for s in segs], np.uint32)
    cap = 512 + 64 * n * 4 + 32 * len(segs) + sum(len(t) for t in transcript_ids)
    buf = C.create_string_buffer(cap)
    args = (buf, cap, sample.encode(), int(sample_frag_count), gene_id.encode(),
            int(gene_frag_count), n, names, fpkm.ctypes.data, cond.ctypes.data, frac.ctypes.data,
            len(segs), sl.ctypes.data if len(segs) else None, sr.ctypes.data if len(segs) else None, int(path_count))
    if seq_stats is None:
        r = L.sbgpu_format_context_row(*args)
    else:
        r = L.sbgpu_format_context_row_seq(*args, float(seq_stats[0]), float(seq_stats[1]), int(seq_stats[2]))
    if r < 0:
        _lib.check(r, "sbgpu_format_context_row")
    return buf.value.decode()


def context_table(sample, total_mapped, gene_ids, transcript_ids, bins, compat, F, fpkm, frac, keep=None, seq_stats=None):
    """The whole `-f` table of a batch of loci from the chain's results.

    bins: exonbin.LocusBins; compat: the kernel's words per hit [n_hits, cw]; F: the EM batch's weights
    (host copy); fpkm / frac per isoform; transcript_ids[l] the isoform names of locus l in batch order;
    keep: per isoform, False for the ones the expression filter erased (estimate.cpp:346-355).
    Sample::printContext runs after the filter, over the surviving isoforms only: a hit counts if it is
    compatible with one of them, per bin the reference prints the weights of the isoforms its LAST such
    hit is compatible with and the number of such hits, bins in std::map order of their coordinate sets.
    seq_stats: (gc[n_bins], entropy[n_bins], flags[n_bins]) from binseq.bin_sequence_stats for a `-b` run."""
    out = [CONTEXT_HEADER]
    hit_bin = np.asarray(bins.hit_bin)
    keep = np.ones(int(bins.iso_off[-1]), bool) if keep is None else np.asarray(keep) != 0
    n_in_bin = np.zeros(bins.n_bins, np.int64)
    last_hit = np.full(bins.n_bins, -1, np.int64)
    hit_locus = np.searchsorted(bins.row_off, hit_bin, side="right") - 1
    for h in np.nonzero(hit_bin >= 0)[0]:
        l = hit_locus[h]
        i0, niso = int(bins.iso_off[l]), int(bins.iso_off[l + 1] - bins.iso_off[l])
        if any((int(compat[h][j >> 5]) >> (j & 31)) & 1 and keep[i0 + j] for j in range(niso)):
            n_in_bin[hit_bin[h]] += 1
            last_hit[hit_bin[h]] = h                      # later hits overwrite earlier ones
    for l, gene in enumerate(gene_ids):
        b0, b1 = int(bins.row_off[l]), int(bins.row_off[l + 1])
        i0, i1 = int(bins.iso_off[l]), int(bins.iso_off[l + 1])
        niso = i1 - i0
        kept = [j for j in range(niso) if keep[i0 + j]]
        if not kept:
            continue
        coords = bins.bin_coords(l)
        gene_frags = int(n_in_bin[b0:b1].sum())
        Fl = np.asarray(F[bins.f_off[l]:bins.f_off[l + 1]]).reshape(b1 - b0, niso)
        for b in sorted(range(b1 - b0), key=lambda k: coords[k]):
            if n_in_bin[b0 + b] == 0:
                continue
            words = compat[last_hit[b0 + b]]
            prob = [Fl[b, j] if (int(words[j >> 5]) >> (j & 31)) & 1 else 0.0 for j in kept]
            out.append(context_row(sample, total_mapped, gene, gene_frags, [transcript_ids[l][j] for j in kept],
                                   [fpkm[i0 + j] for j in kept], prob, [frac[i0 + j] for j in kept], coords[b],
                                   n_in_bin[b0 + b],
                                   None if seq_stats is None else tuple(x[b0 + b] for x in seq_stats)))
    return "".join(out)
