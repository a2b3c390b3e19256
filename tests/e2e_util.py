"""Helpers for the end-to-end golden (tests/golden/e2e_toy): parse the reference's outputs
and restate, for TEST purposes only, the two small structural steps that sit between the
annotation and the bin-weight kernel's inputs:

  * disjoint exon segments  -- IRanges::disjoint, /root/reference/include/interval.hpp:150-191
  * ExonBin::bin_under_iso  -- /root/reference/include/isoform.h:363-411
"""
import os
import re
from bisect import bisect_left

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
E2E = os.path.join(HERE, "golden", "e2e_toy")            # short exons: bins over many segments
E2E_LONG = os.path.join(HERE, "golden", "e2e_toy_long")  # exons longer than any mate gap
E2E_MASS = os.path.join(HERE, "golden", "e2e_toy_mass")  # PCR duplicates + multi-mapped pairs (masses 1, 1/2, 1/3)
E2E_EMP = os.path.join(HERE, "golden", "e2e_toy_emp")     # e2e_toy's reads without -i: empirical insert-size distribution
E2E_SINGLE = os.path.join(HERE, "golden", "e2e_toy_single")  # single-end library: unpaired reads, insert size forced to N(200, 80)
E2E_LONGREAD = os.path.join(HERE, "golden", "e2e_toy_longread")  # unpaired reads of 1001-2600 bases: long-read workflow, F = 1/L
E2E_BIAS = os.path.join(HERE, "golden", "e2e_toy_bias")     # -b genome.fa: six sequence columns per bin in the -f table
E2E_MINUS = os.path.join(HERE, "golden", "e2e_toy_minus")   # every other gene on the minus strand
E2E_CHROMS = os.path.join(HERE, "golden", "e2e_toy_chroms")  # genes alternating between two chromosomes
E2E_FILTER = os.path.join(HERE, "golden", "e2e_toy_filter")  # e2e_toy_long's reads with -e 0.05: isoforms erased
E2E_ASSEMBLY = os.path.join(HERE, "golden", "e2e_toy_assembly")  # DEFAULT mode (no -g / -r): assembled contigs quantified, Frac < 0.01 erased


def load(directory):
    """-> (genes in the reference's isoform order, ctx rows, gtf attrs, theta log)."""
    genes = parse_annotation(os.path.join(directory, "toy.gtf"))
    rows = parse_ctx(os.path.join(directory, "ctx.tsv"))
    # the reference orders a locus' isoforms its own way (sorted by position); the -f table's
    # `transcripts` column is that order, and the theta log follows it.  With the expression filter on,
    # the table only lists the surviving isoforms: the order then comes from the unfiltered run of the
    # same annotation (e2e_toy_long).
    order_rows = parse_ctx(os.path.join(E2E_LONG, "ctx.tsv")) if directory == E2E_FILTER else rows
    ordered = {}
    for r in order_rows:
        if r["gene"] not in ordered:
            by_name = dict(genes[r["gene"]])
            ordered[r["gene"]] = [(t, by_name[t]) for t in r["transcripts"]]
    return (ordered, rows, parse_out_gtf(os.path.join(directory, "out.gtf")),
            parse_theta_log(os.path.join(directory, "theta_log.txt")))


def parse_annotation(path=os.path.join(E2E, "toy.gtf")):
    """-> {gene: [(transcript_id, [(l, r), ...]), ...]} in file order."""
    genes = {}
    for line in open(path):
        f = line.rstrip("\n").split("\t")
        if len(f) < 9 or f[2] != "exon":
            continue
        g = re.search(r'gene_id "([^"]+)"', f[8]).group(1)
        t = re.search(r'transcript_id "([^"]+)"', f[8]).group(1)
        tx = genes.setdefault(g, [])
        if not tx or tx[-1][0] != t:
            tx.append((t, []))
        tx[-1][1].append((int(f[3]), int(f[4])))
    return genes


def gene_strands(directory, column=6):
    """-> {gene: '+' / '-'} from the annotation's strand column."""
    out = {}
    for line in open(os.path.join(directory, "toy.gtf")):
        f = line.rstrip("\n").split("\t")
        if len(f) >= 9 and f[2] == "exon":
            out[re.search(r'gene_id "([^"]+)"', f[8]).group(1)] = f[column]
    return out


def gene_chroms(directory):
    """-> {gene: chromosome name} from the annotation."""
    return gene_strands(directory, column=0)


def disjoint_segments(exons):
    """Cut the union of closed intervals at every exon boundary, keep the covered pieces."""
    bars = sorted(set([l for l, r in exons] + [r + 1 for l, r in exons]))
    out = []
    for a, b in zip(bars[:-1], bars[1:]):
        if any(l <= a and b - 1 <= r for l, r in exons):
            out.append((a, b - 1))
    return out


def isoform_segments(segs, exons):
    """Isoform::_exon_segs: the disjoint segments contained in one of the isoform's exons."""
    return [s for s in segs if any(l <= s[0] and s[1] <= r for l, r in exons)]


def bin_under_iso(bin_coords, iso_segs):
    """-> (seg_lens, implicit_idx) of ExonBin::bin_under_iso, or None when the bin's first or
    last segment is not one of the isoform's (the reference never asks in that case)."""
    starts = [s[0] for s in iso_segs]
    lo = bisect_left(starts, bin_coords[0][0])
    up = bisect_left(starts, bin_coords[-1][0])
    if lo >= len(starts) or up >= len(starts) or starts[lo] != bin_coords[0][0] or starts[up] != bin_coords[-1][0]:
        return None
    segs = iso_segs[lo:up + 1]
    implicit = []
    c = 1
    i = 1
    while i < len(segs) - 1:
        if c < len(bin_coords) and segs[i][0] == bin_coords[c][0]:
            i += 1
            c += 1
        elif c >= len(bin_coords) or segs[i][0] < bin_coords[c][0]:
            implicit.append(i)
            i += 1
        else:
            return None  # assert(false) in the reference: the bin has a segment the isoform lacks
    return [r - l + 1 for l, r in segs], implicit


def parse_ctx(path=os.path.join(E2E, "ctx.tsv")):
    """-> list of dict(gene, transcripts, F[list of float], coords[(l,r)...], count, total_mapped)."""
    rows = []
    for k, line in enumerate(open(path)):
        f = line.rstrip("\n").split("\t")
        if k == 0 or len(f) < 10:
            continue
        rows.append({
            "total_mapped": int(f[1]), "gene": f[2], "transcripts": f[4].split(","),
            "fpkm": [float(x) for x in f[5].split(",")], "F": [float(x) for x in f[6].split(",")],
            "frac": [float(x) for x in f[7].split(",")],
            "coords": [(int(a), int(b)) for a, b in re.findall(r"\[(\d+)-(\d+)\]", f[8])], "count": int(f[9]),
            "seq": f[10:]})     # with -b: gc, entropy (six decimals) and four 0 / 1 flags
    return rows


def parse_out_gtf(path=os.path.join(E2E, "out.gtf")):
    """-> {transcript_id: (FPKM str, Frac str, TPM str)} from the transcript lines."""
    out = {}
    for line in open(path):
        f = line.rstrip("\n").split("\t")
        if len(f) < 9 or f[2] != "transcript":
            continue
        t = re.search(r'transcript_id "([^"]+)"', f[8]).group(1)
        out[t] = tuple(re.search(r'%s "([^"]+)"' % k, f[8]).group(1) for k in ("FPKM", "Frac", "TPM"))
    return out


def parse_theta_log(path=os.path.join(E2E, "theta_log.txt")):
    """-> list (one per locus, in output order) of lists of theta (printed %f)."""
    loci = []
    for line in open(path):
        m = re.match(r"isoform (\d+) has ([0-9.eE+-]+) raw read count", line)
        if not m:
            continue
        if int(m.group(1)) == 1:
            loci.append([])
        loci[-1].append(float(m.group(2)))
    return loci
