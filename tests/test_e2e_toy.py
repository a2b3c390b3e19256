"""End-to-end golden from the reference BINARY (tests/golden/e2e_toy, made by
tools/make_e2e_golden.py): annotation -> bins -> weights -> EM -> FPKM/Frac/TPM.
CPU part: the oracle against the reference's printed outputs.  GPU part (marked): the HIP
kernels on the same inputs."""
import numpy as np
import pytest

import e2e_util as U

RL, MEAN, SD = 75, 250.0, 30.0


@pytest.fixture(scope="module")
def toy():
    return U.load(U.E2E)


@pytest.fixture(scope="module")
def toy_long():
    return U.load(U.E2E_LONG)


def pairs_from_ctx(genes, rows):
    """Every (bin, isoform) weight the reference printed, with the kernel inputs we derive."""
    out = []
    cache = {}
    for r in rows:
        g = r["gene"]
        if g not in cache:
            tx = genes[g]
            segs = U.disjoint_segments([e for _, ex in tx for e in ex])
            cache[g] = [(t, U.isoform_segments(segs, ex), sum(b - a + 1 for a, b in ex)) for t, ex in tx]
        for j, (t, iso_segs, L) in enumerate(cache[g]):
            assert r["transcripts"][j] == t
            if r["F"][j] == 0.0:
                continue
            bu = U.bin_under_iso(r["coords"], iso_segs)
            assert bu is not None, (g, t, r["coords"])
            out.append((bu[0], bu[1], L, r["F"][j]))
    return out


def test_bin_weights_match_reference_context_table(toy, oracle):
    """A4 end to end: disjoint segments + bin_under_iso + effective_len + pdf + sum reproduce
    every nonzero weight of the reference's -f table (12 significant digits)."""
    genes, rows, _, _ = toy
    pairs = pairs_from_ctx(genes, rows)
    assert len(pairs) > 150
    ins = oracle.make_insert(MEAN, SD)
    worst = 0.0
    for segs, imp, L, ref in pairs:
        w = oracle.bin_weight(segs, imp, L, RL, ins)
        worst = max(worst, abs(w - ref) / ref)
    assert worst < 5e-11, worst
    assert max(len(p[0]) for p in pairs) >= 5  # the brute-force >= 5-segment branch is exercised


def locus_inputs(rows, gene):
    rs = [r for r in rows if r["gene"] == gene]
    n = np.array([r["count"] for r in rs], np.int32)
    F = np.array([r["F"] for r in rs], np.float64)
    return n, F


def test_em_reproduces_reference_theta_log(toy_long, oracle):
    """A1/A2 end to end: the -f table's bins (count, weights at 12 digits) through the EM give
    the theta the reference logged (`isoform k has %f raw read count`, src/estimate.cpp:312).
    Uses the long-exon toy: there a bin's fragments all share one isoform compatibility, so the
    table (which prints the weights seen by the LAST fragment of a bin, alignments.cpp:1556-1563)
    is the complete EM input; fragments are unique, so the bin count equals n_i."""
    genes, rows, _, theta_log = toy_long
    assert len(theta_log) == len(genes)
    long_run = 0
    for (g, tx), ref_theta in zip(genes.items(), theta_log):
        n, F = locus_inputs(rows, g)
        theta, status, iters = oracle.em_locus(n, F)
        assert status in (0, 3)
        assert np.abs(theta - np.array(ref_theta)).max() < 1e-6, (g, theta, ref_theta)
        long_run = max(long_run, iters)
    assert long_run > 200  # a slowly converging locus is part of the fixture


def test_abundance_and_tpm_reproduce_reference_gtf(toy_long, oracle):
    """A3/A7/A9: theta (as logged, 6 decimals) -> FPKM/Frac/TPM agree with the GTF attributes to
    the precision theta was logged with; Frac and TPM each sum to 1 / 1e6."""
    genes, rows, gtf, theta_log = toy_long
    total_mapped = rows[0]["total_mapped"]
    fpkm_all, keep_all, names = [], [], []
    for (g, tx), th in zip(genes.items(), theta_log):
        length = [sum(b - a + 1 for a, b in ex) for _, ex in tx]
        fpkm, frac, keep, _ = oracle.abundance_locus(th, length, total_mapped, min_isoform_frac=0.0)  # -r
        for (t, _), f, fr in zip(tx, fpkm, frac):
            ref_f, ref_fr, _ = gtf[t]
            assert abs(f - float(ref_f)) <= 2e-6 * max(1.0, abs(f)) * 10, (t, f, ref_f)
            assert abs(fr - float(ref_fr)) < 2e-6, (t, fr, ref_fr)
            names.append(t)
        fpkm_all += list(fpkm)
        keep_all += list(keep)
    tpm, _ = oracle.tpm(np.array(fpkm_all), np.array(keep_all, np.int32))
    for t, v in zip(names, tpm):
        assert abs(v - float(gtf[t][2])) <= 2e-5 * max(1.0, v), (t, v, gtf[t][2])
    assert abs(tpm.sum() - 1e6) < 1e-3
    # the GTF strings are the first 11 characters of to_string() (contig.cpp:678-700)
    assert all(len(s) <= 11 for v in gtf.values() for s in v)


@pytest.mark.gpu
def test_gpu_pipeline_on_the_reference_toy(toy, toy_long, oracle):
    """The HIP path on the same toy: bin-weight kernel vs the reference's table, then EM +
    epilogue + TPM vs the reference's log / GTF."""
    from strawberry_amd import em, synth
    from strawberry_amd.binweight import InsertSize, bin_weights, pack_pairs
    genes, rows, gtf, theta_log = toy
    ctx = em.default_context(0)
    pairs = pairs_from_ctx(genes, rows)
    seg_off, seg_lens, mask = pack_pairs([p[0] for p in pairs], [p[1] for p in pairs])
    w = bin_weights(seg_off, seg_lens, mask, [p[2] for p in pairs], InsertSize(MEAN, SD), RL, ctx=ctx)
    ref = np.array([p[3] for p in pairs])
    assert (np.abs(w - ref) / ref).max() < 5e-11
    genes, rows, gtf, theta_log = toy_long
    loci, lengths = [], []
    for g, tx in genes.items():
        loci.append(locus_inputs(rows, g))
        lengths.append([sum(b - a + 1 for a, b in ex) for _, ex in tx])
    b = synth.from_loci(loci, lengths)
    s = em.EmBatchSolver(b, ctx)
    s.run_em()
    s.run_abundance(rows[0]["total_mapped"], min_isoform_frac=0.0)
    s.run_tpm()
    r = s.results()
    for l, ref_theta in enumerate(theta_log):
        th = r["theta"][b.iso_off[l]:b.iso_off[l + 1]]
        assert np.abs(th - np.array(ref_theta)).max() < 1e-6
    names = [t for _, tx in genes.items() for t, _ in tx]
    for t, f, fr, tp in zip(names, r["fpkm"], r["frac"], r["tpm"]):
        assert abs(f - float(gtf[t][0])) <= 1e-5 * max(1.0, f)
        assert abs(fr - float(gtf[t][1])) < 2e-6
        assert abs(tp - float(gtf[t][2])) <= 1e-5 * max(1.0, tp)
    o_theta, o_status, o_iters = oracle.em_batch(b.row_off, b.iso_off, b.f_off, b.count, b.F)
    np.testing.assert_array_equal(r["iters"], o_iters)
    assert (np.abs(r["theta"] - o_theta) / np.maximum(np.abs(o_theta), 1e-9)).max() < 1e-9
    assert abs(r["tpm"].sum() - 1e6) < 1e-3


def test_output_formatting_matches_reference_gtf(toy_long, oracle):
    """A9: the strings Contig::print2gtf writes.  Known answers of the 11-character cut, then the
    whole toy: our formatting of (theta as logged -> FPKM/Frac/TPM) reproduces the reference's
    transcript lines -- theta is only known to its 6 logged decimals, so the last printed digit of a
    value may differ; everything else must be byte-identical."""
    from strawberry_amd.output import format_value, gtf_transcript
    assert format_value(28404.567028) == "28404.56702"      # to_string -> 28404.567028, cut to 11
    assert format_value(0.181426) == "0.181426"
    assert format_value(1e6) == "1000000.000"
    assert format_value(0.0) == "0.000000"
    assert format_value(123456789.123) == "123456789.1"
    genes, rows, gtf, theta_log = toy_long
    ref_lines = {}
    for line in open(U.E2E_LONG + "/out.gtf"):
        if "\ttranscript\t" in line:
            t = line.split('transcript_id "')[1].split('"')[0]
            ref_lines[t] = line
    total_mapped = rows[0]["total_mapped"]
    fpkm_all, keep_all, info = [], [], []
    for (g, tx), th in zip(genes.items(), theta_log):
        length = [sum(b - a + 1 for a, b in ex) for _, ex in tx]
        fpkm, frac, keep, _ = oracle.abundance_locus(th, length, total_mapped, min_isoform_frac=0.0)
        for (t, ex), f, fr in zip(tx, fpkm, frac):
            info.append((g, t, ex, f, fr))
        fpkm_all += list(fpkm)
        keep_all += list(keep)
    tpm, _ = oracle.tpm(np.array(fpkm_all), np.array(keep_all, np.int32))
    same = total = 0
    for (g, t, ex, f, fr), tp in zip(info, tpm):
        mine = gtf_transcript("chr1", "+", g, t, ex, f, fr, tp, ref_gene_id=g, ref_gene_name=g).split("\n")[0] + "\n"
        ref = ref_lines[t]
        # same structure byte for byte; the three numbers equal up to the digits theta was logged with
        a, b = mine.split('"'), ref.split('"')
        assert len(a) == len(b)
        for x, y in zip(a, b):
            if x == y:
                same += 1
            else:
                assert abs(float(x) - float(y)) <= 2e-5 * max(1.0, abs(float(y))), (x, y)
            total += 1
    assert same / total > 0.85  # all the text and most of the numbers are identical strings
    # exon lines: ` exon_id "k";` appended, one per exon
    g, t, ex, f, fr = info[0]
    block = gtf_transcript("chr1", "+", g, t, ex, f, fr, tpm[0], ref_gene_id=g, ref_gene_name=g).split("\n")
    assert len(block) == len(ex) + 2 and block[-1] == ""
    assert block[1].endswith(' exon_id "1";') and "\texon\t%d\t%d\t1000\t+\t.\t" % ex[0] in block[1]


@pytest.mark.parametrize("which", ["E2E", "E2E_LONG"])
def test_context_row_formatting_round_trips_reference_table(which):
    """sbgpu_format_context_row (host code): parsing a row of the reference's -f table and printing it
    again gives the same bytes -- to_string for FPKM/Frac, 12 significant digits for the weights."""
    from strawberry_amd.output import CONTEXT_HEADER, context_row
    d = getattr(U, which)
    lines = open(d + "/ctx.tsv").read().splitlines(keepends=True)
    assert lines[0] == CONTEXT_HEADER
    for line in lines[1:]:
        f = line.rstrip("\n").split("\t")
        coords = [(int(a), int(b)) for a, b in __import__("re").findall(r"\[(\d+)-(\d+)\]", f[8])]
        again = context_row(f[0], int(f[1]), f[2], int(f[3]), f[4].split(","), [float(x) for x in f[5].split(",")],
                            [float(x) for x in f[6].split(",")], [float(x) for x in f[7].split(",")], coords, int(f[9]))
        assert again == line


def test_expression_filter_reproduces_reference_gtf(oracle):
    """A3 filter + A7: the reference run with `-r -e 0.05` (tests/golden/e2e_toy_filter) erases the
    isoforms below 5 % of their locus after the EM (estimate.cpp:346-355) and takes TPM over the
    survivors; the oracle's epilogue on the logged theta must keep exactly the transcripts the GTF
    holds and agree with its numbers."""
    genes, rows, gtf, theta_log = U.load(U.E2E_FILTER)
    total_mapped = rows[0]["total_mapped"]
    fpkm_all, keep_all, names = [], [], []
    for (g, tx), th in zip(genes.items(), theta_log):
        length = [sum(b - a + 1 for a, b in ex) for _, ex in tx]
        fpkm, frac, keep, _ = oracle.abundance_locus(th, length, total_mapped, min_isoform_frac=0.05)
        for (t, _), f, fr, k in zip(tx, fpkm, frac, keep):
            assert bool(k) == (t in gtf), (t, fr)
            if k:
                assert abs(f - float(gtf[t][0])) <= 2e-5 * max(1.0, abs(f))
                assert abs(fr - float(gtf[t][1])) < 2e-6
            names.append(t)
        fpkm_all += list(fpkm)
        keep_all += list(keep)
    assert 0 < sum(1 for k in keep_all if not k) < len(keep_all)
    tpm, _ = oracle.tpm(np.array(fpkm_all), np.array(keep_all, np.int32))
    for t, v, k in zip(names, tpm, keep_all):
        if k:
            assert abs(v - float(gtf[t][2])) <= 2e-5 * max(1.0, v), (t, v, gtf[t][2])
    assert abs(tpm[np.array(keep_all) != 0].sum() - 1e6) < 1e-3
