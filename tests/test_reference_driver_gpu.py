"""The drop-in, proven with the reference's OWN driver (north star: "drops in under the existing Strawberry.cpp
driver").  oracle/_ref/strawberry_sbgpu is the reference program linked from its unmodified objects -- main, BAM
decode, clustering, LocusContext, GTF / table output -- with exactly the two functions of the seam replaced:
EmSolver::init and EmSolver::run (/root/reference/src/estimate.cpp:366-488, called at :305-308) are weakened in a
copy of estimate.o and defined by oracle/sbgpu_em_shim.cpp over sbgpu::EmSolver (include/sbgpu_host.hpp), i.e. the
HIP kernels behind the C ABI.  The toy BAMs are regenerated from the committed fragments (tests/golden/*/reads.npz,
the simulation's own order and read names) with oracle/_ref/sam2bam; the program runs with the golden runs'
command lines; its out.gtf and -f table must equal the reference binary's committed outputs byte for byte.

Test infrastructure only: nothing under strawberry_amd/ refers to the binary or the shim.  Skipped where oracle/_ref
was not built (it needs /root/reference at build time; the built files travel to the GPU box)."""
import os
import subprocess

import numpy as np
import pytest

import e2e_util as U

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
DRIVER = os.path.join(REF_DIR, "strawberry_sbgpu")
# the BATCHED drop-in (oracle/sbgpu_batched_shim.cpp): Sample::procSample replaced as well -- clusters collected with the
# reference's own classes, ONE sbgpu_em_batch call for the whole sample, then the reference's epilogue in cluster order
BATCHED = os.path.join(REF_DIR, "strawberry_sbgpu_batched")
# the drop-in ONE LEVEL UP (oracle/sbgpu_chain_shim.cpp): procSample collects every locus' transcripts and unique hits with the
# reference's classes; ONE sbgpu_quantify_host call does exon bins + bin weights + EM for the whole sample (no LocusContext)
CHAIN = os.path.join(REF_DIR, "strawberry_sbgpu_chain")
# the drop-in AT ITS DEEPEST (oracle/sbgpu_front_shim.cpp): all three passes over the BAM replaced (inspect_read_len, preProcess,
# procSample) -- one inflate, then decode -> read stream -> pairs -> unique hits -> bins -> weights -> EM through the library
FRONT = os.path.join(REF_DIR, "strawberry_sbgpu_front")
# the same restructured loop with the reference's own EmSolver bodies doing the solve: no GPU in it (the CPU suite's check)
BATCHED_REFEM = os.path.join(REF_DIR, "strawberry_batched_refem")
SAM2BAM = os.path.join(REF_DIR, "sam2bam")
RL = 75

# (golden directory, extra command-line arguments of its golden run, -i given?)
RUNS = {
    "E2E": (U.E2E, [], True),
    "E2E_LONG": (U.E2E_LONG, [], True),
    "E2E_MASS": (U.E2E_MASS, ["--allow-multimapped-hits"], True),
    "E2E_FILTER": (U.E2E_FILTER, ["-e", "0.05"], True),
    "E2E_EMP": (U.E2E_EMP, [], False),
    "E2E_MINUS": (U.E2E_MINUS, [], True),
    "E2E_CHROMS": (U.E2E_CHROMS, [], True),
    # BASELINE config 4 (and config 1's plumbing): the reference's DEFAULT mode.  No -g / -r: the first pass assembles the
    # transcripts (host side, the reference's own code: alignments.cpp:1658), the second quantifies the ASSEMBLED contigs
    # (alignments.cpp:1091-1101) -- their EM runs on the device -- and erases isoforms with Frac < 0.01 afterwards
    # (estimate.cpp:346-355; one of this run's 18 isoforms goes)
    "E2E_ASSEMBLY": (U.E2E_ASSEMBLY, [], True),
    # the runs that are not paired-end defaults (round 5):
    # a single-end library: every record unpaired (flag 0); the reference forces the insert size to N(200, 80) whatever
    # -i says (/root/reference/src/Strawberry.cpp:329-333) and every unique hit is one read
    "E2E_SINGLE": (U.E2E_SINGLE, [], True),
    # unpaired reads of 1001-2600 bases: more than ten read lengths above 1000 make it a long-read sample
    # (Strawberry.cpp:292-303), whose bin weights are 1 / L_j (estimate.cpp:236-247)
    "E2E_LONGREAD": (U.E2E_LONGREAD, [], True),
    # -b genome.fa: six sequence columns per bin in the -f table (alignments.cpp:1622-1636); the abundances do not depend on them
    "E2E_BIAS": (U.E2E_BIAS, ["-b", "genome.fa"], True),
}
ASSEMBLY_MODE = {"E2E_ASSEMBLY"}
# what the deepest program (all three BAM passes replaced) does not cover: it says so and stops
FRONT_NOT_COVERED = {"E2E_ASSEMBLY", "E2E_BIAS"}


def need_driver(driver=DRIVER):
    if not (os.path.exists(driver) and os.path.exists(SAM2BAM)):
        from conftest import need_ref
        need_ref("oracle/_ref/%s" % os.path.basename(driver))


def cigar(blocks):
    s = ""
    for k, (a, b) in enumerate(blocks):
        if k:
            s += "%dN" % (a - blocks[k - 1][1] - 1)
        s += "%dM" % (b - a + 1)
    return s


def write_sam(directory, path):
    """The SAM of the golden run, rebuilt from reads.npz: one pair of records per sequenced copy, names in simulation
    order, NH and XS tags as simulated, records sorted by (chromosome, position) -- stable, like the generator's sort."""
    z = dict(np.load(os.path.join(directory, "reads.npz")))
    genes = list(U.parse_annotation(os.path.join(directory, "toy.gtf")))
    strands, chroms = U.gene_strands(directory), U.gene_chroms(directory)
    recs, rid, top = [], 0, 0
    for k in range(len(z["gene"])):
        g = genes[int(z["gene"][k])]
        left = [(int(a), int(b)) for a, b in zip(z["left_l"][z["left_off"][k]:z["left_off"][k + 1]], z["left_r"][z["left_off"][k]:z["left_off"][k + 1]])]
        right = [(int(a), int(b)) for a, b in zip(z["right_l"][z["right_off"][k]:z["right_off"][k + 1]], z["right_r"][z["right_off"][k]:z["right_off"][k + 1]])]
        c = chroms[g]
        if not right:
            # an unpaired read (single-end and long-read libraries): flag 0, no mate fields, the read as long as its blocks
            rlen = sum(b - a + 1 for a, b in left)
            top = max(top, left[-1][1])
            for nh in z["nh"][z["nh_off"][k]:z["nh_off"][k + 1]]:
                rid += 1
                recs.append(((c, left[0][0]), "r%06d\t0\t%s\t%d\t255\t%s\t*\t0\t0\t%s\t%s\tNH:i:%d\tXS:A:%s" % (
                    rid, c, left[0][0], cigar(left), "A" * rlen, "I" * rlen, nh, strands[g])))
            continue
        tlen = right[-1][1] - left[0][0] + 1
        top = max(top, right[-1][1])
        for nh in z["nh"][z["nh_off"][k]:z["nh_off"][k + 1]]:
            rid += 1
            name = "r%06d" % rid
            recs.append(((c, left[0][0]), "%s\t99\t%s\t%d\t255\t%s\t=\t%d\t%d\t%s\t%s\tNH:i:%d\tXS:A:%s" % (
                name, c, left[0][0], cigar(left), right[0][0], tlen, "A" * RL, "I" * RL, nh, strands[g])))
            recs.append(((c, right[0][0]), "%s\t147\t%s\t%d\t255\t%s\t=\t%d\t%d\t%s\t%s\tNH:i:%d\tXS:A:%s" % (
                name, c, right[0][0], cigar(right), left[0][0], -tlen, "A" * RL, "I" * RL, nh, strands[g])))
    recs.sort(key=lambda r: r[0])
    with open(path, "w") as f:
        f.write("@HD\tVN:1.0\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (c, top + 10000) for c in sorted(set(chroms.values()))))
        for _, line in recs:
            f.write(line + "\n")
    return len(recs)


def run_driver(which, tmp_path, driver=DRIVER, with_f=True):
    directory, extra, insert = RUNS[which]
    sam, bam = str(tmp_path / "toy.sam"), str(tmp_path / "toy.bam")
    n = write_sam(directory, sam)
    assert ("%d read records" % n) in open(os.path.join(directory, "README.txt")).read()     # the golden run's input
    subprocess.check_call([SAM2BAM, sam, bam])
    if "-b" in extra:
        # the genome beside the run, with the index the reference cannot build itself (fasta.cpp:89-91): one sequence, 60
        # bases per line
        fa = open(os.path.join(directory, "genome.fa")).read()
        open(str(tmp_path / "genome.fa"), "w").write(fa)
        lines = fa.split("\n")
        assert lines[0].startswith(">") and all(len(l) == 60 for l in lines[1:-2])
        open(str(tmp_path / "genome.fa.fai"), "w").write("%s\t%d\t%d\t60\t61\n" % (lines[0][1:], sum(len(l) for l in lines[1:]), len(lines[0]) + 1))
    annot = [] if which in ASSEMBLY_MODE else ["-g", os.path.join(directory, "toy.gtf"), "-r"]
    cmd = [driver, bam] + annot + (["-i", "250/30"] if insert else []) + [
        "-o", str(tmp_path / "out.gtf"), "-T", str(tmp_path / "log.txt")] + (["-f", str(tmp_path / "ctx.tsv")] if with_f else []) + extra
    return subprocess.run(cmd, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)


def test_reference_driver_over_libsbgpu_has_no_cpu_path(tmp_path):
    """Without a GPU the linked program must fail in sbgpu_init -- the proof that the reference's call site reaches the
    library and that nothing of the reference's own EmSolver is left on the path."""
    need_driver()
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = run_driver("E2E_LONG", tmp_path)
    assert r.returncode != 0 and "sbgpu_init" in (r.stderr + r.stdout)


def check_files(which, tmp_path, r, log_in_locus_order=True, names=("out.gtf", "ctx.tsv")):
    directory = RUNS[which][0]
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    for name in names:
        # (the GTF's first line is a comment holding the program's own command line, temporary paths included)
        got = [l for l in open(str(tmp_path / name), "rb").read().split(b"\n") if not l.startswith(b"#/")]
        want = [l for l in open(os.path.join(directory, name), "rb").read().split(b"\n") if not l.startswith(b"#/")]
        assert got == want, "%s of %s differs from the reference binary's" % (name, which)
    # the theta lines of the reference's log (estimate.cpp:312) as well
    want = open(os.path.join(directory, "theta_log.txt")).read()
    got = "".join(l for l in open(str(tmp_path / "log.txt")) if "raw read count" in l or "not compatible" in l)
    if not log_in_locus_order:
        # the batched loop builds every LocusContext (whose constructor logs the rejected pairs, estimate.hpp:74-76) before the
        # first locus is solved: the same lines, the "not compatible" ones ahead of the theta lines instead of between them
        order = lambda text: "".join(sorted(text.splitlines(True), key=lambda l: "raw read count" in l))  # noqa: E731 (stable)
        want, got = order(want), order(got)
    assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("which", sorted(RUNS))
def test_reference_driver_over_libsbgpu_reproduces_reference_files(which, tmp_path):
    need_driver()
    check_files(which, tmp_path, run_driver(which, tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("which", sorted(RUNS))
def test_batched_reference_driver_reproduces_reference_files(which, tmp_path):
    """collect -> ONE sbgpu_em_batch -> epilogue under the reference's own main: same files, byte for byte."""
    need_driver(BATCHED)
    check_files(which, tmp_path, run_driver(which, tmp_path, BATCHED), log_in_locus_order=False)


@pytest.mark.gpu
@pytest.mark.parametrize("which", sorted(RUNS))
def test_chain_level_reference_driver_reproduces_reference_files(which, tmp_path):
    """The reference's main, BAM decode, clustering, pairing and collapse; bins, weights and EM of ALL loci in one
    sbgpu_quantify_host call; the reference's print2gtf and the library's -f formatter: same files, byte for byte -- in
    quant-only mode, with duplicates and multi-mapped reads, the -e filter, empirical insert sizes, both strands, two
    chromosomes, and in assembly mode (C4: the assembled contigs are the annotation, Frac < 0.01 erased)."""
    need_driver(CHAIN)
    check_files(which, tmp_path, run_driver(which, tmp_path, CHAIN), log_in_locus_order=False)


@pytest.mark.gpu
@pytest.mark.parametrize("which", sorted(set(RUNS) - FRONT_NOT_COVERED))
def test_front_level_reference_driver_reproduces_reference_files(which, tmp_path):
    """The reference's main, option parsing, GTF reader and print2gtf -- and NOTHING of its BAM handling: the file is inflated
    once, every record decided on the device (sbgpu_bam_decode_device = BAMHitFactory::getHitFromBuf), offered to the clusters
    of the reference's own addRef2Cluster (sbgpu_assign_reads_device), paired, collapsed, quantified: same files, byte for
    byte -- duplicates and multi-mapped reads, the -e filter, empirical insert sizes (the sample of fragment lengths comes
    from sbgpu_frag_lens_host), both strands, two chromosomes."""
    need_driver(FRONT)
    check_files(which, tmp_path, run_driver(which, tmp_path, FRONT), log_in_locus_order=False)


@pytest.mark.gpu
@pytest.mark.parametrize("which", sorted(set(RUNS) - FRONT_NOT_COVERED))
def test_front_level_reference_driver_resident_without_f(which, tmp_path):
    """The same program WITHOUT -f (round 6): nothing but the abundances is wanted, so the unique hits never leave the device --
    the shim's preProcess ends in ONE sbgpu_quantify_resident (pass 1 on the device: the empirical insert-size law of the default
    mode, E2E_EMP; bins, weights, EM, FPKM / Frac / keep / TPM), main builds its own law from the fragment lengths the law's
    histogram gives back, procSample only prints.  The GTF and the theta log: the reference binary's, byte for byte (E2E_MASS'
    fractional masses are declined by the device grouping: that run takes the host route, and says nothing about it)."""
    need_driver(FRONT)
    r = run_driver(which, tmp_path, FRONT, with_f=False)
    check_files(which, tmp_path, r, log_in_locus_order=False, names=("out.gtf",))
    assert not os.path.exists(str(tmp_path / "ctx.tsv"))
    if which != "E2E_MASS":
        env = dict(os.environ, SBGPU_DROPIN_TIMING="1")
        sam, bam = str(tmp_path / "toy.sam"), str(tmp_path / "toy.bam")
        directory, extra, insert = RUNS[which]
        os.remove(str(tmp_path / "out.gtf"))
        cmd = [FRONT, bam, "-g", os.path.join(directory, "toy.gtf"), "-r"] + (["-i", "250/30"] if insert else []) + [
            "-o", str(tmp_path / "out.gtf"), "-T", str(tmp_path / "log2.txt")] + extra
        r2 = subprocess.run(cmd, cwd=str(tmp_path), capture_output=True, text=True, timeout=600, env=env)
        assert r2.returncode == 0 and "sbgpu_front (resident)" in r2.stderr, r2.stderr[-1500:]
        del sam


@pytest.mark.parametrize("which", sorted(RUNS))
def test_batched_loop_by_itself_reproduces_reference_files(which, tmp_path):
    """The restructured loop with the REFERENCE's EmSolver doing the solve (no GPU): what the replacement of
    Sample::procSample changes -- nothing -- checked on the CPU, apart from what the device adds."""
    need_driver(BATCHED_REFEM)
    check_files(which, tmp_path, run_driver(which, tmp_path, BATCHED_REFEM), log_in_locus_order=False)
