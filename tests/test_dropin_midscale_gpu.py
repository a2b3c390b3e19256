"""The drop-in under the reference's own driver at a scale the toys do not reach: ~1 500 loci of the
chain workload's law (strawberry_amd/chain.py::DeviceSample: 1-12 exons, 1-6 isoforms, log-normal expression, noise
pairs) written out as GTF + coordinate-sorted BAM; the reference program (oracle/_ref/strawberry_ref, ~3 s) and the two
deep drop-ins -- strawberry_sbgpu_chain (ONE sbgpu_quantify_host for bins + weights + EM) and strawberry_sbgpu_front
(none of the reference's BAM handling either) -- run with the same command line.

What must hold (north star: integers and structure bit-exact, FPKM / TPM within 1e-4 relative): the files are the SAME TEXT --
every line, every string and integer identical -- and a printed number (11 characters of %f in the GTF, 12 significant
digits in the -f table) differs by ONE unit of its last printed digit at most (a weight that differs in its 16th digit
flips a rounded digit now and then), in no more than 1e-4 of all printed numbers.

Test infrastructure: the programs travel to the GPU box as oracle/_ref binaries; without them the test skips (fails with
SBGPU_REQUIRE_REF=1)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
sys.path.insert(0, os.path.join(ROOT, "tools"))

N_LOCI, N_FRAGS = 1500, 4.5e5


@pytest.fixture(scope="module")
def sample(tmp_path_factory):
    import torch
    from conftest import need_ref
    for f in ("strawberry_ref", "strawberry_sbgpu_chain", "strawberry_sbgpu_front", "sam2bam"):
        if not os.path.exists(os.path.join(REF_DIR, f)):
            need_ref("oracle/_ref/" + f)
    from oracle.lib import SAM2BAM, write_gtf_from_annotation, write_sam_from_hits
    from strawberry_amd import chain
    tmp = tmp_path_factory.mktemp("dropin")
    s = chain.DeviceSample(torch, torch.device("cuda", 0), n_loci=N_LOCI, n_frags=N_FRAGS, seed=1234)
    hits = s.host_hits(N_LOCI)
    write_gtf_from_annotation(str(tmp / "s.gtf"), s.annot, N_LOCI)
    n_rec = write_sam_from_hits(str(tmp / "s.sam"), hits)
    subprocess.check_call([SAM2BAM, str(tmp / "s.sam"), str(tmp / "s.bam")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert n_rec > 5e5
    return tmp


def run(program, tmp):
    out = tmp / program
    out.mkdir()
    cmd = [os.path.join(REF_DIR, program), str(tmp / "s.bam"), "-g", str(tmp / "s.gtf"), "-r", "-i", "250/30", "-o", str(out / "out.gtf"),
           "-T", str(out / "log.txt"), "-f", str(out / "ctx.tsv")]
    r = subprocess.run(cmd, cwd=str(out), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (program, r.stderr[-2000:])
    # (the GTF's first line is a comment holding the program's own command line)
    return open(str(out / "out.gtf")).read().split("\n", 1)[1], open(str(out / "ctx.tsv")).read()


@pytest.fixture(scope="module")
def reference_files(sample):
    return run("strawberry_ref", sample)


@pytest.mark.gpu
@pytest.mark.parametrize("program", ["strawberry_sbgpu_chain", "strawberry_sbgpu_front"])
def test_deep_dropins_write_the_reference_files_at_1500_loci(program, sample, reference_files):
    from dropin_timing import compare_text
    got = run(program, sample)
    for name, g, w in (("out.gtf", got[0], reference_files[0]), ("-f table", got[1], reference_files[1])):
        assert len(w) > 1e5, "the reference wrote next to nothing"
        ok, what = compare_text(g, w)
        assert ok, (program, name, what)
        # "... N of M printed numbers differ ..." -- or identical
        m = re.search(r"(\d+) of (\d+) printed numbers differ", what)
        if m:
            n_diff, n_num = int(m.group(1)), int(m.group(2))
            assert n_num > 1e4 and n_diff <= 1e-4 * n_num + 1, (program, name, what)
