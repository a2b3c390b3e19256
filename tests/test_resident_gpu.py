"""The resident path end to end (sbgpu_quantify_resident, include/sbgpu.h): unique hits in HBM -> pass 1 on the device
(Sample::fragLenDist, /root/reference/src/alignments.cpp:1363-1410: the empirical insert-size law of the reference's DEFAULT
mode) -> bins -> weights -> EM -> FPKM / Frac / keep -> the FPKM all-reduce -> TPM (estimate.cpp:314-355,
alignments.cpp:1821-1829), nothing but the results crossing PCIe.

  * the reference binary's own run without -i (tests/golden/e2e_toy_emp): theta, FPKM, Frac, TPM of its log and GTF;
  * the law the device builds == the law the library's host form + InsertSize(frag_lens) build, BITWISE (mean, sd, extremes,
    histogram), on the toy and on a chain sample of 6 000 loci; theta == sbgpu_quantify_host's, bitwise;
  * FPKM / Frac / keep / TPM == the oracle's epilogue on the same theta;
  * two ranks on one GPU (the exchange through sbgpu_comm_init_host + gloo): the sharded sample's law, totals and TPM ==
    the single rank's.
"""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import e2e_util as U
import exonbin_util as XU

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RL = 75


@pytest.fixture(scope="module")
def ctx():
    from strawberry_amd import em
    return em.default_context(0)


@pytest.fixture(scope="module")
def oracle():
    from oracle import OracleLib
    return OracleLib()


def byhand_law(annot, hits, ctx):
    """The library's HOST form of pass 1 (sbgpu_exonbin_host words -> sbgpu_frag_lens_host) + InsertSize(frag_lens)."""
    from strawberry_amd import exonbin as eb
    from strawberry_amd.quantify import InsertSize
    compat, _ = eb.compat_and_keys(annot, hits, ctx)
    return InsertSize.from_frag_lens(eb.frag_lens(annot, hits, compat))


def iso_lengths(annot):
    ex = annot.exon_right.astype(np.int64) - annot.exon_left.astype(np.int64) + 1
    return np.add.reduceat(ex, annot.exon_off[:-1]).astype(np.int32)


def assert_same_law(got, want):
    assert got["use_emp"] and got["total_reads"] == want.total_reads
    assert got["start_offset"] == want.start_offset and got["end_offset"] == want.end_offset
    assert got["mean"] == want.mean and got["sd"] == want.sd          # bitwise
    np.testing.assert_array_equal(got["emp_hist"], want.emp_hist)


def test_reference_default_mode_run_through_the_resident_entry(ctx, oracle):
    """tests/golden/e2e_toy_emp = the reference binary WITHOUT -i on the toy reads: the law comes from pass 1."""
    from strawberry_amd.quantify import quantify_resident
    d = U.E2E_EMP
    ordered, rows, gtf, theta_log = U.load(d)
    annot, hits, names, _ = XU.e2e_inputs(d, ordered)
    r = quantify_resident(annot, hits, None, RL, hits.total_mapped, ctx=ctx)
    assert r["total_mapped_reads"] == rows[0]["total_mapped"] == hits.total_mapped
    want = byhand_law(annot, hits, ctx)
    assert_same_law(r["insert"], want)
    assert r["n_frag_lens"] == want.total_reads > 1000
    for l, ref_theta in enumerate(theta_log):
        th = r["theta"][annot.iso_off[l]:annot.iso_off[l + 1]]
        assert np.abs(th - np.array(ref_theta)).max() < 1e-6, (names[l], th, ref_theta)
    tx_names = [t for g in names for t, _ in ordered[g]]
    assert set(gtf) == set(t for t, k in zip(tx_names, r["keep"]) if k)
    for t, f, fr, tp, k in zip(tx_names, r["fpkm"], r["frac"], r["tpm"], r["keep"]):
        if not k:
            continue
        assert abs(f - float(gtf[t][0])) <= 1e-5 * max(1.0, f), t       # (the bar is 1e-4 relative)
        assert abs(fr - float(gtf[t][1])) < 2e-6, t
        assert abs(tp - float(gtf[t][2])) <= 1e-5 * max(1.0, tp), t
    assert abs(r["tpm"].sum() - 1e6) < 1e-3 and abs(r["total_fpkm"] - r["fpkm"][r["keep"] != 0].sum()) <= 1e-9 * r["total_fpkm"]


def test_a_given_law_and_the_other_entries(ctx, oracle):
    """-i mode through the resident entry == sbgpu_quantify_host's theta bitwise + the oracle's epilogue; the host and device
    entries without a law build the same law as the resident one."""
    from strawberry_amd import exonbin as eb
    from strawberry_amd import synth
    from strawberry_amd.quantify import InsertSize, quantify_host, quantify_resident
    loci = synth.make_gene_models(80, seed=71)
    hl, pairs = synth.make_fragments(loci, 120, seed=72, noise=0.2)
    rows = [(l, eb.hit_features(lb, rb)) for l, (lb, rb) in zip(hl, pairs)]
    rows = [(l, f) for l, f in rows if f is not None]
    annot = eb.Annotation(loci)
    hits = eb.Hits([l for l, _ in rows], [f for _, f in rows])
    for insert in (InsertSize(250.0, 30.0), None):
        h = quantify_host(annot, hits, insert, RL, ctx=ctx)
        for frac in (0.0, 0.05):
            r = quantify_resident(annot, hits, insert, RL, hits.n_hits, ctx=ctx, min_isoform_frac=frac)
            np.testing.assert_array_equal(r["theta"], h["theta"])
            np.testing.assert_array_equal(r["status"], h["status"])
            np.testing.assert_array_equal(r["iters"], h["iters"])
            want = oracle.abundance(annot.iso_off, h["theta"], h["status"], iso_lengths(annot), hits.n_hits, min_isoform_frac=frac)
            np.testing.assert_array_equal(r["keep"], want["keep"])
            np.testing.assert_allclose(r["fpkm"], want["fpkm"], rtol=1e-14, atol=0)
            np.testing.assert_allclose(r["frac"], want["frac"], rtol=1e-14, atol=0)
            np.testing.assert_allclose(r["tpm"], want["tpm"], rtol=1e-12, atol=0)
        if insert is None:
            want_law = byhand_law(annot, hits, ctx)
            assert_same_law(r["insert"], want_law)
            assert_same_law(h["insert"], want_law)
    # a sample in which no hit fits exactly one transcript has no empirical law: "Not enough reads" (read.cpp:241-245)
    from strawberry_amd import _lib
    none = eb.Hits([], [])
    with pytest.raises(_lib.SbgpuError, match="Not enough reads"):
        quantify_resident(annot, none, None, RL, 1, ctx=ctx)


def test_chain_scale_empirical_device_equals_host(ctx, oracle):
    """6 000 loci / 2.2e7 read pairs resident in HBM, no law given: the device's pass 1 == the host form bitwise, theta ==
    sbgpu_quantify_host's on the same hits bitwise (the host entry uploads the hits and takes the same pass 1)."""
    from strawberry_amd import chain
    from strawberry_amd.quantify import quantify_host
    q = chain.ChainQuantifier(ctx, n_loci=6000, n_frags=2.2e7, seed=17, resident=True, empirical=True)
    q.step()
    assert q.law["use_emp"] and q.total_mapped_reads == q.n_frags
    hits = q.hits.host_hits(q.n_loci)
    want = byhand_law(q.annot, hits, ctx)
    assert_same_law(q.law, want)
    assert 200 < want.mean < 300 and want.total_reads > 0.3 * q.n_hits
    h = quantify_host(q.annot, hits, None, q.read_len, ctx=ctx)
    assert_same_law(h["insert"], want)
    np.testing.assert_array_equal(q.theta[:q.n_iso], h["theta"])
    np.testing.assert_array_equal(q.status[:q.n_loci], h["status"])
    np.testing.assert_array_equal(q.iters[:q.n_loci], h["iters"])
    o = oracle.abundance(q.annot.iso_off, h["theta"], h["status"], iso_lengths(q.annot), q.n_frags, min_isoform_frac=0.0)
    np.testing.assert_array_equal(q.keep[:q.n_iso], o["keep"])
    np.testing.assert_allclose(q.fpkm[:q.n_iso], o["fpkm"], rtol=1e-14, atol=0)
    np.testing.assert_allclose(q.tpm[:q.n_iso], o["tpm"], rtol=1e-12, atol=0)
    # and the law matters: the same hits under -i 250/30 give another theta
    q2 = chain.ChainQuantifier(ctx, n_loci=6000, n_frags=2.2e7, seed=17, resident=True, empirical=False)
    q2.step()
    assert not q2.law["use_emp"] and not np.array_equal(q2.theta[:q.n_iso], q.theta[:q.n_iso])
    rel = np.abs(q2.theta[:q.n_iso] - q.theta[:q.n_iso]) / np.maximum(q.theta[:q.n_iso], 1.0)
    assert np.median(rel) < 0.05
    q.close(), q2.close()


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch
    sys.path.insert(0, %(root)r)
    from strawberry_amd import dist, em, front
    rank, world, _ = dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    ctx = em.Context(0)
    comm = dist.HostComm(ctx, rank, world)
    q = front.FrontQuantifier(ctx, n_loci=%(n_loci)d, n_frags=%(n_frags)g, seed=44, loci_subset=(rank, world), resident=True, empirical=True, comm=comm)
    q.step()
    q.step()
    assert comm.calls == 2 * 3, comm.calls       # per step: max (histogram length) + sum (histogram, mapped reads) + sum (FPKM total)
    np.savez(os.path.join(%(out)r, "rank%%d.npz" %% rank), tpm=q.tpm[:q.n_iso], fpkm=q.fpkm[:q.n_iso], frac=q.frac[:q.n_iso],
             theta=q.theta[:q.n_iso], keep=q.keep[:q.n_iso], status=q.status[:q.n_loci], iters=q.iters[:q.n_loci],
             total_mapped=q.total_mapped_reads, total_fpkm=q.total_fpkm, mean=q.law["mean"], sd=q.law["sd"], hist=q.law["emp_hist"],
             start=q.law["start_offset"], n=q.law["total_reads"])
    q.close()
    dist.barrier()
""")


def test_records_to_tpm_sharded_over_two_ranks(tmp_path, ctx):
    """BAM records -> TPM in the reference's default mode, ONE sample sharded by cluster over two ranks (two processes on the one
    GPU; their exchange is the caller's: sbgpu_comm_init_host over gloo).  Checked against the two shards run ONE AT A TIME in
    this process without any exchange: the sharded run's law must be the law of the two shards' histograms added up, its
    mapped-read total their sum; and with THAT law and total given, each shard alone must reproduce the rank's theta, FPKM,
    Frac and keep bit for bit and -- after dividing by the two shards' FPKM totals added up -- its TPM."""
    from strawberry_amd import front
    from strawberry_amd.quantify import InsertSize
    n_loci, n_frags, port = 2400, 2.4e6, 29652
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "out": str(tmp_path), "n_loci": n_loci, "n_frags": n_frags})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SB_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    two = [np.load(tmp_path / ("rank%d.npz" % k)) for k in range(2)]
    # the shards alone: their own laws and totals
    alone = []
    for k in range(2):
        q = front.FrontQuantifier(ctx, n_loci=n_loci, n_frags=n_frags, seed=44, loci_subset=(k, 2), resident=True, empirical=True)
        q.step()
        alone.append((q, dict(q.law), q.total_mapped_reads))
    lo = min(a[1]["start_offset"] for a in alone)
    hi = max(a[1]["end_offset"] for a in alone)
    hist = np.zeros(hi - lo + 1)
    for _, law, _ in alone:
        hist[law["start_offset"] - lo:law["end_offset"] - lo + 1] += law["emp_hist"]
    want = InsertSize.from_hist(lo, hist)
    mapped = sum(a[2] for a in alone)
    for z in two:
        assert int(z["total_mapped"]) == mapped and int(z["n"]) == want.total_reads and int(z["start"]) == want.start_offset
        assert float(z["mean"]) == want.mean and float(z["sd"]) == want.sd
        np.testing.assert_array_equal(z["hist"], want.emp_hist)
    # each shard alone under the sample's law and total
    totals = []
    for k, (q, _, _) in enumerate(alone):
        q.set_law(want)
        q.mapped_override = mapped
        q.step()
        z = two[k]
        for name in ("theta", "fpkm", "frac"):
            np.testing.assert_array_equal(getattr(q, name)[:q.n_iso], z[name], err_msg=name)
        np.testing.assert_array_equal(q.keep[:q.n_iso], z["keep"])
        np.testing.assert_array_equal(q.iters[:q.n_loci], z["iters"])
        np.testing.assert_array_equal(q.status[:q.n_loci], z["status"])
        totals.append(q.total_fpkm)
    total = totals[0] + totals[1]
    tpm_sum = 0.0
    for k, (q, _, _) in enumerate(alone):
        z = two[k]
        assert abs(float(z["total_fpkm"]) - total) <= 1e-12 * total
        np.testing.assert_allclose(z["tpm"], np.where(z["keep"] != 0, 1e6 * z["fpkm"] / total, 0.0), rtol=1e-12, atol=0)
        tpm_sum += z["tpm"].sum()
        q.close()
    assert abs(tpm_sum - 1e6) < 1e-3


def test_pass_one_on_loci_of_many_isoforms(ctx):
    """The device's pass 1 reads the compat words of a hit as its locus' isoform count says: loci of 33-70 isoforms have two and
    three words per hit (bits beyond the locus' last isoform are not isoforms), waves whose 64 hits straddle loci gather instead
    of walking one locus' table, and a locus of more than 63 isoforms or 64 exons takes the scalar walk: the law must be the host
    form's, bitwise, on an annotation that mixes them with small loci."""
    from strawberry_amd import exonbin as eb
    from strawberry_amd.quantify import quantify_resident
    rng = np.random.default_rng(314)
    loci, base = [], 10000
    for l in range(60):
        niso = int(rng.choice([1, 2, 3, 33, 40, 64, 65, 70])) if l % 3 == 0 else int(rng.integers(1, 6))
        n_ex = int(rng.integers(3, 9))
        starts = base + np.sort(rng.choice(np.arange(0, 6000, 150), n_ex, replace=False))
        exons = [(int(s), int(s) + int(rng.integers(80, 140))) for s in starts]
        isos = []
        for j in range(niso):
            keep = [e for k, e in enumerate(exons) if k in (0, n_ex - 1) or rng.random() < 0.7]
            if j % 2:            # a shortened first exon: another isoform with the same inner structure
                keep = [(keep[0][0] + 10 + j % 30, keep[0][1])] + keep[1:]
            isos.append(keep)
        # distinct isoforms only
        uniq = []
        for i in isos:
            if i not in uniq:
                uniq.append(i)
        loci.append(uniq)
        base += 20000
    annot = eb.Annotation(loci)
    assert annot.compat_words >= 2
    hl, feats = [], []
    for l, isos in enumerate(loci):
        for _ in range(int(rng.integers(5, 120))):
            iso = isos[int(rng.integers(0, len(isos)))]
            k = int(rng.integers(0, len(iso) - 1))
            a, b = iso[k], iso[k + 1]
            lb = [(a[1] - 40, a[1]), (b[0], b[0] + 33)]          # a read spliced over the junction k | k + 1
            kk = min(k + 1 + int(rng.integers(0, 2)), len(iso) - 1)
            c = iso[kk]
            rb = [(c[1] - 60, c[1] - 10)] if kk > k + 1 else [(b[0] + 40, min(b[0] + 90, b[1]))]
            f = eb.hit_features(lb, rb)
            if f is not None:
                hl.append(l)
                feats.append(f)
    # HitCluster's order inside a locus: (left end, right end); equal fragments once
    keyed = sorted(set((hl[i], feats[i][1][0], feats[i][2][-1], tuple(map(tuple, feats[i]))) for i in range(len(hl))))
    hits = eb.Hits([k[0] for k in keyed], [tuple(list(x) for x in k[3]) for k in keyed])
    want = byhand_law(annot, hits, ctx)
    assert want.total_reads > 50
    r = quantify_resident(annot, hits, None, 75, hits.n_hits, ctx=ctx)
    assert_same_law(r["insert"], want)
    from strawberry_amd.quantify import quantify_host
    h = quantify_host(annot, hits, None, 75, ctx=ctx)
    assert_same_law(h["insert"], want)
    np.testing.assert_array_equal(r["theta"], h["theta"])
