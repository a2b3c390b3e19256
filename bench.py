#!/usr/bin/env python3
"""bench.py -- EM-to-convergence throughput of the per-locus EM hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c2u] [--scaling weak|strong]

A step = one pass of the hot path over one batch of synthetic loci that is already
resident in HBM: EM (init + run to convergence) for every locus, the FPKM/Frac
epilogue, the TPM all-reduce (N > 1) and the TPM kernel.  N > 1 is launched by
torchrun (one rank per GPU, RCCL).  Two measurements, both in the line:
  strong_scaling  BASELINE config 3, "loci sharded 1 -> 2 -> 4 -> 8": ONE batch, its loci
                  dealt to the ranks by dist.shard_loci, one all-reduce per step; value =
                  that batch's loci per second -- the headline `value` (default);
  weak_scaling    every rank holds its OWN full-size batch (seed + rank); value = the loci
                  all ranks processed per second (`--scaling weak` makes it the headline).
At N = 1 the two are the same run.

Prints ONE JSON line on rank 0.  Besides the driver's fields it carries
  roofline     the dominant kernel against the HBM roofline (algorithmic bytes /
               kernel time) and, because the loop runs on-chip, the
               FP64-VALU view of the same kernel,
  cpu_baseline the reference's EmSolver (oracle/_ref, kind "reference") or the C
               restatement (kind "port") on this box's host cores, bounded sample,
  parity       the timed batch's GPU result against that CPU run (every locus); a mismatch fails the bench.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VALU_PEAK_TF = 78.6     # vendor FP64 vector peak (256 CUs)

WORKLOADS = {
    "c3": "C3 human-scale synthetic: 60000 loci/GPU, niso~1+Geom(0.25), nrow~LogNormal(ln30,1), 2e8 fragments",
    "c2": "C2 synthetic: 10000 loci x 8 isoforms x 1000 fragments, 32 exon bins",
    "c2u": "C2-U synthetic: 2000 loci x 8 isoforms x 1000 un-binned fragments (1000 rows)",
    "c3t": "C3-T synthetic: C3 plus a human-annotation-shaped tail of 300 loci with 65-400 isoforms and 200-3000 bins "
           "(the loci a workgroup's registers do not hold: em_wide_kernel)",
    "c3-chain": "C3-scale chain: 60000 DISTINCT gene models (1-12 exons, 1-6 isoforms), ~2e8 read pairs drawn on the device "
                "(log-normal share per locus, expression per isoform, N(250,30) fragments, 10% pairs that fit fewer isoforms or none), "
                "resident in HBM, through fragment x isoform compatibility + bin keys -> bins -> (bin, isoform) pairs -> bin weights "
                "-> EM -> theta (sbgpu_quantify_device)",
    "c3-front": "C3-scale, records -> theta: the BAM alignment records of the c3-chain sample's read pairs (two records per pair, "
                "~170 bytes each: the uncompressed stream behind a BAM file's header, coordinate-sorted) resident in HBM, through "
                "sbgpu_bam_decode_device -> sbgpu_assign_reads_device -> sbgpu_pair_mates_device -> sbgpu_collapse_pairs_device -> "
                "sbgpu_quantify_device; only cluster offsets and theta come back to the host",
    "c5": "C5 synthetic: the C3 law at 4e8 fragments, bias factors 2^U(-1,1) on the weights; fp32 variant of the EM "
          "timed next to the fp64 path (tolerance sweep: tools/c5_sweep.py)",
}


def make_batch(name, rank):
    from strawberry_amd import synth
    if name == "c3":
        return synth.make_c3(seed=0x5743 + rank)
    if name == "c2":
        return synth.make_c2(seed=0x5742 + rank)
    if name == "c2u":
        return synth.make_c2(n_loci=2000, seed=0x5742 + rank, unbinned=True)
    if name == "c5":
        return synth.make_c5(seed=0x5745 + rank)
    if name == "c3t":
        return synth.make_c3t(seed=0x5743 + rank)
    raise SystemExit("unknown workload " + name)


def pmc_traffic(workload, kind, kernel_key=None):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this
    very command (tools/profile_round.sh: separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` runs,
    summarised in profiles/rNN_<workload>_pmc_summary.json).  A counter pass cannot run inside the
    bench itself, so the number is read back; None when no summary for this workload is committed.
    Unit and correction per MI355X_MICROARCH.md (HBM): the counters are in KB and on gfx950 FETCH_SIZE
    tallies 128-B requests at 64 B, so it is doubled before it is compared with a byte count."""
    import glob
    import json
    key = kernel_key or ["em_fused_kernel<0, 1,", "em_fused_kernel<0, 2,", "em_fused_kernel<0, 4,", "em_fused_kernel<4, 2,",
                         "em_fused_kernel<4, 12,", "em_wide_kernel"][kind]
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "profiles", "r*_%s_pmc_summary.json" % workload)))
    if not files:
        return None, "no PMC summary committed for this workload"
    try:
        d = json.load(open(files[-1]))
        from strawberry_amd import _lib
        mine = _lib.load().sbgpu_build_id().decode()
        theirs = d.get("_build_id")
        if theirs != mine:
            return None, "%s was taken with library build %s, this is build %s: not quoted (re-run tools/profile_round.sh)" % (
                os.path.basename(files[-1]), theirs, mine)
        # a stage may be several instantiations of one kernel template (bins_accum_kernel<1024,256,..>, <2048,1400,..>, ...), each
        # launched once per step: the stage's traffic is the SUM over the summary's entries that match, like its time
        match = [v for k, v in d.items() if isinstance(v, dict) and key in k]
        if not match:
            raise StopIteration
        fetch_kb = sum(c["FETCH_SIZE"]["mean_per_dispatch"] for c in match)
        write_kb = sum(c["WRITE_SIZE"]["mean_per_dispatch"] for c in match)
    except (StopIteration, KeyError, ValueError):
        return None, "PMC summary %s has no FETCH_SIZE/WRITE_SIZE for %s" % (os.path.basename(files[-1]), key)
    return (int((2.0 * fetch_kb + write_kb) * 1024),
            "%s: 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE = 2 x %.0f KB + %.0f KB per launch" % (
                os.path.basename(files[-1]), fetch_kb, write_kb))


def cpu_baseline(batch, gpu, budget_s=12.0):
    """Time the reference's EmSolver (or the port) on this box's host cores, on a
    bounded prefix of the same batch: one thread (the reference's deterministic mode)
    and all cores with a static locus partition (the analogue of `-p T`).  The all-core
    run solves the WHOLE batch, so it doubles as the parity check of what was just timed
    on the GPU (`gpu` = its results): status and iteration counts exact, theta to 1e-9."""
    from oracle import OracleLib, RefLib, have_ref
    kind = "reference" if have_ref() else "port"
    lib = RefLib() if have_ref() else OracleLib()
    cores = os.cpu_count() or 1

    def run(n, threads):
        sub = batch if n >= batch.n_loci else batch.select(np.arange(n))
        t = time.perf_counter()
        out = lib.em_batch(sub.row_off, sub.iso_off, sub.f_off, sub.count, sub.F, threads=threads)
        return time.perf_counter() - t, sub, out

    # calibrate on 2000 loci, then size the single-thread sample for ~budget/2 seconds
    dt, _, _ = run(min(2000, batch.n_loci), 1)
    rate1 = min(2000, batch.n_loci) / max(dt, 1e-9)
    n1 = int(min(batch.n_loci, max(2000, rate1 * budget_s * 0.5)))
    dt1, sub1, _ = run(n1, 1)
    dtN, subN, cpu = run(batch.n_loci, cores)
    if kind == "reference":
        # EmSolver exposes two bools (init, run): MAXITER is not observable, both 0 and 3 read "ran"
        theta, flags = cpu
        status = np.where((flags & 1) == 0, 1, np.where((flags & 2) == 0, 2, 0)).astype(np.int32)
        iters = None
    else:
        theta, status, iters = cpu
    out = {
        "value": n1 / dt1, "unit": "loci/s", "cores": 1, "kind": kind,
        "sample": "first %d loci of the same batch, EmSolver init+run, 1 thread, %.2f s" % (n1, dt1),
        "mfrags_per_s": sub1.n_frags / dt1 / 1e6,
        "all_cores": {"value": batch.n_loci / dtN, "unit": "loci/s", "cores": cores,
                      "sample": "whole batch (%d loci), static partition over %d threads, %.3f s" % (
                          batch.n_loci, cores, dtN)},
    }
    # parity of the timed GPU result with this CPU run
    g_st = np.where(gpu["status"] == 3, 0, gpu["status"]) if kind == "reference" else gpu["status"]
    c_st = status
    err = np.abs(gpu["theta"] - theta) / np.maximum(np.abs(theta), 1e-9)
    parity = {"checked_loci": int(batch.n_loci), "against": kind, "status_mismatches": int((g_st != c_st).sum()),
              "theta_max_rel_err": float(err.max()) if len(err) else 0.0, "tolerance": 1e-9}
    if iters is not None:
        parity["iteration_count_mismatches"] = int((gpu["iters"] != iters).sum())
    parity["ok"] = bool(parity["status_mismatches"] == 0 and parity["theta_max_rel_err"] < 1e-9 and
                        parity.get("iteration_count_mismatches", 0) == 0)
    return out, parity


def timed_steps(quant, steps, warmup, dev, sdist, torch):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize on both sides; -> (wall seconds,
    max over ranks; device milliseconds between two events on torch's stream)."""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(warmup):
        quant.step()
    ev[0].record()   # first use of a timing event on this stream happens here, not inside the timed region
    torch.cuda.synchronize(dev)
    sdist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    ev[0].record()
    for _ in range(steps):
        quant.step()
    ev[1].record()
    torch.cuda.synchronize(dev)
    timed_steps.own_wall = time.perf_counter() - t0   # this rank's K steps, before it waits for the others (diagnostic)
    sdist.barrier()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    quant.finish()   # raises if a run failed on the device
    tmax = torch.tensor([wall], dtype=torch.float64, device=dev)
    sdist.allreduce_max_(tmax)
    return float(tmax.item()), ev[0].elapsed_time(ev[1])


def chain_cpu_baseline(q, budget_frags=3.2e6):
    """The reference PROGRAM (oracle/_ref/strawberry_ref: Strawberry's own main, BAM decode, clustering, LocusContext,
    EmSolver, output -- compiled from /root/reference) on the first loci of the chain sample, one thread, timed on
    this box's host, in the mode the GPU leg ran in: its DEFAULT mode (no -i: pass 1 builds the empirical insert-size law)
    when the leg was empirical, else -i 250/30.  Parity: an empirical law is the law of the sample it was built from, so the
    GPU side of the comparison is sbgpu_quantify_resident on the SAME first loci alone (their hits, their law, their mapped-read
    total) -- theta against the theta lines of the program's log (estimate.cpp:312), FPKM and TPM against its GTF.
    The sample's hits are turned back into a coordinate-sorted BAM (our sam2bam) and a GTF of its gene models.
    -> (cpu_baseline dict, parity dict), or (None, None) where oracle/_ref is not built."""
    import re
    import subprocess
    import tempfile
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    ref_bin, sam2bam = os.path.join(ref_dir, "strawberry_ref"), os.path.join(ref_dir, "sam2bam")
    if not (os.path.exists(ref_bin) and os.path.exists(sam2bam)):
        return None, None
    a = q.annot
    K = int(np.searchsorted(q.hits.locus_hit_off, budget_frags, side="right"))
    K = max(1, min(K, q.n_loci))
    h = q.hits.host_hits(K)
    from oracle.lib import write_gtf_from_annotation, write_sam_from_hits
    with tempfile.TemporaryDirectory() as tmp:
        gtf, sam, bam = (os.path.join(tmp, n) for n in ("s.gtf", "s.sam", "s.bam"))
        write_gtf_from_annotation(gtf, a, K)
        # hits -> read pairs: the features before the GAP are the left mate's, those behind it the right mate's; every read
        # pair behind a unique hit gets its own two records (oracle/sam_writer.c)
        write_sam_from_hits(sam, h)
        subprocess.check_call([sam2bam, sam, bam], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        empirical = bool(getattr(q, "empirical", False))
        cmd = [ref_bin, bam, "-g", gtf, "-r"] + ([] if empirical else ["-i", "250/30"]) + [
            "-o", os.path.join(tmp, "out.gtf"), "-T", os.path.join(tmp, "log.txt"), "-f", os.path.join(tmp, "ctx.tsv")]
        t = time.perf_counter()
        r = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True)
        dt = time.perf_counter() - t
        if r.returncode != 0:
            return {"error": "strawberry_ref failed: " + r.stderr[-300:]}, None
        # theta per locus from the log, loci and isoform order from the -f table
        thetas = []
        for line in open(os.path.join(tmp, "log.txt")):
            m = re.match(r"isoform (\d+) has ([0-9.eE+-]+) raw read count", line)
            if m:
                if int(m.group(1)) == 1:
                    thetas.append([])
                thetas[-1].append(float(m.group(2)))
        genes = []
        for k, line in enumerate(open(os.path.join(tmp, "ctx.tsv"))):
            f = line.rstrip("\n").split("\t")
            if k and len(f) >= 10 and (not genes or genes[-1][0] != f[2]):
                genes.append((f[2], f[4].split(",")))
        ref_gtf = {}
        for line in open(os.path.join(tmp, "out.gtf")):
            f = line.rstrip("\n").split("\t")
            if len(f) >= 9 and f[2] == "transcript":
                t = re.search(r'transcript_id "([^"]+)"', f[8]).group(1)
                ref_gtf[t] = tuple(float(re.search(r'%s "([^"]+)"' % k, f[8]).group(1)) for k in ("FPKM", "TPM"))
    n_pairs = int(h.mass.sum())
    mode = "default mode: empirical insert-size law from pass 1" if empirical else "-i 250/30"
    if getattr(q, "resident", False):
        # the GPU side: the same first loci alone, resident entry (pass 1 on the device in default mode, FPKM / TPM on the device)
        from strawberry_amd.quantify import quantify_resident
        sub = quantify_resident(a.prefix(K), h, None if empirical else q.insert, q.read_len, n_pairs, ctx=q.ctx)
        g_theta, g_status = sub["theta"], sub["status"]
    else:
        sub, g_theta, g_status = None, q.theta, q.status
    base = {"value": n_pairs / dt, "unit": "fragments/s", "cores": 1, "kind": "reference",
            "sample": "the first %d loci of the same sample (%d read pairs, %d unique hits) as a BAM + GTF through the reference program "
                      "(strawberry_ref -g -r, %s: two BAM passes, clustering, bins, weights, EM, output), 1 thread, %.2f s" % (K, n_pairs, h.n_hits, mode, dt),
            "loci_per_s": K / dt}
    worst, checked, worst_tpm, worst_fpkm, n_tpm = 0.0, 0, 0.0, 0.0, 0
    ok = len(genes) == len(thetas)
    for (gid, tx), th in zip(genes, thetas):
        l = int(gid[1:])
        for t, v in zip(tx, th):
            j = int(t.split(".")[1]) - 1
            k = int(a.iso_off[l]) + j
            worst = max(worst, abs(float(g_theta[k]) - v) / max(abs(v), 1.0))
            checked += 1
            if sub is not None and t in ref_gtf:      # FPKM / TPM as the program printed them (six decimals)
                rf, rt = ref_gtf[t]
                worst_fpkm = max(worst_fpkm, abs(float(sub["fpkm"][k]) - rf) / max(abs(rf), 1.0))
                worst_tpm = max(worst_tpm, abs(float(sub["tpm"][k]) - rt) / max(abs(rt), 1.0))
                n_tpm += 1
        ok &= int(g_status[l]) in (0, 2, 3)
    parity = {"against": "reference program's theta log (printed %f)" + (", its GTF's FPKM and TPM" if sub is not None else ""), "mode": mode,
              "loci_checked": len(genes), "isoforms_checked": checked,
              "theta_max_err": worst, "tolerance": 2e-6, "ok": bool(ok and checked > 0 and worst < 2e-6)}
    if sub is not None:
        parity.update({"gpu_side": "sbgpu_quantify_resident on the same first loci alone (their own law and mapped-read total)",
                       "transcripts_checked_in_gtf": n_tpm, "fpkm_max_rel_err": worst_fpkm, "tpm_max_rel_err": worst_tpm, "tpm_tolerance": 1e-4,
                       "law": {k: v for k, v in sub["insert"].items() if k != "emp_hist"}, "mapped_reads": sub["total_mapped_reads"]})
        parity["ok"] = bool(parity["ok"] and n_tpm > 0 and worst_tpm < 1e-4 and worst_fpkm < 1e-4)
    return base, parity


def resident_comm(ctx, rank, world, comm, sdist):
    """The communicator sbgpu_quantify_resident exchanges over: the C ABI's RCCL one where bench.py made it, else (several ranks on
    one device, SB_COMM=torch) the caller's exchange through torch.distributed (sbgpu_comm_init_host); None for one rank."""
    if world == 1:
        return None
    return comm if comm is not None else sdist.HostComm(ctx, rank, world)


def chain_leg(args, ctx, dev, rank, world, sdist, torch, strong=False, with_cpu=True, comm=None):
    """The fragments -> abundances chain (sbgpu_quantify_device) on the chain sample (strawberry_amd/chain.py): 60 000
    distinct gene models, ~2e8 read pairs resident in HBM.  Weak scaling: every rank its own sample; strong: ONE
    sample, locus l on rank l mod world (no locus data crosses ranks; the caller's FPKM total is the one collective).
    -> dict for the JSON line (rank 0), None on the other ranks."""
    from strawberry_amd import chain
    n_frags = float(os.environ.get("SB_CHAIN_FRAGS", "2e8"))
    n_loci = int(float(os.environ.get("SB_CHAIN_LOCI", "60000")))
    sub = (rank, world) if (strong and world > 1) else None
    # the reference's DEFAULT mode end to end (SB_CHAIN_INSERT=given: -i 250/30): pass 1 on the device, the law and the mapped-read
    # total all-reduced, bins, weights, EM, FPKM / Frac, the FPKM all-reduce, TPM -- sbgpu_quantify_resident, one call per step
    empirical = os.environ.get("SB_CHAIN_INSERT", "empirical") != "given"
    q = chain.ChainQuantifier(ctx, n_loci=n_loci, n_frags=n_frags, seed=31 + (0 if strong else rank), loci_subset=sub, resident=True,
                              empirical=empirical, comm=resident_comm(ctx, rank, world, comm, sdist))
    wall, _ = timed_steps(q, args.steps, args.warmup, dev, sdist, torch)
    counts = torch.tensor([q.n_loci, q.n_frags, q.n_hits], dtype=torch.int64, device=dev)
    sdist.allreduce_sum_(counts)
    per_rank = None
    if world > 1:   # every rank's own K steps (before it waits for the others), its loci and unique hits
        per_rank = sdist.gather_values([timed_steps.own_wall / args.steps * 1e3, q.n_loci, q.n_hits], rank, world, device=dev)
    stage = q.stage_ms()      # (one more step, EVERY rank: the step's collectives are inside sbgpu_quantify_resident)
    if rank != 0:
        q.close()
        return None
    ms = wall / args.steps * 1e3
    feats = q.hits.n_features / max(q.n_hits, 1)
    cw, kw = q.annot.compat_words, q.annot.key_words
    info = q.info or {}
    # algorithmic bytes of the kernel stages: what each must read and write once
    alg = {
        "exonbin_kernel": q.n_hits * (9.0 * feats + 12.0 + 4.0 * (cw + kw) + 12.0),      # features, offsets, locus; words, span + hash out
        "bins_accum_kernel": q.n_hits * (4.0 * (cw + kw) + 4.0 + 8.0 + 4.0 + 1.0) + info.get("n_bins", 0) * (8.0 + 4.0 * cw),
        "binweight_kernel": info.get("n_pairs", 0) * (8.0 + 4.0 + 4.0 + 8.0 + 8.0 + 4.0 * 3.0),  # offsets, mask, length, index, F out, ~3 segments
    }
    dom = max((k for k in stage if k in alg), key=lambda k: stage[k], default=None)
    roof = None
    if dom:
        ach = alg[dom] / (stage[dom] * 1e-3) / 1e9
        traffic, tnote = pmc_traffic("c3chain", 0, kernel_key="sb::" + dom)
        roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_note": tnote, "kernel_ms": stage[dom], "algorithmic_bytes": int(alg[dom]),
                "note": "kernel time from HIP events on the kernels' stream (sbgpu_last_stage_ms)"}
    out = {
        "workload": WORKLOADS["c3-chain"], "scaling": "strong" if strong else "weak",
        "insert": "empirical" if empirical else "given (-i 250/30)", "tpm": True,
        "law": {k: v for k, v in (q.law or {}).items() if k != "emp_hist"}, "total_mapped_reads": q.total_mapped_reads,
        "tpm_sum_this_rank": float(q.tpm[:q.n_iso].sum()),
        "collectives_per_step": ("none (one rank)" if world == 1 else
                                 ("all-reduce(max) of the histogram's length + all-reduce(sum) of the fragment-length histogram with the mapped-read total"
                                  if empirical else "all-reduce(sum) of the mapped-read total") + " + all-reduce(sum) of the FPKM total, inside sbgpu_quantify_resident"),
        "loci": int(counts[0]), "fragments": int(counts[1]), "unique_hits": int(counts[2]), "features_per_hit": feats,
        "ms_per_step": ms, "loci_per_s": int(counts[0]) * args.steps / wall, "gfrags_per_s": int(counts[1]) * args.steps / wall / 1e9,
        "kernel_ms": stage, "kernels_sum_ms": float(sum(stage.values())),
        "host_and_gaps_ms": ms - float(sum(stage.values())),
        "shape": info,
        "em_status": {"ok": int((q.status[:q.n_loci] == 0).sum()), "init_empty": int((q.status[:q.n_loci] == 1).sum()),
                      "denom_zero": int((q.status[:q.n_loci] == 2).sum()), "maxiter": int((q.status[:q.n_loci] == 3).sum()),
                      "mean_iters": float(q.iters[:q.n_loci].mean())},
        "roofline": roof,
    }
    if per_rank is not None:
        out.update({"per_rank_ms": [float(x) for x in per_rank[:, 0]], "slowest_rank": int(per_rank[:, 0].argmax()),
                    "loci_per_rank": [int(x) for x in per_rank[:, 1]], "unique_hits_per_rank": [int(x) for x in per_rank[:, 2]]})
    if with_cpu and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"], out["parity"] = chain_cpu_baseline(q)
    if with_cpu and world == 1 and os.environ.get("SB_CHAIN_PCIE", "1") == "1":
        # the same sample through the HOST entry (sbgpu_quantify_host): hits in pageable host memory, uploaded inside the call.
        # Never the headline (`value` is the resident number); reported beside it.
        pc = q.host_entry(reps=2)
        pc["note"] = ("sbgpu_quantify_host on the same sample: %.1f GB of hits uploaded from pageable host memory per call, same kernels, "
                      "theta downloaded; whole call = %.0f ms against %.1f ms with the hits resident" % (pc["hit_bytes"] / 1e9, pc["ms_per_call"], ms))
        out["pcie_inclusive"] = pc
    return out


def front_leg(args, ctx, dev, rank, world, sdist, torch, comm=None):
    """The records -> TPM leg of the default line (strawberry_amd/front.py, `--workload c3-front` is its own line): the chain
    sample's BAM alignment records resident in HBM -> decode -> read stream -> pairs -> unique hits -> pass 1 (the empirical
    insert-size law: the reference's default mode) -> bins -> weights -> EM -> FPKM -> TPM.  N > 1: ONE sample, locus l (its cluster
    and its records) on rank l mod N; the law's histogram, the mapped-read total and the FPKM total are all-reduced inside the last
    stage of every step.  The ranks agree beforehand that each of them holds its shard (a rank that could not make it hands in a
    zero and nobody steps); the gather at the end is reached by every rank whatever happened to it."""
    from strawberry_amd import front
    stages = front.FrontQuantifier.STAGES
    vals, note, q = [0.0] * (5 + len(stages)), None, None
    law, totals = None, (None, None, None)
    n_loci = int(float(os.environ.get("SB_FRONT_LOCI", "60000")))
    n_frags = float(os.environ.get("SB_FRONT_FRAGS", "2e8"))
    steps = max(1, min(args.steps, 5))
    try:
        torch.cuda.empty_cache()
        ctx.L.sbgpu_release_idle_memory()
        need = 2.8 * 2 * 175.0 * n_frags / world      # the shard's records, the library's arenas for them, the packer's temporaries
        free = torch.cuda.mem_get_info(dev)[0]
        if free < need:
            note = "skipped: %.0f GB free on the device, %.0f GB wanted" % (free / 1e9, need / 1e9)
        else:
            q = front.FrontQuantifier(ctx, n_loci=n_loci, n_frags=n_frags, seed=31, loci_subset=(rank, world) if world > 1 else None,
                                      resident=True, empirical=True, comm=resident_comm(ctx, rank, world, comm, sdist))
            torch.cuda.empty_cache()
    except Exception as e:      # (memory, a launch that fails: the default line must come out all the same)
        note = "failed: %r" % (e,)
        q = None
    # The steps are collective from here on (the law, the mapped-read total and the FPKM total are all ranks'): the ranks first
    # agree that every one of them holds its shard -- a rank that could not make it must not leave the others in an all-reduce
    ready = sdist.gather_values([1.0 if q is not None else 0.0], rank, world, device=dev)
    all_ready = bool((ready[:, 0] > 0).all())
    try:
        if q is not None and all_ready:
            for _ in range(2):
                q.step()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                q.step()
            torch.cuda.synchronize(dev)
            ms = (time.perf_counter() - t0) / steps * 1e3
            st = dict(q.stage_wall_ms)
            law = {k: v for k, v in q.law.items() if k != "emp_hist"}
            totals = (q.total_mapped_reads, q.total_fpkm, float(q.tpm[:q.n_iso].sum()))
            ok = bool(q.compare_with_chain()["ok"])
            vals = [1.0, 1.0 if ok else 0.0, ms, float(q.n_loci), float(q.n_records)] + [float(st[k]) for k in stages]
        elif q is not None:
            note = "skipped: another rank could not hold its shard"
    except Exception as e:      # noqa: BLE001
        note = "failed: %r" % (e,)
    finally:
        try:
            if q is not None:
                q.close()
            del q
            torch.cuda.empty_cache()
            ctx.L.sbgpu_release_idle_memory()
        except Exception:
            pass
    table = sdist.gather_values(vals, rank, world, device=dev)
    if rank != 0:
        return None
    ran = [int(r) for r in range(world) if table[r][0] > 0]
    out = {"workload": WORKLOADS["c3-front"], "scaling": "strong", "steps": steps, "ranks_that_ran": ran, "insert": "empirical", "tpm": True}
    if note:
        out["note_rank0"] = note[:300]
    if len(ran) == world:
        ms_all = [float(table[r][2]) for r in range(world)]
        step = max(ms_all)
        out.update({
            "ms_per_step": step, "per_rank_ms": ms_all, "slowest_rank": int(np.argmax(ms_all)),
            "loci": int(table[:, 3].sum()), "records": int(table[:, 4].sum()),
            "loci_per_s": float(table[:, 3].sum()) / (step * 1e-3), "grecords_per_s": float(table[:, 4].sum()) / (step * 1e-3) / 1e9,
            "loci_per_rank": [int(x) for x in table[:, 3]], "records_per_rank": [int(x) for x in table[:, 4]],
            "per_rank_stage_ms": [dict(zip(stages, (float(x) for x in table[r][5:]))) for r in range(world)],
            "parity_with_chain": {"ranks_ok": int(table[:, 1].sum()), "ok": int(table[:, 1].sum()) == world},
            "law": law, "total_mapped_reads": totals[0], "total_fpkm": totals[1], "tpm_sum_rank0": totals[2],
            "timing": "every rank its own K steps; the last stage (sbgpu_quantify_resident) exchanges the law's histogram, the mapped-read total and "
                      "the FPKM total with the other ranks inside the step; the step is the slowest rank's",
        })
    return out


def chain_main(args, ctx, dev, rank, world, sdist, torch, launch, comm=None):
    """--workload c3-chain: the chain is the headline of the line."""
    c = chain_leg(args, ctx, dev, rank, world, sdist, torch, strong=args.scaling == "strong", comm=comm)
    if rank != 0:
        return
    out = {
        "metric": "loci/s and G fragments/s, fragments -> abundances chain (C3-scale)", "value": c["loci_per_s"],
        "unit": "loci/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": c["ms_per_step"],
        "higher_is_better": True, "scaling": c["scaling"], "vs_baseline": None, "dtype": "u32 intervals + f64", "data": "synthetic",
        "mfrags_per_s": c["gfrags_per_s"] * 1e3,
        "config": {"workload": WORKLOADS["c3-chain"], "loci": c["loci"], "fragments": c["fragments"],
                   "sharding": "locus l on rank l mod N of ONE sample" if c["scaling"] == "strong" else "every rank its own sample"},
        "launch": launch, "chain": c, "roofline": c["roofline"],
    }
    if c.get("cpu_baseline"):
        out["cpu_baseline"], out["parity"] = c["cpu_baseline"], c.get("parity")
    print(json.dumps(out))
    if c.get("parity") and not c["parity"]["ok"]:
        raise SystemExit("bench.py: the chain's theta does not match the reference program's: %r" % c["parity"])


def front_main(args, ctx, dev, rank, world, sdist, torch, launch, comm=None):
    """--workload c3-front: alignment records in HBM -> theta, every stage of SURVEY 8(f) rank 4 in front of the chain
    (strawberry_amd/front.py).  A step = one pass over ALL records of the sample.  Size: SB_FRONT_LOCI / SB_FRONT_FRAGS
    (default: the chain sample, 60 000 loci / 2e8 read pairs = ~3.9e8 records, ~66 GB of record bytes)."""
    from strawberry_amd import front
    n_loci = int(float(os.environ.get("SB_FRONT_LOCI", "60000")))
    n_frags = float(os.environ.get("SB_FRONT_FRAGS", "2e8"))
    # N > 1: ONE sample, locus l (= cluster l) on rank l mod N with its records -- the stages work cluster by cluster, nothing
    # but the step's barrier couples the ranks (strong scaling, like c3-chain's)
    q = front.FrontQuantifier(ctx, n_loci=n_loci, n_frags=n_frags, seed=31, loci_subset=(rank, world) if world > 1 else None,
                              resident=True, empirical=os.environ.get("SB_CHAIN_INSERT", "empirical") != "given",
                              comm=resident_comm(ctx, rank, world, comm, sdist))
    torch.cuda.empty_cache()      # the packer's temporaries go back to the driver: the library allocates for itself
    wall, _ = timed_steps(q, args.steps, args.warmup, dev, sdist, torch)
    ms = wall / args.steps * 1e3
    if world > 1:
        with_chain = q.compare_with_chain()
        tot = torch.tensor([q.n_loci, q.n_records, q.n_frags, 1 if with_chain["ok"] else 0], dtype=torch.float64, device=dev)
        sdist.allreduce_sum_(tot)
        per_rank = sdist.gather_values([timed_steps.own_wall / args.steps * 1e3, q.n_loci, q.n_records] + [q.stage_wall_ms[k] for k in front.FrontQuantifier.STAGES],
                                       rank, world, device=dev)
        if rank != 0:
            return
        loci, recs, frags, oks = (float(x) for x in tot.tolist())
        out = {
            "metric": "loci/s and G records/s, BAM alignment records -> abundances (C3-scale)", "value": loci * args.steps / wall,
            "unit": "loci/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u8 records, u32 intervals, f64", "data": "synthetic",
            "grecords_per_s": recs * args.steps / wall / 1e9, "mfrags_per_s": frags * args.steps / wall / 1e6,
            "config": {"workload": WORKLOADS["c3-front"], "loci": int(loci), "read_pairs": int(frags), "records": int(recs),
                       "sharding": "locus l (its cluster and its records) on rank l mod N of ONE sample"},
            "insert": "empirical" if q.empirical else "given (-i 250/30)", "tpm": True,
            "law": {k: v for k, v in q.law.items() if k != "emp_hist"}, "total_mapped_reads": q.total_mapped_reads, "total_fpkm": q.total_fpkm,
            "launch": launch, "per_rank_ms": [p[0] for p in per_rank], "slowest_rank": int(np.argmax([p[0] for p in per_rank])),
            "loci_per_rank": [int(p[1]) for p in per_rank], "records_per_rank": [int(p[2]) for p in per_rank],
            "per_rank_stage_ms": [dict(zip(front.FrontQuantifier.STAGES, p[3:])) for p in per_rank],
            "parity_with_chain": {"ranks_ok": int(oks), "ok": int(oks) == world,
                                  "what": "every rank: theta / status / iterations of its records -> theta pass against sbgpu_quantify_device on its "
                                          "shard's own unique hits, bit for bit wherever the span filter dropped no pair"},
        }
        print(json.dumps(out))
        if int(oks) != world:
            raise SystemExit("bench.py: a rank's records -> theta pass does not reproduce the chain's theta on its shard")
        return
    probe = []
    for _ in range(3):
        q.step()
        probe.append(dict(q.stage_wall_ms))
    stage = {k: float(np.median([p[k] for p in probe])) for k in front.FrontQuantifier.STAGES}
    chain_stage = q.stage_ms()    # the chain's kernels inside the last stage (HIP events)
    filtered = None
    if not args.no_cpu_baseline:
        # the loci the reference's span filter touched, at full size, against the oracle's collapse (and the chain on its hits)
        from oracle import OracleLib
        filtered = q.check_filtered_loci(OracleLib())
    with_chain = q.compare_with_chain()
    c = q.counts
    n_rec, acc, feats = c["records"], c["accepted_records"], c["features"]
    blocks = acc * 1.3
    # algorithmic bytes per stage: what each must read and write once
    alg = {
        "bam_decode": q.n_bytes + 8.0 * (n_rec + 1) + n_rec + acc * (8 + 8 + 4 * 8 + 1 + 8) + blocks * 8,
        "assign_reads": acc * (4 + 4 + 4 + 1 + 4 + 1),
        "pair_mates": acc * (8 + 8 + 4 + 1 + 4) + blocks * 8 + (acc / 2) * (8 + 16) + blocks * 2 * 9.0,
        "collapse_pairs": (acc / 2) * (8 + 16) + blocks * 2 * 9.0 + c["unique_hits"] * (4 + 8 + 4) + feats * 9.0,
        "quantify": 9.0 * feats + q.n_hits * (24.0 + 4.0 * (q.annot.compat_words + q.annot.key_words)),   # its exon-bin kernel's bytes
    }
    dom = max(stage, key=lambda k: stage[k])
    ach = alg[dom] / (stage[dom] * 1e-3) / 1e9
    # counter traffic of the dominant stage's OWN kernels (profiles/rNN_c3front_pmc_summary.json, quoted only for this build); its
    # device-wide sorts and scans are rocPRIM kernels whose names the stages share, so they are not attributed to a stage
    own = {"bam_decode": ["sb::bam_scan_kernel", "sb::bam_fill_kernel"], "assign_reads": ["sb::assign_reads_kernel", "sb::cluster_bounds_kernel"],
           "pair_mates": ["sb::flat_mate_"], "collapse_pairs": ["sb::flat_keys_kernel", "sb::flat_flags_kernel", "sb::flat_heads_kernel", "sb::flat_fill_kernel",
                                                                "sb::flat_mass", "sb::flat_gather_kernel", "sb::flat_sd_kernel"],
           "quantify": ["sb::exonbin_kernel", "sb::bins_", "sb::binweight_kernel", "sb::em_"]}[dom]
    traffic, tnote = 0, None
    for key in own:
        t, note = pmc_traffic("c3front", 0, kernel_key=key)
        if t is None:
            traffic, tnote = None, note
            break
        traffic += t
    if traffic is not None:
        tnote = "the stage's own kernels (%s); its rocPRIM sorts and scans not included" % ", ".join(own)
    roof = {"bound": "hbm", "kernel": "sbgpu_%s%s (whole call: its kernels, device-wide sorts and scans, and the host synchronisations between them)" % (
                dom, "" if dom == "quantify" else "_device"),
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_note": tnote, "stage_ms": stage[dom], "algorithmic_bytes": int(alg[dom]),
            "per_stage": {k: {"ms": stage[k], "algorithmic_bytes": int(alg[k]), "GBps": alg[k] / (stage[k] * 1e-3) / 1e9,
                              "frac": alg[k] / (stage[k] * 1e-3) / 1e9 / HBM_PEAK_GBS} for k in stage},
            "note": "stage times are wall clock around synchronised calls (every stage's totals size the next one's arrays)"}
    out = {
        "metric": "loci/s and G records/s, BAM alignment records -> abundances (C3-scale)", "value": q.n_loci * args.steps / wall,
        "unit": "loci/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8 records, u32 intervals, f64", "data": "synthetic",
        "grecords_per_s": n_rec * args.steps / wall / 1e9, "mfrags_per_s": q.n_frags * args.steps / wall / 1e6,
        "config": {"workload": WORKLOADS["c3-front"], "loci": q.n_loci, "read_pairs": q.n_frags, "records": n_rec,
                   "record_bytes": q.n_bytes, "bytes_per_record": q.n_bytes / max(n_rec, 1), "unique_hits": c["unique_hits"],
                   "records_packed_in_s": q.pack_s},
        "launch": launch, "stage_ms": stage, "stages_sum_ms": float(sum(stage.values())),
        "insert": "empirical" if q.empirical else "given (-i 250/30)", "tpm": True,
        "law": {k: v for k, v in q.law.items() if k != "emp_hist"}, "total_mapped_reads": q.total_mapped_reads, "total_fpkm": q.total_fpkm,
        "tpm_sum": float(q.tpm[:q.n_iso].sum()),
        "chain_kernel_ms": chain_stage, "shape": q.info, "counts": c,
        "em_status": {"ok": int((q.status[:q.n_loci] == 0).sum()), "init_empty": int((q.status[:q.n_loci] == 1).sum()),
                      "denom_zero": int((q.status[:q.n_loci] == 2).sum()), "maxiter": int((q.status[:q.n_loci] == 3).sum()),
                      "mean_iters": float(q.iters[:q.n_loci].mean())},
        "roofline": roof,
        "parity_of_the_span_filtered_loci": dict(filtered or {}, what="the clusters in which the reference's span filter (alignments.cpp:666-682) dropped pairs: "
                                                 "the sample's pairs through oracle/collapse_oracle.c, then sbgpu_quantify_resident on the oracle's unique hits of those loci alone under "
                                                 "the pass' law and total: hit counts and theta / status / iterations / FPKM / Frac / keep of the records -> TPM pass, bit for bit"),
        "parity_with_chain": dict(with_chain, what="theta / status / iterations of the records -> theta pass against sbgpu_quantify_device on the "
                                  "sample's own unique hits (c3-chain), locus by locus: bit for bit wherever the reference's span filter "
                                  "(alignments.cpp:666-682) dropped no pair", unique_hits_front=c["unique_hits"], unique_hits_sample=q.n_hits),
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"], out["parity"] = chain_cpu_baseline(q)
    print(json.dumps(out))
    if not with_chain["ok"]:
        raise SystemExit("bench.py: the records -> theta pass does not reproduce the chain's theta on the same sample")
    if filtered is not None and not filtered["ok"]:
        raise SystemExit("bench.py: the span-filtered loci do not match the oracle: %r" % (filtered,))
    if out.get("parity") and not out["parity"]["ok"]:
        raise SystemExit("bench.py: theta does not match the reference program's: %r" % out["parity"])


def front_host_main(args, ctx, dev, torch, launch):
    """--workload c3-front --from-host: the same records -> TPM pass for a caller that holds the inflated record stream in HOST
    memory (page-locked where the box allows): sbgpu_front_stream_begin / push / end (include/sbgpu.h) -- chunks of whole
    records (SB_FRONT_CHUNK_MB, default 512) uploaded while the chunk before is decoded, paired, collapsed; one
    sbgpu_quantify_resident over the unique hits at the end.  The device never holds more than two chunk buffers, a chunk's
    arenas and the unique hits.  Reported: the pass' wall time against the upload alone (the same chunks copied to the device and
    nothing else) and against the resident pass (the compute alone), its peak device memory, and that its results equal the
    resident pass' bit for bit.  BGZF inflate is the caller's and not in any of the numbers.
    A sample of more than 2.4e8 read pairs (SB_FRONT_FRAGS=4e8: BASELINE config 5's size) cannot even be PACKED on the device in
    one piece: it is made as two samples of half the size on references 0 and 1, brought to the host one after the other, and pushed
    as ONE stream of 120 000 clusters; its parity is theta / status / iterations of a given-law pass against the two halves'
    resident passes (with a given law the loci are independent), its timing the empirical pass."""
    from strawberry_amd import front
    n_loci = int(float(os.environ.get("SB_FRONT_LOCI", "60000")))
    n_frags = float(os.environ.get("SB_FRONT_FRAGS", "2e8"))
    chunk = int(float(os.environ.get("SB_FRONT_CHUNK_MB", "512")) * (1 << 20))
    empirical = os.environ.get("SB_CHAIN_INSERT", "empirical") != "given"
    pinned = os.environ.get("SB_FRONT_PINNED", "1") == "1"
    note = None
    try:
        import psutil
        avail = psutil.virtual_memory().available
        need = 2 * 175.0 * n_frags * 1.15
        if avail < need:
            n_frags = float(int(avail / 1.15 / 350.0))
            note = "host memory: %.0f GB available, sample cut to %.3g read pairs" % (avail / 1e9, n_frags)
    except ImportError:
        pass
    n_parts = 2 if n_frags > 2.4e8 else 1
    size_of = lambda q: {"theta": q.n_iso, "fpkm": q.n_iso, "frac": q.n_iso, "tpm": q.n_iso, "keep": q.n_iso, "status": q.n_loci, "iters": q.n_loci}  # noqa: E731
    parts, want, resident_ms, to_host_s, pinned_all = [], [], 0.0, 0.0, True
    for k in range(n_parts):
        # (a multi-part sample is checked under a GIVEN law: the halves' resident passes are then independent of each other)
        q = front.FrontQuantifier(ctx, n_loci=n_loci, n_frags=n_frags / n_parts, seed=31 + 1000 * k, resident=True,
                                  empirical=empirical and n_parts == 1)
        torch.cuda.empty_cache()
        for _ in range(2):
            q.step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(3):
            q.step()
        torch.cuda.synchronize(dev)
        resident_ms += (time.perf_counter() - t0) / 3 * 1e3
        want.append(({n: getattr(q, n)[:m].copy() for n, m in size_of(q).items()}, q.front_hit_off.copy()))
        t0 = time.perf_counter()
        host = q.to_host(chunk, pinned=pinned, ref_id=k)
        to_host_s += time.perf_counter() - t0
        pinned_all = pinned_all and host["pinned"]
        note = note or host["note"]
        q.unpin()
        ctx.L.sbgpu_release_idle_memory()
        torch.cuda.empty_cache()
        parts.append(q)
    n_bytes, n_records = sum(q.n_bytes for q in parts), sum(q.n_records for q in parts)
    free0 = torch.cuda.mem_get_info(dev)[0]
    # the upload alone: the same chunks, host -> one device buffer, nothing else
    dbuf = torch.empty(chunk, dtype=torch.uint8, device=dev)
    up = []
    for _ in range(2):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for q in parts:
            for (i, j, b0, b1, off) in q.h_chunks:
                dbuf[:b1 - b0].copy_(q.h_bytes[b0:b1], non_blocking=True)
        torch.cuda.synchronize(dev)
        up.append((time.perf_counter() - t0) * 1e3)
    upload_ms = min(up)
    del dbuf
    torch.cuda.empty_cache()
    steps = max(1, min(args.steps, 3))
    if n_parts == 1:
        q = parts[0]
        run = lambda: q.stream_step()                                        # noqa: E731
    else:
        run = lambda: front.FrontQuantifier.stream_parts(parts, empirical=empirical)   # noqa: E731
    res = run()                                 # (warm-up: the pool gets its blocks)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        res = run()
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / steps * 1e3
    if n_parts == 1:
        info = res
        same = bool(np.array_equal(q.front_hit_off, want[0][1]) and all(np.array_equal(getattr(q, n)[:size_of(q)[n]], v) for n, v in want[0][0].items()))
        parity_what = "unique hits, theta, FPKM, Frac, keep, TPM, status, iterations == the resident pass' (sbgpu_bam_decode_device .. sbgpu_quantify_resident on the whole sample), bit for bit"
        law, tpm_sum, n_loci_all, n_pairs_all = q.law, float(q.tpm[:q.n_iso].sum()), q.n_loci, q.n_frags
    else:
        info, law, tpm_sum = res["info"], res["law"], float(res["tpm"].sum())
        given = front.FrontQuantifier.stream_parts(parts, empirical=False)
        same, j0, l0 = True, 0, 0
        for q, (w, _) in zip(parts, want):
            same = same and bool(np.array_equal(given["theta"][j0:j0 + q.n_iso], w["theta"]) and np.array_equal(given["status"][l0:l0 + q.n_loci], w["status"]) and
                                 np.array_equal(given["iters"][l0:l0 + q.n_loci], w["iters"]))
            j0, l0 = j0 + q.n_iso, l0 + q.n_loci
        parity_what = ("a given-law (-i 250/30) pass over the joined stream: theta, status, iterations of each half == that half's own resident pass, bit for bit "
                       "(the timed pass is the empirical one; the joined sample cannot be resident: that is the point)")
        n_loci_all, n_pairs_all = sum(q.n_loci for q in parts), sum(q.n_frags for q in parts)
    bound = max(upload_ms, resident_ms)
    peak = int(free0 - info["least_free_device_bytes"])
    out = {
        "metric": "loci/s and G records/s, BAM alignment records in HOST memory -> abundances (C3-scale)", "value": n_loci_all / (ms * 1e-3),
        "unit": "loci/s", "n_gpus": 1, "steps": steps, "warmup": 1, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8 records, u32 intervals, f64", "data": "synthetic",
        "grecords_per_s": n_records / (ms * 1e-3) / 1e9, "mfrags_per_s": n_pairs_all / (ms * 1e-3) / 1e6,
        "config": {"workload": WORKLOADS["c3-front"] + "; the records start in HOST memory and are pushed in chunks (sbgpu_front_stream_*)",
                   "loci": n_loci_all, "read_pairs": n_pairs_all, "records": n_records, "record_bytes": n_bytes, "chunk_bytes": chunk,
                   "chunks": info["chunks"], "host_memory": "page-locked" if pinned_all else "pageable", "parts": n_parts, "note": note},
        "launch": launch, "insert": "empirical" if empirical else "given (-i 250/30)", "tpm": True,
        "law": {k: v for k, v in law.items() if k != "emp_hist"}, "tpm_sum": tpm_sum,
        "from_host": {
            "ms_per_step": ms, "upload_alone_ms": upload_ms, "upload_GBps": n_bytes / upload_ms / 1e6, "resident_pass_ms": resident_ms,
            "over_the_larger_of_the_two": ms / bound, "end_to_end_GBps": n_bytes / ms / 1e6,
            "peak_device_bytes": peak, "peak_device_GB": peak / 1e9, "device_free_before_GB": free0 / 1e9, "stream": info,
            "parity_ok": same, "parity": parity_what, "records_to_host_s": to_host_s,
            "what": "peak_device_bytes: free device memory before the first pass (the library's pool empty, torch's cache empty) minus the least free "
                    "memory any pass saw (hipMemGetInfo after every chunk and after the last stage): two chunk buffers of 2 x chunk_bytes, a chunk's "
                    "arenas, the store of unique hits, the last stage's scratch, and whatever the library's pool holds idle; resident_pass_ms: "
                    "the device entries on the resident sample%s" % (" (the two halves' passes added)" if n_parts > 1 else "")},
        "roofline": {"bound": "pcie", "kernel": "host -> device copies of the record stream (the pass is upload-bound)", "achieved": n_bytes / ms / 1e6,
                     "peak": n_bytes / upload_ms / 1e6, "unit": "GB/s", "frac": upload_ms / ms, "traffic": None,
                     "note": "peak = the measured rate of the same chunks' uploads alone on this box; the kernels' own rooflines are c3-front's (resident)"},
    }
    print(json.dumps(out))
    if not same:
        raise SystemExit("bench.py: the chunked pass does not reproduce the resident pass")


def self_launch(args):
    """`--gpus N` (N > 1) without a torchrun environment: start the N ranks ourselves, one process per GPU, with the
    driver's own command line (`python -m torch.distributed.run --nproc-per-node N bench.py ...`).  This process has
    not touched the GPU (no HIP call, no torch.cuda query besides the device count) and never does: it waits for
    the ranks, whose rank 0 prints the JSON line on the stdout they inherit, and exits with their return code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def dry_run(args):
    """SB_BENCH_DRY_RUN=1: the launch and the collectives of a run without its GPU work -- what the CPU test of the
    self-launch drives (gloo, no device).  Every rank goes through the same rendezvous, barrier, max-over-ranks and
    sum-over-ranks calls as a real run; rank 0 prints a line that says it measured nothing."""
    import torch
    from strawberry_amd import dist as sdist
    rank, world, _ = sdist.init_process_group("gloo")
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    sdist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    sdist.allreduce_max_(t)
    n = torch.tensor([1, rank], dtype=torch.int64)
    sdist.allreduce_sum_(n)
    tab = sdist.gather_values([float(rank + 1), 10 * rank], rank, world)   # the per-rank table of a real run's line
    sdist.barrier()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_seen": int(n[0]), "rank_sum": int(n[1]),
                          "max_over_ranks": float(t.item()), "steps": args.steps, "warmup": args.warmup,
                          "workload": args.workload, "scaling": args.scaling,
                          "per_rank_ms": [float(x) for x in tab[:, 0]], "slowest_rank": int(tab[:, 0].argmax()),
                          "capped_loci_per_rank": [int(x) for x in tab[:, 1]], "value_strong": None, "value_weak": None,
                          "note": "SB_BENCH_DRY_RUN: launch + collectives only, nothing measured"}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps K (default: 200 for the EM workloads c3 / c2 / c5 -- a step is under a millisecond, and a run of batches needs ~20 steps to reach its steady rate --, 20 for the others)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps W before them (default: 20 for c3 / c2 / c5, 3 for the others)")
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="strong (default): ONE batch / ONE chain sample, its loci sharded over the ranks -- BASELINE config 3, "
                         "'loci sharded 1 -> 2 -> 4 -> 8' -- is the headline (`value`, `ms_per_step`); weak: every rank its own "
                         "full batch.  Both are measured either way and reported under `strong_scaling` / `weak_scaling`; "
                         "at one rank they are the same run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-chain", action="store_true", help="default workload: leave the fragments -> abundances leg out of the line")
    ap.add_argument("--from-host", action="store_true", help="c3-front: the records start in host memory and are pushed in chunks (sbgpu_front_stream_*)")
    ap.add_argument("--no-front", action="store_true", help="default workload: leave the records -> theta leg out of the line (also SB_BENCH_NO_FRONT=1)")
    args = ap.parse_args()
    em_workload = args.workload in ("c3", "c2", "c5")
    if args.steps is None:
        args.steps = 200 if em_workload else 20
    if args.warmup is None:
        args.warmup = 20 if em_workload else 3
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))       # before anything here has touched the GPU
    if os.environ.get("SB_BENCH_DRY_RUN") == "1":
        return dry_run(args)

    import torch
    from strawberry_amd import dist as sdist
    from strawberry_amd import em

    # One rank per GPU over RCCL.  A box with fewer devices than ranks (the 1-GPU test box with --gpus 2) still runs
    # all ranks, several per device: RCCL refuses two ranks on one device, so the collective is then gloo on a host
    # copy, and the line says so (`oversubscribed`).
    n_dev = torch.cuda.device_count()             # does not initialise the GPU
    if n_dev < 1:
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    want_world = int(os.environ.get("WORLD_SIZE", "1"))
    oversub = want_world > n_dev
    rank, world, local_rank = sdist.init_process_group("gloo" if oversub else None)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ctx = em.Context(local_rank)
    # the per-step collective: the C ABI's own RCCL binding (sbgpu_allreduce_sum_*) whenever there is more than one
    # rank; SB_COMM=torch selects torch.distributed's all-reduce instead.  All ranks take the same one.
    comm, comm_note = None, None
    if world > 1 and not oversub and os.environ.get("SB_COMM", "abi") == "abi":
        try:
            comm = sdist.AbiComm(ctx)
            ok = 1
        except Exception as e:   # noqa: BLE001 -- reported in the line, and the ranks agree on what to use
            comm_note, ok = "C-ABI communicator failed (%s); torch.distributed used" % e, 0
        flag = torch.tensor([ok], dtype=torch.int64, device=dev)
        sdist.allreduce_sum_(flag)
        if int(flag.item()) != world:
            comm = None
            comm_note = comm_note or "C-ABI communicator failed on another rank; torch.distributed used"
    collective = ("C ABI (sbgpu_allreduce_sum_f64 over RCCL)" if comm is not None else
                  "gloo on a host copy (%d ranks share %d device(s): RCCL refuses two ranks per device)" % (world, n_dev)
                  if oversub else "torch.distributed (RCCL)" if world > 1 else "none (one rank)")
    launch = {"world_size": world, "devices": n_dev, "collective": collective,
              "rccl_ranks": comm.rccl_ranks() if comm is not None else None}     # ncclCommCount of the C-ABI communicator
    if oversub:
        launch["oversubscribed"] = True
    if comm_note:
        launch["collective_note"] = comm_note

    if args.workload == "c3-chain":
        return chain_main(args, ctx, dev, rank, world, sdist, torch, launch, comm=comm)
    if args.workload == "c3-front" and args.from_host:
        if world != 1:
            raise SystemExit("bench.py: --from-host is a one-GPU line")
        return front_host_main(args, ctx, dev, torch, launch)
    if args.workload == "c3-front":
        return front_main(args, ctx, dev, rank, world, sdist, torch, launch, comm=comm)

    def make_quant(b, f32=False, solver=None):
        solver = solver or em.EmBatchSolver(b, ctx)
        # pass-1 normaliser (alignments.cpp:1372): global mapped fragments, one all-reduce at set-up
        tot = torch.tensor([b.n_frags], dtype=torch.int64, device=dev)
        (comm.allreduce_sum_(tot) if comm is not None else sdist.allreduce_sum_(tot))
        total_mapped = int(min(int(tot.item()), 2**31 - 1))   # the reference holds it in an int
        return solver, sdist.ShardQuantifier(solver, total_mapped, min_isoform_frac=0.0, comm=comm, f32=f32,
                                             pipelined=os.environ.get("SB_PIPELINE", "1" if args.steps >= 10 else "0") != "0")  # quant-only (-r): keep all
        # (a run of batches: from ten steps on -- the first steps of a burst run side by side in pairs before the kinds have
        # drifted apart, 5 steps take 0.88 ms each as a run and 0.83 one after the other, 20 take 0.74, 200 take 0.70)

    # ---- weak scaling: every rank holds its OWN full-size batch
    batch = make_batch(args.workload, rank)
    solver, quant = make_quant(batch)
    c5_bias = None
    if args.workload == "c5":
        # the bias factors: per bin from its sequence's GC ratio (bin-sequence kernel, on the device), per isoform a model
        # constant; applied by the EM kernels at tile load (sbgpu_em_run_device_bias) for both precisions
        from strawberry_amd import bias as sbias
        rb, ib, c5_bias = sbias.make_c5_bias(ctx, batch, seed=0xB1A5 + rank)
        solver.set_bias(rb, ib)
    wall, gpu_ms = timed_steps(quant, args.steps, args.warmup, dev, sdist, torch)
    counts = torch.tensor([batch.n_loci, batch.n_frags], dtype=torch.int64, device=dev)
    sdist.allreduce_sum_(counts)
    n_loci_all, n_frags_all = int(counts[0].item()), int(counts[1].item())
    weak = {"value": n_loci_all * args.steps / wall, "ms_per_step": wall / args.steps * 1e3,
            "mfrags_per_s": n_frags_all * args.steps / wall / 1e6, "loci": n_loci_all}
    # the same K steps strictly one after the other (no step starts before the one before it has written its TPM): the latency
    # of ONE batch beside the rate of a run of batches above -- reported, never `value`
    serial_ms = None
    if quant.pipelined:
        quant.pipelined = False
        serial_wall, _ = timed_steps(quant, args.steps, min(args.warmup, 2), dev, sdist, torch)
        serial_ms = serial_wall / args.steps * 1e3
        quant.pipelined = True

    # ---- strong scaling (BASELINE config 3: "loci sharded 1 -> 2 -> 4 -> 8"): ONE batch, LPT shards, one all-reduce
    if world > 1:
        whole = make_batch(args.workload, 0)
        shard = whole.select(sdist.shard_loci(whole.nrow, whole.niso, world)[rank])
        ssolver, squant = make_quant(shard)
        swall, _ = timed_steps(squant, args.steps, args.warmup, dev, sdist, torch)
        strong = {"value": whole.n_loci * args.steps / swall, "ms_per_step": swall / args.steps * 1e3,
                  "mfrags_per_s": whole.n_frags * args.steps / swall / 1e6, "loci": whole.n_loci,
                  "loci_this_rank": shard.n_loci}
        # Who sets the step.  Every step ends in the all-reduce, so the ranks' wall times are coupled; what tells them
        # apart is each rank's OWN kernel time: the shard's EM kernels alone (HIP events on their streams, a few untimed
        # steps without the collective).  Gathered over one all-reduce of a zero-padded table.
        own_ms = timed_steps.own_wall / args.steps * 1e3
        sres = ssolver.results()
        ssolver.set_timing(True)
        em_probe = []
        for _ in range(3):
            ssolver.run_em()
            ssolver.synchronize()
            em_probe.append(max(ssolver.last_kernel_ms()))
        ssolver.set_timing(False)
        tab = sdist.gather_values([own_ms, float(np.min(em_probe)), shard.n_loci, int((sres["status"] == 3).sum()),
                                   float((shard.nrow * shard.niso).sum())], rank, world, device=dev)
        strong.update({"per_rank_ms": [float(x) for x in tab[:, 0]], "per_rank_em_kernel_ms": [float(x) for x in tab[:, 1]],
                       "slowest_rank": int(np.argmax(tab[:, 1])), "loci_per_rank": [int(x) for x in tab[:, 2]],
                       "capped_loci_per_rank": [int(x) for x in tab[:, 3]], "elements_per_rank": [int(x) for x in tab[:, 4]],
                       "per_rank_note": "per_rank_ms: a rank's K steps up to its own synchronize (the per-step all-reduce couples "
                                        "the ranks); per_rank_em_kernel_ms: its EM kernels alone, longest kind, HIP events -- a "
                                        "shard cannot end before one of its 1000-iteration loci does (capped_loci_per_rank)"})
        del squant, ssolver
    else:
        strong = dict(weak, loci_this_rank=batch.n_loci)   # one rank: the same run
    strong["sharding"] = "one %d-locus batch, LPT shards by elements x predicted iterations (dist.shard_loci), " \
                         "no locus data crosses ranks, 1 all-reduce (8 B) per step" % strong["loci"]

    # per-kind EM kernel time: HIP events on the streams the kernels run on, averaged over a few
    # extra (untimed) steps -- reading them synchronises, so it stays out of the timed region
    # (pipelined steps: the events of the LAST step of a short burst -- a kernel that ran, as in the timed region, beside the
    # tail of the step before it; a step by itself would time the kernels alone on the chip)
    probe, phase_probe, alone = [], [], []
    solver.set_timing(True)
    for _ in range(5):
        for _ in range(4 if getattr(quant, "pipelined", False) else 1):
            quant.step()
        quant.finish()
        probe.append(solver.last_kernel_ms())
        phase_probe.append(solver.last_phase_ms())
    for _ in range(5 if getattr(quant, "pipelined", False) else 0):   # a step by itself: which kernel is the longest one
        quant.step()
        quant.finish()
        alone.append(solver.last_kernel_ms())
    solver.set_timing(False)
    kern_ms = np.mean(np.array(probe), axis=0)
    kern_ms_alone = np.mean(np.array(alone), axis=0) if alone else kern_ms
    phase_ms = [float(x) for x in np.mean(np.array(phase_probe), axis=0)] if phase_probe and phase_probe[0] else []

    # ---- C5: the fp32 variant next to the fp64 path just timed (same batch, same plan)
    c5 = None
    if args.workload == "c5":
        quant.step()
        res64 = solver.results()
        _, q32 = make_quant(batch, f32=True, solver=solver)
        wall32, _ = timed_steps(q32, args.steps, args.warmup, dev, sdist, torch)
        res32 = solver.results()
        ok = np.isin(res64["status"], (0, 3)) & np.isin(res32["status"], (0, 3))
        m = ok[np.repeat(np.arange(batch.n_loci), batch.niso)]
        tpm_rel = np.abs(res32["tpm"] - res64["tpm"])[m] / np.maximum(res64["tpm"][m], 1e-3)
        c5 = {"bias": c5_bias,
              "f32": {"value": n_loci_all * args.steps / wall32, "ms_per_step": wall32 / args.steps * 1e3},
              "f64": {"value": weak["value"], "ms_per_step": weak["ms_per_step"]},
              "tolerance": {"isoforms_within_1e-4_relative_tpm": float((tpm_rel < 1e-4).mean()),
                            "tpm_rel_err_p99": float(np.percentile(tpm_rel, 99)), "tpm_rel_err_max": float(tpm_rel.max()),
                            "loci_status_changed": int((res64["status"] != res32["status"]).sum()),
                            "loci_iteration_count_changed": int((res64["iters"] != res32["iters"]).sum()),
                            "histogram": "profiles/r05_c5_sweep.json (tools/c5_sweep.py)"}}
        # the fp32 kernels' own times (HIP events per kind), for their roofline with s = 4
        probe32 = []
        solver.set_timing(True)
        for _ in range(5):
            q32.step()
            probe32.append(solver.last_kernel_ms())
        solver.set_timing(False)
        c5["f32"]["kernel_ms_by_kind"] = [float(x) for x in np.mean(np.array(probe32), axis=0)]
        quant.step()   # leave the fp64 result in place for the roofline / parity legs below

    # ---- the chain leg of the default line: fragments -> abundances on 2e8 read pairs in HBM (every rank takes part:
    # its collectives are collective); --no-chain leaves it out
    chain_obj = None
    if args.workload == "c3" and not args.no_chain:
        res_keep = solver.results() if rank == 0 else None       # (the chain reuses the context's scratch; results first)
        chain_obj = chain_leg(args, ctx, dev, rank, world, sdist, torch, strong=args.scaling == "strong", comm=comm)
    front_obj = None
    if args.workload == "c3" and not args.no_front and not os.environ.get("SB_BENCH_NO_FRONT"):
        if chain_obj is None:
            res_keep = solver.results() if rank == 0 else None
        front_obj = front_leg(args, ctx, dev, rank, world, sdist, torch, comm=comm)
    if rank != 0:
        return
    res = res_keep if (chain_obj is not None or front_obj is not None) else solver.results()
    head = strong if args.scaling == "strong" else weak
    ms_per_step = head["ms_per_step"]

    # ---- roofline of the dominant EM kernel (this rank's batch)
    kinds = solver.plan.locus_kinds()
    kind_names = ["em_fused_kernel<0,1> (wave form, half tile)", "em_fused_kernel<0,2> (wave form, base tile)",
                  "em_fused_kernel<0,4> (wave form, double tile)", "em_fused_kernel<4,2> (256-lane block form)",
                  "em_fused_kernel<4,12> (256-lane block form, tall tile)", "em_wide_kernel (several workgroups per locus) + em_stream_kernel"]
    nrow, niso = batch.nrow, batch.niso
    b_locus = nrow * niso * 8 + nrow * 4 + niso * 8 + 24
    # the dominant kernel: the longest one of a step by itself where one stands out (C3-T: the wide-locus rounds); where the kinds
    # run side by side for about the same time (C3: 0.75 / 0.69 / 0.5-0.7 ms, in an order that changes from run to run) the one
    # that moves most of the batch's bytes
    order = np.argsort(-kern_ms_alone)
    if kern_ms_alone[order[0]] > 2.0 * kern_ms_alone[order[1]]:
        dom = int(order[0])
    else:
        dom = int(np.argmax([float(b_locus[kinds == k].sum()) if kern_ms_alone[k] > 0 else -1.0 for k in range(6)]))
    sel = kinds == dom
    fl_locus = res["iters"].astype(np.int64) * (5 * nrow * niso + nrow + 3 * niso)
    dom_s = kern_ms[dom] * 1e-3
    ach_gbs = float(b_locus[sel].sum()) / dom_s / 1e9
    traffic, traffic_note = pmc_traffic(args.workload, dom)
    weak_ms = weak["ms_per_step"]
    roofline = {
        "bound": "hbm", "kernel": kind_names[dom], "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
        "kernel_ms": float(kern_ms[dom]), "kernel_ms_of_a_step_by_itself": float(kern_ms_alone[dom]), "kernel_loci": int(sel.sum()),
        "algorithmic_bytes": int(b_locus[sel].sum()),
        "note": "F stays in registers for all iterations: the loop is FP64-VALU/latency bound, see fp64_valu",
        "fp64_valu": {"achieved": float(fl_locus[sel].sum()) / dom_s / 1e12, "peak": FP64_VALU_PEAK_TF,
                      "unit": "TFLOP/s", "frac": float(fl_locus[sel].sum()) / dom_s / 1e12 / FP64_VALU_PEAK_TF,
                      "algorithmic_flops": int(fl_locus[sel].sum())},
        "all_kernels_ms": {kind_names[k]: float(kern_ms[k]) for k in range(6) if kern_ms[k] > 0},
        "whole_batch": {"achieved": float(b_locus.sum()) / (weak_ms * 1e-3) / 1e9, "unit": "GB/s",
                        "frac": float(b_locus.sum()) / (weak_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
    }

    out = {
        "metric": "loci/s, EM-to-convergence (+ FPKM/TPM epilogue), %s" % args.workload.upper(),
        "value": head["value"], "unit": "loci/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "mfrags_per_s": head["mfrags_per_s"],
        "config": {"workload": WORKLOADS[args.workload], "loci_per_gpu": batch.n_loci if args.scaling == "weak" else strong["loci_this_rank"],
                   "fragments_per_gpu": batch.n_frags,
                   "sharding": "independent loci per rank (own batch each), 1 all-reduce (8 B) per step" if args.scaling == "weak" else strong["sharding"],
                   "collective": collective, "size_classes": solver.plan.info()["n_classes"],
                   "steps_pipelined": bool(getattr(quant, "pipelined", False)),
                   "ms_per_step_one_after_the_other": serial_ms,
                   "steps_pipelined_note": "a step's epilogue (abundance_kernel, the all-reduce, tpm_kernel) runs on a stream of its own beside the NEXT step's "
                                           "EM kernels (sbgpu_em_run_device_split joins the kernels into that stream), theta / status / iterations double-"
                                           "buffered; every step produces all of its outputs, all K are complete when the clock stops.  "
                                           "ms_per_step_one_after_the_other: the same K steps with no overlap between two steps (SB_PIPELINE=0 makes that the "
                                           "timed region)"},
        "launch": launch,
        "em_status": {"ok": int((res["status"] == 0).sum()), "init_empty": int((res["status"] == 1).sum()),
                      "denom_zero": int((res["status"] == 2).sum()), "maxiter": int((res["status"] == 3).sum()),
                      "mean_iters": float(res["iters"].mean())},
        "gpu_event_ms_per_step": gpu_ms / args.steps,
        "wave_phase_ms": phase_ms,
        "weak_scaling": weak,
        "strong_scaling": strong,
        # both values beside the headline, for a reader of the N = 1, 2, 4, 8 series: the strong one divides ONE batch
        # (bounded below by a 1000-iteration locus per rank), the weak one gives every rank a batch of its own
        "value_strong": strong["value"], "value_weak": weak["value"],
        "per_rank_ms": strong.get("per_rank_ms"), "slowest_rank": strong.get("slowest_rank"),
        "capped_loci_per_rank": strong.get("capped_loci_per_rank"),
        "roofline": roofline,
    }
    if chain_obj is not None:
        out["chain"] = chain_obj
    if front_obj is not None:
        out["front"] = front_obj
    if c5 is not None:
        # the headline of this workload is the fp32 variant; the fp64 numbers of the same run sit beside it
        out.update({"value": c5["f32"]["value"], "ms_per_step": c5["f32"]["ms_per_step"], "dtype": "f32",
                    "mfrags_per_s": n_frags_all / (c5["f32"]["ms_per_step"] * 1e-3) / 1e6, "c5": c5})
        # the headline's own roofline: the fp32 instantiation of the dominant kind, algorithmic bytes with s = 4
        # (weights and theta in fp32, counts int32, + the two factor arrays the tile load reads)
        k32 = np.array(c5["f32"]["kernel_ms_by_kind"])
        dom32 = int(np.argmax(k32))
        sel32 = kinds == dom32
        b32 = nrow * niso * 4 + nrow * 4 + niso * 4 + 24 + nrow * 4 + niso * 4
        ach32 = float(b32[sel32].sum()) / (k32[dom32] * 1e-3) / 1e9
        out["roofline_f64"] = out["roofline"]
        out["roofline"] = {"bound": "hbm", "kernel": kind_names[dom32].replace(">", ", float>", 1) + " with the bias factors applied at tile load",
                           "achieved": ach32, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach32 / HBM_PEAK_GBS, "traffic": None,
                           "kernel_ms": float(k32[dom32]), "kernel_loci": int(sel32.sum()), "algorithmic_bytes": int(b32[sel32].sum()),
                           "note": "fp32 kernel, HIP-event time on its own stream; the loop runs on-chip (see roofline_f64.fp64_valu for the fp64 twin)"}
    if world == 1 and not args.no_cpu_baseline:
        cpu_batch = batch
        if c5 is not None:   # the CPU solves the same biased problem: the factors multiplied into a host copy of F
            from strawberry_amd import bias as sbias
            from strawberry_amd.synth import LocusBatch
            cpu_batch = LocusBatch(batch.row_off, batch.iso_off, batch.f_off, batch.count,
                                   sbias.biased_weights(batch, solver.d_row_bias.cpu().numpy(), solver.d_iso_bias.cpu().numpy()),
                                   batch.length, "C5 biased")
        out["cpu_baseline"], out["parity"] = cpu_baseline(cpu_batch, res)
        if c5 is not None:
            # not a parity path: exp2 on the device against numpy's may move a weight's last bit, and with it, rarely, a count
            out["parity"]["note"] = "fp64 run with the factors applied on the device vs the reference EmSolver on host-multiplied weights; informational"
            out["parity"]["ok"] = bool(out["parity"]["status_mismatches"] <= 5 and out["parity"]["theta_max_rel_err"] < 1e-6)
        if not out["parity"]["ok"]:
            print(json.dumps(out))
            raise SystemExit("bench.py: the GPU result of the timed batch does not match the CPU %s: %r" % (
                out["parity"]["against"], out["parity"]))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
