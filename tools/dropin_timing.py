#!/usr/bin/env python3
"""The drop-in under the reference's OWN driver, timed.

A synthetic sample of >= 10 000 loci (strawberry_amd/chain.py::DeviceSample, the chain workload's law at a smaller size) is
written out as GTF + coordinate-sorted BAM, and five programs run on it with the same command line
(`<bam> -g <gtf> -r -i 250/30`, one thread):

  strawberry_ref            the reference program, compiled from /root/reference (oracle/Makefile)
  strawberry_sbgpu          the reference's objects, EmSolver::init/run served by libsbgpu.so ONE LOCUS AT A TIME
                            (oracle/sbgpu_em_shim.cpp: a plan, an upload, a launch and a synchronisation per locus)
  strawberry_sbgpu_batched  the reference's objects, Sample::procSample restructured into collect -> ONE sbgpu_em_batch ->
                            epilogue (oracle/sbgpu_batched_shim.cpp)
  strawberry_sbgpu_chain    the same one level up: the loci's transcripts and unique hits are collected, ONE
                            sbgpu_quantify_host call does bins + weights + EM on the device (oracle/sbgpu_chain_shim.cpp)

  strawberry_sbgpu_front    the deepest: the reference's three passes over the BAM file (read lengths, preProcess, procSample)
                            replaced -- one inflate, then sbgpu_bam_decode_device -> sbgpu_assign_reads_device ->
                            sbgpu_pair_mates_device -> sbgpu_collapse_pairs_device -> sbgpu_quantify_host (oracle/sbgpu_front_shim.cpp)

All must write the same out.gtf and -f table; the table of wall times goes to stdout (-> profiles/r04_dropin.txt).
Test infrastructure: runs on the GPU box (the programs travel there as oracle/_ref/ binaries)."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from oracle.lib import SAM2BAM, write_gtf_from_annotation, write_sam_from_hits
    from strawberry_amd import chain
    n_loci = int(float(os.environ.get("SB_DROPIN_LOCI", "12000")))
    n_frags = float(os.environ.get("SB_DROPIN_FRAGS", "3.6e6"))
    dev = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
    s = chain.DeviceSample(torch, dev, n_loci=n_loci, n_frags=n_frags, seed=77)
    hits = s.host_hits(n_loci)
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    progs = [("strawberry_ref", "reference program (CPU)"),
             ("strawberry_sbgpu", "reference driver, EmSolver on the device one locus per call"),
             ("strawberry_sbgpu_batched", "reference driver, procSample batched: ONE sbgpu_em_batch"),
             ("strawberry_sbgpu_chain", "reference driver, procSample batched one level up: ONE sbgpu_quantify_host (bins + weights + EM on the device)"),
             ("strawberry_sbgpu_front", "reference driver, none of its BAM handling: inflate once, then decode -> stream -> pairs -> unique hits -> chain on the device")]
    rows, outs = [], {}
    with tempfile.TemporaryDirectory() as tmp:
        t = time.perf_counter()
        write_gtf_from_annotation(os.path.join(tmp, "s.gtf"), s.annot, n_loci)
        n_rec = write_sam_from_hits(os.path.join(tmp, "s.sam"), hits)
        subprocess.check_call([SAM2BAM, os.path.join(tmp, "s.sam"), os.path.join(tmp, "s.bam")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        os.remove(os.path.join(tmp, "s.sam"))
        print("# sample: %d loci, %d isoforms, %d read pairs (%d unique hits), %d BAM records (%.0f MB), written in %.1f s" % (
            n_loci, int(s.annot.iso_off[-1]), int(hits.mass.sum()), hits.n_hits, n_rec, os.path.getsize(os.path.join(tmp, "s.bam")) / 1e6,
            time.perf_counter() - t))
        for name, what in progs:
            exe = os.path.join(ref_dir, name)
            if not os.path.exists(exe):
                print("# %s not built" % name)
                continue
            best = None
            for rep in range(2):        # the second run has the BAM in the page cache and the GPU initialised once before
                for f in ("out.gtf", "ctx.tsv", "log.txt"):
                    if os.path.exists(os.path.join(tmp, f)):
                        os.remove(os.path.join(tmp, f))
                cmd = [exe, "s.bam", "-g", "s.gtf", "-r", "-i", "250/30", "-o", "out.gtf", "-T", "log.txt", "-f", "ctx.tsv"]
                t = time.perf_counter()
                r = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True, env=dict(os.environ, SBGPU_DROPIN_TIMING="1"))
                dt = time.perf_counter() - t
                if r.returncode != 0:
                    print("# %s failed: %s" % (name, r.stderr[-500:]))
                    break
                note = [l for l in r.stderr.splitlines() if l.startswith("sbgpu")]
                if best is None or dt < best[0]:
                    best = (dt, note)
            if best is None:
                continue
            outs[name] = (open(os.path.join(tmp, "out.gtf")).read().split("\n", 1)[1], open(os.path.join(tmp, "ctx.tsv")).read())
            rows.append((name, what, best[0], best[1]))
    n_pairs = int(hits.mass.sum())
    print("%-26s %10s %12s %14s   %s" % ("program", "wall s", "loci/s", "read pairs/s", "what"))
    for name, what, dt, note in rows:
        print("%-26s %10.2f %12.0f %14.0f   %s" % (name, dt, n_loci / dt, n_pairs / dt, what))
        for l in note:
            print("    " + l)
    bad = False
    if "strawberry_ref" in outs:
        for name in outs:
            if name == "strawberry_ref":
                continue
            verdicts = [compare_text(outs[name][k], outs["strawberry_ref"][k]) for k in (0, 1)]
            print("# %-26s out.gtf: %s; -f table: %s" % (name, verdicts[0][1], verdicts[1][1]))
            bad |= not (verdicts[0][0] and verdicts[1][0])
    if bad:
        raise SystemExit(1)


NUM = None


def printed_ulp(text):
    """one unit in the last printed digit of a decimal number's text"""
    mant, _, exp = text.lower().partition("e")
    dec = len(mant.partition(".")[2])
    return 10.0 ** (-(dec) + (int(exp) if exp else 0))


def compare_text(got, want):
    """Two output files of the same run: identical bytes, or -- field by field -- the same text with numbers that differ
    by ONE unit of their last printed digit at most (the -f table prints weights to 12 significant digits, the GTF 11
    characters of %f: a weight that differs in its 16th digit moves a printed digit once in a thousand numbers).
    -> (acceptable?, description)"""
    import re
    global NUM
    if got == want:
        return True, "IDENTICAL, byte for byte"
    NUM = NUM or re.compile(r"(?<![A-Za-z_.\d])[-+]?\d+\.\d+(?:[eE][-+]?\d+)?|(?<![A-Za-z_.\d])[-+]?\d+[eE][-+]?\d+")
    gl, wl = got.split("\n"), want.split("\n")
    if len(gl) != len(wl):
        return False, "DIFFERENT: %d lines against %d" % (len(gl), len(wl))
    n_num = n_diff = 0
    worst = 0.0
    for a, b in zip(gl, wl):
        if a == b:
            n_num += len(NUM.findall(a))
            continue
        if NUM.sub("#", a) != NUM.sub("#", b):
            return False, "DIFFERENT text: %r against %r" % (a[:200], b[:200])
        for x, y in zip(NUM.findall(a), NUM.findall(b)):
            n_num += 1
            if x != y:
                n_diff += 1
                worst = max(worst, abs(float(x) - float(y)) / max(printed_ulp(x), printed_ulp(y)))
    ok = worst <= 1.01
    return ok, "same text, %d of %d printed numbers differ, by %.0f unit%s of their last printed digit at most%s" % (
        n_diff, n_num, worst, "" if worst <= 1.01 else "s", "" if ok else " -- TOO FAR")


if __name__ == "__main__":
    main()
