#!/usr/bin/env python3
"""The drop-in under the reference's OWN driver, timed.

A synthetic sample of >= 10 000 loci (strawberry_amd/chain.py::DeviceSample, the chain workload's law at a smaller size) is
written out as GTF + coordinate-sorted BAM, and three programs run on it with the same command line
(`<bam> -g <gtf> -r -i 250/30`, one thread):

  strawberry_ref            the reference program, compiled from /root/reference (oracle/Makefile)
  strawberry_sbgpu          the reference's objects, EmSolver::init/run served by libsbgpu.so ONE LOCUS AT A TIME
                            (oracle/sbgpu_em_shim.cpp: a plan, an upload, a launch and a synchronisation per locus)
  strawberry_sbgpu_batched  the reference's objects, Sample::procSample restructured into collect -> ONE sbgpu_em_batch ->
                            epilogue (oracle/sbgpu_batched_shim.cpp)

All three must write the same out.gtf and -f table; the table of wall times goes to stdout (-> profiles/r04_dropin.txt).
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
             ("strawberry_sbgpu_batched", "reference driver, procSample batched: ONE sbgpu_em_batch")]
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
    same = all(outs[n] == outs["strawberry_ref"] for n in outs) if "strawberry_ref" in outs else None
    print("# out.gtf and the -f table of the %d programs are %s" % (len(outs), "IDENTICAL, byte for byte" if same else "DIFFERENT" if same is False else "not compared"))
    if same is False:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
