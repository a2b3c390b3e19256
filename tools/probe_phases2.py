#!/usr/bin/env python3
"""GPU box: where the phased wave pipeline spends its time on C3 (diagnostic).
Sub-batches of C3 (by kernel kind, by locus size, by iteration count) under different phase settings."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from strawberry_amd import em, synth

ctx = em.default_context(0)


def run(b, label, phases=None, lam=None, reps=3):
    for k, v in (("SBGPU_PHASES", phases), ("SBGPU_PHASE_LAMBDA", lam)):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    s = em.EmBatchSolver(b, ctx)
    s.set_timing(True)
    s.run_em(); torch.cuda.synchronize()
    best, ph = 1e9, []
    for _ in range(reps):
        s.run_em(); torch.cuda.synchronize()
        ms = s.last_kernel_ms()
        if max(ms) < best:
            best, ph, kms = max(ms), s.last_phase_ms(), ms
    r = s.results()
    print("%-52s %6d loci phases %-12s lam %-12s %7.3f ms  kinds %s  phase_ms %s  iters mean %.0f max %d" % (
        label, b.n_loci, phases, lam, best, " ".join("%.3f" % x for x in kms if x > 0), " ".join("%.3f" % x for x in ph),
        r["iters"].mean(), r["iters"].max()), flush=True)
    return s, r


b = synth.make_c3()
os.environ["SBGPU_PHASES"] = "0"
s, r = run(b, "C3 full", "0")
it, kinds = r["iters"], s.plan.locus_kinds()
el = b.nrow * b.niso
wave = kinds <= 2
bw = b.select(np.nonzero(wave)[0])
itw, elw = it[wave], el[wave]
run(bw, "wave-kind loci only", "0")
run(bw, "wave-kind loci only", "32,128,512", "8,2,0.25")
run(bw, "wave-kind loci only", "32,128,512", "2,0.5,0.1")
for lo, hi in ((0, 64), (64, 256), (256, 512), (512, 1024), (1024, 4096)):
    m = (elw >= lo) & (elw < hi)
    sub = bw.select(np.nonzero(m)[0])
    run(sub, "wave loci with %d <= elements < %d" % (lo, hi), "0")
    run(sub, "wave loci with %d <= elements < %d" % (lo, hi), "32,128,512", "8,2,0.25")
    mc = m & (itw >= 1000)
    if mc.any():
        subc = bw.select(np.nonzero(mc)[0])
        run(subc, "  its capped loci only", "0")
        for lam in ("0", "0.5", "4"):
            run(subc, "  its capped loci only, lane-rich from iteration 2", "2", lam)
