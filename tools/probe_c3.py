#!/usr/bin/env python3
"""GPU box: time sub-batches of C3 (long-running loci only, per class) to find what is slow."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from strawberry_amd import em, synth

def run(b, ctx, label):
    s = em.EmBatchSolver(b, ctx)
    s.set_timing(True)
    s.run_em(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s.run_em(); torch.cuda.synchronize()
        best = min(best, max(s.last_kernel_ms()))
    r = s.results()
    print("%-40s %6d loci  %8.3f ms  iters mean %7.1f max %4d  classes %d" % (label, b.n_loci, best, r["iters"].mean(), r["iters"].max(), s.plan.info()["n_classes"]), flush=True)
    return s, r

ctx = em.default_context(0)
b = synth.make_c3()
s, r = run(b, ctx, "C3 full")
it = r["iters"]; kinds = s.plan.locus_kinds()
wave = kinds < 3
run(b.select(np.nonzero(wave)[0]), ctx, "C3 wave-kind loci only")
run(b.select(np.nonzero(wave & (it == 1000))[0]), ctx, "wave-kind MAXITER loci")
run(b.select(np.nonzero(wave & (it > 256))[0]), ctx, "wave-kind it>256")
run(b.select(np.nonzero(wave & (it <= 64))[0]), ctx, "wave-kind it<=64")
nrow, niso = b.nrow, b.niso
for lo, hi in ((1, 2), (3, 4), (5, 8), (9, 16), (17, 64)):
    m = wave & (it == 1000) & (niso >= lo) & (niso <= hi)
    if m.sum():
        run(b.select(np.nonzero(m)[0]), ctx, "MAXITER niso %d-%d" % (lo, hi))
m = wave & (it == 1000)
for l in np.nonzero(m)[0][:6]:
    run(b.select(np.array([l])), ctx, "single MAXITER locus %dx%d" % (nrow[l], niso[l]))
