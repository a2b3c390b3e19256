#!/usr/bin/env python3
"""GPU box: makespan of each kernel kind of C3 run alone, and of its long-running loci."""
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
    print("%-44s %6d loci  %8.3f ms  iters mean %7.1f max %4d  nrow max %5d niso max %3d" % (
        label, b.n_loci, best, r["iters"].mean(), r["iters"].max(), b.nrow.max(), b.niso.max()), flush=True)
    return s, r

ctx = em.default_context(0)
b = synth.make_c3()
s, r = run(b, ctx, "C3 full")
it = r["iters"]; kinds = s.plan.locus_kinds()
for k, name in ((3, "block"), (4, "tall block")):
    m = kinds == k
    if not m.any():
        continue
    run(b.select(np.nonzero(m)[0]), ctx, name + " loci only")
    mm = m & (it == 1000)
    print("   %d MAXITER loci; iters>256: %d" % (mm.sum(), (m & (it > 256)).sum()))
    if mm.any():
        run(b.select(np.nonzero(mm)[0]), ctx, name + " MAXITER only")
        for l in np.nonzero(mm)[0][:4]:
            run(b.select(np.array([l])), ctx, "  single %s MAXITER locus %dx%d" % (name, b.nrow[l], b.niso[l]))
m = (kinds == 3) | (kinds == 4)
run(b.select(np.nonzero(m)[0]), ctx, "both block kinds together")
