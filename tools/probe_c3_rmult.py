#!/usr/bin/env python3
"""GPU box: how fast do C3's long-running wave-kind loci go with smaller tiles (more lanes per locus)?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from strawberry_amd import em, synth
ctx = em.default_context(0)
b = synth.make_c3()
s = em.EmBatchSolver(b, ctx); s.run_em(); r = s.results()
s.set_timing(True)
it = r["iters"]; kinds = s.plan.locus_kinds()
sel = np.nonzero((kinds < 3) & (it == 1000))[0]
sub = b.select(sel)
s2 = em.EmBatchSolver(sub, ctx)
s2.set_timing(True)
for _ in range(2):
    s2.run_em(); torch.cuda.synchronize()
best = min((s2.run_em(), torch.cuda.synchronize(), max(s2.last_kernel_ms()))[2] for _ in range(3))
print("SBGPU_WAVE_RMULT=%s  %d MAXITER loci  %.3f ms  kinds %s" % (os.environ.get("SBGPU_WAVE_RMULT", "auto"), len(sel), best,
      np.bincount(s2.plan.locus_kinds(), minlength=6).tolist()))
one = b.select(sel[:1])
s3 = em.EmBatchSolver(one, ctx)
s3.set_timing(True)
for _ in range(2):
    s3.run_em(); torch.cuda.synchronize()
best = min((s3.run_em(), torch.cuda.synchronize(), max(s3.last_kernel_ms()))[2] for _ in range(3))
print("   single locus %dx%d: %.3f ms" % (one.nrow[0], one.niso[0], best))
