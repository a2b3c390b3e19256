#!/usr/bin/env python3
"""GPU box: time per iteration of the wide-locus kernel by shape (one locus at a time; all run the 1000-iteration cap)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from strawberry_amd import em, synth
from strawberry_amd.synth import _generate
ctx = em.default_context(0)
rng = np.random.Generator(np.random.PCG64(5))
for nrow, niso in ((128, 194), (256, 194), (512, 194), (1160, 194), (2320, 194), (256, 100), (1024, 100), (64, 400), (512, 400), (2048, 400)):
    b = _generate(rng, np.array([nrow], np.int64), np.array([niso], np.int64), np.array([nrow * 50], np.int64))
    s = em.EmBatchSolver(b, ctx)
    s.run_em(); s.synchronize()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); s.run_em(); s.synchronize(); best = min(best, time.perf_counter() - t)
    r = s.results()
    print("%5d x %3d: %8.3f ms for %4d iterations = %6.2f us per iteration" % (nrow, niso, best * 1e3, r["iters"][0], best * 1e6 / max(1, r["iters"][0])), flush=True)
