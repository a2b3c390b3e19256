#!/usr/bin/env python3
"""GPU box: ONE wide locus of a given shape, `reps` solves -- for rocprofv3 counter passes of em_wide_kernel
(instructions per iteration = counter / (waves x iterations x launches))."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from strawberry_amd import em
from strawberry_amd.synth import _generate
nrow, niso, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = em.default_context(0)
rng = np.random.Generator(np.random.PCG64(5))
b = _generate(rng, np.array([nrow], np.int64), np.array([niso], np.int64), np.array([nrow * 50], np.int64))
s = em.EmBatchSolver(b, ctx)
best = 1e9
for _ in range(reps):
    t = time.perf_counter(); s.run_em(); s.synchronize(); best = min(best, time.perf_counter() - t)
r = s.results()
print("%d x %d: %.3f ms, %d iterations, %.2f us per iteration" % (nrow, niso, best * 1e3, r["iters"][0], best * 1e6 / r["iters"][0]))
