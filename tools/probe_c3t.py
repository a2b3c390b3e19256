#!/usr/bin/env python3
"""GPU box: C3-T (C3 + 300 loci of 65..400 isoforms) -- step time, per-kind kernel times, and the tail alone."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from strawberry_amd import em, synth

ctx = em.default_context(0)


def run(b, label, reps=2):
    s = em.EmBatchSolver(b, ctx)
    s.set_timing(True)
    s.run_em(); s.synchronize()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter()
        s.run_em(); s.synchronize()
        dt = (time.perf_counter() - t) * 1e3
        if dt < best:
            best, ms = dt, s.last_kernel_ms()
    r = s.results()
    el = (b.nrow * b.niso * r["iters"]).sum()
    print("%-40s %6d loci  %9.3f ms  kinds %s  iters mean %.0f max %d  %.2f TFLOP/s (4 flops per element-iteration)" % (
        label, b.n_loci, best, " ".join("%.2f" % x for x in ms), r["iters"].mean(), r["iters"].max(), 4 * el / best / 1e9), flush=True)
    return r


b = synth.make_c3t()
run(b, "C3-T")
tail = b.select(np.nonzero(b.niso > 64)[0])
r = run(tail, "its 300 tail loci alone")
for n in (1, 8, 32, 100):
    run(tail.select(np.arange(n)), "first %d tail loci" % n)
big = int(np.argmax(tail.nrow * tail.niso))
one = tail.select(np.array([big]))
rr = run(one, "largest tail locus %dx%d" % (tail.nrow[big], tail.niso[big]))
