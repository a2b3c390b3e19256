#!/usr/bin/env python3
"""GPU box: the host-buffer entry sbgpu_em_batch (plan + H2D + solve + D2H) on C3 and C2, with its parts."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from strawberry_amd import em, synth
ctx = em.default_context(0)
for name, b in (("c3", synth.make_c3()), ("c2", synth.make_c2())):
    em.em_batch_host(b, ctx)
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); em.em_batch_host(b, ctx); t.append(time.perf_counter() - t0)
    tp = []
    for _ in range(5):
        t0 = time.perf_counter(); p = em.Plan(ctx, b.row_off, b.iso_off, b.f_off); tp.append(time.perf_counter() - t0); p.close()
    print("%s: sbgpu_em_batch %.2f ms (min of 5), of which sbgpu_plan_create %.2f ms; input %.1f MB" % (
        name, min(t) * 1e3, min(tp) * 1e3, (b.F.nbytes + b.count.nbytes) / 1e6))
