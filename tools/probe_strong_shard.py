#!/usr/bin/env python3
"""One rank's share of a strong-scaling C3 run, on one GPU: the LPT shard 0 of world sizes 1, 2, 4, 8 through the EM + epilogue,
under different phase settings of the wave kind (SBGPU_PHASES / SBGPU_PHASE_LAMBDA, read when the plan is made).  With an
eighth of the loci the chip is no longer throughput-bound: what is left is the 1000-iteration loci's latency."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from strawberry_amd import dist, em, synth
ctx = em.default_context(0)
whole = synth.make_c3(seed=0x5743)
settings = [("", ""), ("32,128,512", "t,t,0.25"), ("64,256", "t,0.25"), ("128", "0.25"), ("256", "0.25"), ("64", "0.25"), ("32", "0.25"), ("32,256", "0.5,0.25"), ("16", "0.5")]
if len(sys.argv) > 1:
    settings = [tuple(a.split("/")) if "/" in a else (a, "") for a in sys.argv[1:]]
for world in (1, 2, 4, 8):
    shard = whole if world == 1 else whole.select(dist.shard_loci(whole.nrow, whole.niso, world)[0])
    ref = None
    for ph, lam in settings:
        for k, v in (("SBGPU_PHASES", ph), ("SBGPU_PHASE_LAMBDA", lam)):
            if v:
                os.environ[k] = v
            else:
                os.environ.pop(k, None)
        s = em.EmBatchSolver(shard, ctx)
        q = dist.ShardQuantifier(s, shard.n_frags, min_isoform_frac=0.0)
        for _ in range(5):
            q.step()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            q.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / 20 * 1e3
        r = s.results()
        if ref is None:
            ref = r
        same = bool(np.array_equal(r["theta"], ref["theta"], equal_nan=True) and (r["iters"] == ref["iters"]).all() and (r["status"] == ref["status"]).all())
        ndiff = int((~((r["theta"] == ref["theta"]) | (np.isnan(r["theta"]) & np.isnan(ref["theta"])))).sum())
        print("world %d  shard %5d loci  phases %-12s lambda %-10s  %.3f ms/step  maxiter loci %d  %s" % (
            world, shard.n_loci, ph or "-", lam or "-", ms, int((r["status"] == 3).sum()), "bitwise == one phase" if same else "DIFFERENT in %d theta (max rel %.1e), %d iteration counts" % (ndiff, float(np.nanmax(np.abs(r["theta"] - ref["theta"]) / np.maximum(np.abs(ref["theta"]), 1e-9))), int((r["iters"] != ref["iters"]).sum()))), flush=True)
        del s, q
