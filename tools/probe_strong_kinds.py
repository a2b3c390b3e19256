#!/usr/bin/env python3
"""Per-kind EM kernel times of one rank's strong-scaling shard (world 1, 2, 4, 8; every rank of world 8)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from strawberry_amd import dist, em, synth
ctx = em.default_context(0)
whole = synth.make_c3(seed=0x5743)
for world, ranks in ((1, [0]), (2, [0]), (4, [0]), (8, list(range(8)))):
    parts = dist.shard_loci(whole.nrow, whole.niso, world) if world > 1 else [np.arange(whole.n_loci)]
    for rank in ranks:
        shard = whole.select(parts[rank]) if world > 1 else whole
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
        s.set_timing(True)
        probe = []
        for _ in range(5):
            q.step()
            probe.append(s.last_kernel_ms())
        s.set_timing(False)
        kinds = np.bincount(s.plan.locus_kinds(), minlength=6)
        r = s.results()
        print("world %d rank %d  %5d loci  %.3f ms/step  kernel ms by kind %s  loci by kind %s  capped %d  elements %d" % (
            world, rank, shard.n_loci, ms, " ".join("%.3f" % x for x in np.mean(np.array(probe), 0)), kinds.tolist(), int((r["status"] == 3).sum()),
            int((shard.nrow * shard.niso).sum())), flush=True)
