#!/usr/bin/env python3
"""One rank's share of a strong-scaling run of records -> theta (bench.py --workload c3-front --gpus N), on one GPU: ranks 0
and world - 1 of world sizes 1, 2, 4, 8 of the chain sample's alignment records (strawberry_amd/front.py: the SAME sample,
locus l -- its cluster and its records -- on rank l mod world), through decode -> read stream -> pairs -> unique hits -> chain.
Every stage works per record / per pair, so a rank's time falls with its share (compare tools/probe_strong_chain.py)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from strawberry_amd import em, front
ctx = em.default_context(0)
n_loci, n_frags = int(float(os.environ.get("SB_FRONT_LOCI", "60000"))), float(os.environ.get("SB_FRONT_FRAGS", "2e8"))
base = None
for world in (1, 2, 4, 8):
    for rank in sorted({0, world - 1}):
        q = front.FrontQuantifier(ctx, n_loci=n_loci, n_frags=n_frags, seed=31, loci_subset=None if world == 1 else (rank, world))
        torch.cuda.empty_cache()
        for _ in range(2):
            q.step()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(4):
            q.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / 4 * 1e3
        base = base or ms
        print("world %d rank %d: %6d loci %10d records  %8.3f ms/step  (%.2fx the whole sample's rate per rank-step)  %s" % (
            world, rank, q.n_loci, q.n_records, ms, base / ms, " ".join("%s %.1f" % (k.split("_")[0], v) for k, v in q.stage_wall_ms.items())), flush=True)
        q.close()
        del q
        torch.cuda.empty_cache()
        ctx.L.sbgpu_release_idle_memory()
