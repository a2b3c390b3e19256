#!/usr/bin/env python3
"""One rank's share of a strong-scaling run of the CHAIN (fragments -> abundances), on one GPU: ranks 0 and world - 1 of world sizes
1, 2, 4, 8 of the chain sample (strawberry_amd/chain.py: the SAME 60 000-locus sample, a rank keeps its loci's fragments), through
sbgpu_quantify_device.  Unlike the EM batch alone (tools/probe_strong_shard.py: a floor of one 1000-iteration locus), the chain's
kernels work per fragment, so a rank's time falls with its share."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from strawberry_amd import chain, em
ctx = em.default_context(0)
n_loci, n_frags = int(float(os.environ.get("SB_CHAIN_LOCI", "60000"))), float(os.environ.get("SB_CHAIN_FRAGS", "2e8"))
base = None
for world in (1, 2, 4, 8):
    for rank in sorted({0, world - 1}):
        q = chain.ChainQuantifier(ctx, n_loci=n_loci, n_frags=n_frags, seed=31, loci_subset=None if world == 1 else (rank, world))
        for _ in range(3):
            q.step()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            q.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / 10 * 1e3
        base = base or ms
        print("world %d rank %d: %6d loci %10d read pairs  %.3f ms/step  (%.2fx the whole sample's rate per rank-step)" % (
            world, rank, q.n_loci, int(q.n_frags), ms, base / ms), flush=True)
        q.close()
        del q
        torch.cuda.empty_cache()
