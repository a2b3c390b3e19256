#!/usr/bin/env python3
"""GPU box: the chain workload's step under SBGPU_HOST_TIMING=2 -- the host clock at every stage mark of
sbgpu_quantify_device (chain_api.hip, exonbin_api.hip), WITHOUT the synchronisations of =1: where the host thread
waits, and what it does between the kernels.  Then the same steps untimed.  DESIGN.md §3.8 cites the output."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SBGPU_HOST_TIMING"] = "2"
import torch
from strawberry_amd import chain, em
n_loci = int(float(sys.argv[1])) if len(sys.argv) > 1 else 60000
n_frags = float(sys.argv[2]) if len(sys.argv) > 2 else 2e8
ctx = em.default_context(0)
q = chain.ChainQuantifier(ctx, n_loci=n_loci, n_frags=n_frags)
for k in range(4):
    torch.cuda.synchronize()
    t = time.time(); q.step(); torch.cuda.synchronize()
    print("---- step %d: %.2f ms; kernels %s" % (k, (time.time() - t) * 1e3, {n: round(v, 3) for n, v in q.stage_ms().items()}), file=sys.stderr, flush=True)
