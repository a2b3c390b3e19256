#!/usr/bin/env python3
"""GPU box: the LPT shards of a world-8 (and world-4) strong-scaling C3 run, a rank at a time on one GPU, under plan
settings that trade lanes for iteration latency (environment read when the plan is made): the default plan, the
half tile for the wave kind, lane-rich later phases, both.  Per setting: ms per step (EM + epilogue) and the EM
kernels' own time by kind."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from strawberry_amd import dist, em, synth
ctx = em.default_context(0)
whole = synth.make_c3(seed=0x5743)
SETTINGS = [("default", {}), ("auto small-shard plan off", {"SBGPU_SMALL_SHARD": "0"}), ("half tile", {"SBGPU_WAVE_RMULT": "1", "SBGPU_SMALL_SHARD": "0"}),
            ("phases 64 / 0.25", {"SBGPU_PHASES": "64", "SBGPU_PHASE_LAMBDA": "0.25"}),
            ("phases 32 / 0.25", {"SBGPU_PHASES": "32", "SBGPU_PHASE_LAMBDA": "0.25"}),
            ("phases 16 / 0.1", {"SBGPU_PHASES": "16", "SBGPU_PHASE_LAMBDA": "0.1"}),
            ("half tile + phases 32 / 0.25", {"SBGPU_WAVE_RMULT": "1", "SBGPU_PHASES": "32", "SBGPU_PHASE_LAMBDA": "0.25"})]
KEYS = ["SBGPU_WAVE_RMULT", "SBGPU_PHASES", "SBGPU_PHASE_LAMBDA", "SBGPU_SMALL_SHARD"]
worlds = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8]
for world in worlds:
    shards = dist.shard_loci(whole.nrow, whole.niso, world)
    for rank in range(world):
        shard = whole.select(shards[rank])
        ref = None
        for name, env in SETTINGS:
            for k in KEYS:
                os.environ.pop(k, None)
            os.environ.update(env)
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
            s.set_timing(True); s.run_em(); s.synchronize(); kms = s.last_kernel_ms(); s.set_timing(False)
            r = s.results()
            if ref is None:
                ref = r
            same = (r["iters"] == ref["iters"]).all() and (r["status"] == ref["status"]).all()
            err = float(np.nanmax(np.abs(r["theta"] - ref["theta"]) / np.maximum(np.abs(ref["theta"]), 1e-9)))
            print("world %d rank %d  %5d loci  %-30s %.3f ms/step  kinds %s  capped %d  status+iters %s theta vs default %.1e" % (
                world, rank, shard.n_loci, name, ms, " ".join("%.3f" % x for x in kms), int((r["status"] == 3).sum()),
                "same" if same else "DIFFER", err), flush=True)
            del s, q
