#!/usr/bin/env python3
"""GPU box: tools/stress_em.py's seeds 48..159 again, printing every locus whose status or iteration count differs from the
oracle's (kind, shape, both values, the theta difference) and whether a second run repeats it."""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from strawberry_amd import em, synth
from oracle import OracleLib
o = OracleLib(); ctx = em.default_context(0)
for seed in range(48, 160):
    parts = [synth.make_random(n_loci=3000, max_nrow=80, max_niso=12, density=0.35, max_count=80, seed=1000 + seed),
             synth.make_random(n_loci=400, max_nrow=600, max_niso=40, density=0.2, max_count=30, seed=2000 + seed),
             synth.make_random(n_loci=40, max_nrow=1500, max_niso=150, density=0.15, max_count=20, seed=3000 + seed),
             synth.make_random(n_loci=300, max_nrow=6, max_niso=3, density=0.9, max_count=3, seed=4000 + seed)]
    loci = [p.locus(l) for p in parts for l in range(p.n_loci)]
    b = synth.from_loci(loci)
    s = em.EmBatchSolver(b, ctx); s.run_em(); r = s.results()
    th, st, it = o.em_batch(b.row_off, b.iso_off, b.f_off, b.count, b.F, threads=64)
    bad = np.nonzero((r["status"] != st) | (r["iters"] != it))[0]
    if bad.size:
        k = s.plan.locus_kinds()
        for l in bad:
            i0, i1 = b.iso_off[l], b.iso_off[l + 1]
            err = np.abs(r["theta"][i0:i1] - th[i0:i1]).max()
            print("seed", seed, "locus", int(l), "kind", int(k[l]), "nrow", int(b.nrow[l]), "niso", int(b.niso[l]), "status", int(r["status"][l]), int(st[l]), "iters", int(r["iters"][l]), int(it[l]), "max abs theta diff", err, flush=True)
        # run again: deterministic?
        s2 = em.EmBatchSolver(b, ctx); s2.run_em(); r2 = s2.results()
        print("  second run same as first:", bool((r2["iters"] == r["iters"]).all() and (r2["status"] == r["status"]).all()), flush=True)
print("done")
