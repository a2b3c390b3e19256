#!/usr/bin/env python3
"""GPU box, under `rocprofv3 --pmc ...`: the C3 batch's loci split by isoform count (the register-tile layout -- columns per
lane x column lanes -- is a function of it, plan.h::layout_for), each subset solved on its own, three launches each.  Writes
the order of the subsets and their algorithmic work (sum over loci of iterations x bins x isoforms) to a JSON beside the
counter file; tools/pmc_em_layouts.sh joins the two into instructions per algorithmic FMA by layout."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from strawberry_amd import em, synth
out = sys.argv[1]
ctx = em.default_context(0)
b = synth.make_c3()
runs = []
for ni in list(range(1, 25)) + [(25, 64)]:
    lo, hi = (ni, ni) if isinstance(ni, int) else ni
    idx = np.nonzero((b.niso >= lo) & (b.niso <= hi))[0]
    if len(idx) < 16:
        continue
    sub = b.select(idx)
    s = em.EmBatchSolver(sub, ctx)
    for _ in range(3):
        s.run_em(); s.synchronize()
    r = s.results()
    kinds = np.bincount(s.plan.locus_kinds(), minlength=6).tolist()
    cls = s.plan.classes() if hasattr(s.plan, "classes") else None
    runs.append({"niso": [lo, hi], "loci": int(len(idx)), "launches": 3, "kinds": kinds,
                 "elem_iters": int((sub.nrow * sub.niso * r["iters"].astype(np.int64)).sum()),
                 "elements": int((sub.nrow * sub.niso).sum()), "iters_sum": int(r["iters"].sum()), "capped": int((r["status"] == 3).sum())})
    del s
json.dump(runs, open(out, "w"))
print("subsets", len(runs))
