#!/usr/bin/env python3
"""GPU box: random clusters of alignment records (tests/matepair_util.py: proper pairs, duplicates, single reads, orphans, mates
elsewhere, strands that disagree, reads aligned at several places under one id, a record beyond kMaxFragSpan) through
sbgpu_pair_mates_device -- the positional form where it serves, the sorted form where it steps aside -- against the host form:
the same pairs in the same order, the same counts.  usage: stress_pairing.py [seeds=40]"""
import os
import sys
import time

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import matepair_util as MU  # noqa: E402
from strawberry_amd import em, exonbin as eb  # noqa: E402

ctx = em.default_context(0)
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
t0, bad, n_rec, n_pos = time.time(), 0, 0, 0
for seed in range(n_seeds):
    rng = np.random.default_rng(9000 + seed)
    n_loci = int(rng.integers(1, 60))
    clusters = [MU.random_cluster(rng, int(rng.integers(0, 4000 if rng.random() < 0.1 else 300)), base=500000 * (l + 1), exotic=rng.random() < 0.8)
                for l in range(n_loci)]
    if rng.random() < 0.3:      # a read aligned twice at ONE place somewhere: the positional form must step aside
        c = clusters[int(rng.integers(0, n_loci))]
        if len(c) > 4:
            k = int(rng.integers(0, len(c)))
            c.insert(k, dict(c[k]))
    loc = [l for l, c in enumerate(clusters) for _ in c]
    if not loc:
        continue
    reads = eb.Reads(loc, *MU.arrays([r for c in clusters for r in c]))
    got, host = eb.pair_mates(n_loci, reads, device=ctx), eb.pair_mates(n_loci, reads)
    ok = all(np.array_equal(got[k], host[k]) for k in ("pair_off", "mass", "left_off", "right_off"))
    ok = ok and all(np.array_equal(x, y) for side in ("left", "right") for x, y in zip(got[side], host[side]))
    ok = ok and all(got["info"][k] == host["info"][k] for k in ("pairs", "complete", "single", "refused", "orphan"))
    n_rec += len(loc)
    n_pos += int(got["positional"])
    if not ok:
        bad += 1
        print("seed %d: MISMATCH (positional %s, why_sorted %d)" % (seed, got["positional"], got["why_sorted"]), flush=True)
print("total %d records in %d calls (%d served by the positional form), %d failures, %.1f s" % (n_rec, n_seeds, n_pos, bad, time.time() - t0))
sys.exit(1 if bad else 0)
