#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own EmSolver.

Runs only where /root/reference is mounted (this build container): it needs
oracle/_ref/libstrawberry_ref.so (make -C oracle ref), i.e. the reference's
unmodified estimate.cpp compiled with the reference's flags.  The fixtures hold
numbers only: inputs (n, F as CSR-of-loci) and the reference outputs (theta at
full double precision, init/run flags).  Iteration counts are not observable
through EmSolver's interface, so they are not part of the goldens.

    python tools/make_goldens.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import RefLib, build  # noqa: E402
from strawberry_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def save(name, batch, ref):
    theta, flags = ref.em_batch(batch.row_off, batch.iso_off, batch.f_off, batch.count, batch.F)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, row_off=batch.row_off, iso_off=batch.iso_off, f_off=batch.f_off,
                        count=batch.count, F=batch.F, length=batch.length,
                        ref_theta=theta, ref_flags=flags)
    print("%-18s %5d loci  %8d elems  init_false=%d run_false=%d  -> %s (%d KB)" % (
        name, batch.n_loci, batch.F.size, int(((flags & 1) == 0).sum()),
        int((((flags & 1) == 1) & ((flags & 2) == 0)).sum()), os.path.relpath(path, ROOT),
        os.path.getsize(path) // 1024))


def edge_batch():
    """Hand-made edge cases (SURVEY 8(c) `em_edge`) + the survey's known-answer loci."""
    rng = np.random.Generator(np.random.PCG64(777))
    loci = []
    # known-answer tests captured from the oracle during the survey
    loci.append(([100, 50, 30], [[.002, .001], [.003, 0], [0, .004]]))            # toy
    loci.append(([0, 5], [[.1, 0], [0, .1]]))                                      # denom_zero
    loci.append(([3, 4], [[1e-5, 1e-6], [0, 1e-5]]))                               # all rows dropped
    loci.append(([10, 20], [[.2, .1, 0], [.05, .3, 0]]))                           # zero column
    loci.append(([7, 10, 20], [[1e-6, 1e-6], [.2, .1], [.05, .3]]))                # row dropped
    loci.append(([10, 20], [[.2], [.05]]))                                         # single isoform
    loci.append(([9], [[.2, .1, .4]]))                                             # single row
    # rows with n_i = 0 among normal rows
    F = rng.uniform(1e-3, .3, (12, 5)) * (rng.random((12, 5)) < .5)
    n = rng.integers(0, 40, 12)
    n[::3] = 0
    loci.append((n, F))
    # a row exactly at the 1e-5 threshold (dropped: needs > 1e-5) next to one just above
    loci.append(([5, 6, 7], [[1e-5, 1e-5], [1.0000001e-5, 0], [.1, .2]]))
    # all counts zero -> theta0 = 0 -> denominators 0 -> run() false
    loci.append(([0, 0, 0], [[.1, .2], [.3, .1], [.2, .2]]))
    # slow convergence: two nearly identical columns, large counts (hits the 1000 cap)
    base = rng.uniform(.01, .3, 40)
    F = np.stack([base, base * (1 + 1e-3 * rng.standard_normal(40)), rng.uniform(.01, .3, 40)], 1)
    loci.append((rng.integers(1000, 100000, 40), F))
    # nrow = 1000 x niso = 8 with n_i = 1 (un-binned shape)
    F = rng.uniform(1e-3, .3, (1000, 8)) * (rng.random((1000, 8)) < .4)
    loci.append((np.ones(1000, np.int64), F))
    # wide: 40 isoforms (beyond the 32-column register tile), 70 and 300 isoforms (streaming path)
    for niso, nrow in ((40, 90), (70, 150), (300, 64)):
        F = rng.uniform(1e-3, .3, (nrow, niso)) * (rng.random((nrow, niso)) < .3)
        loci.append((rng.integers(0, 200, nrow), F))
    # tall: 3000 and 20000 rows x 4 (workgroup-per-locus and streaming paths)
    for nrow in (3000, 20000):
        F = rng.uniform(1e-3, .3, (nrow, 4)) * (rng.random((nrow, 4)) < .6)
        loci.append((rng.integers(0, 30, nrow), F))
    # 1 x 1
    loci.append(([17], [[.25]]))
    # large-magnitude weights and counts
    F = rng.uniform(1e2, 1e4, (20, 6)) * (rng.random((20, 6)) < .5)
    loci.append((rng.integers(0, 2_000_000, 20), F))
    return synth.from_loci([(np.asarray(n, np.int32), np.asarray(F, np.float64)) for n, F in loci], name="edge")


def main():
    build(with_ref=True)
    ref = RefLib()
    os.makedirs(OUT, exist_ok=True)
    save("em_edge", edge_batch(), ref)
    save("em_random_256", synth.make_random(256, seed=12345), ref)
    c2 = synth.make_c2(n_loci=64)
    save("em_c2_64", c2, ref)
    c3 = synth.make_c3(n_loci=400, total_frags=400 / 60000 * 2e8)
    save("em_c3_400", c3, ref)


if __name__ == "__main__":
    main()
