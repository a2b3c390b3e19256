#!/usr/bin/env python3
"""How well does a locus' SHAPE predict how many EM iterations it runs?  (CPU only: the oracle on the C3 batch.)

The plan orders work by plan.cpp::predicted_iterations(nrow, niso); this prints what that is built on: the mean
iteration count by nrow / niso and niso, and how much of the 1000-iteration loci the top x % of a score holds."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import OracleLib  # noqa: E402
from strawberry_amd import synth  # noqa: E402


def main():
    b = synth.make_c3()
    theta, status, iters = OracleLib().em_batch(b.row_off, b.iso_off, b.f_off, b.count, b.F, threads=8)
    nrow, niso = np.diff(b.row_off), np.diff(b.iso_off)
    r = nrow / niso
    print("%d loci, %d at the iteration cap, mean %.1f iterations" % (len(iters), (iters >= 1000).sum(), iters.mean()))
    w = np.select([r < 0.5, r < 0.75, r < 1.25, r < 1.5, r < 2, r < 3, r < 5, r < 10], [0.10, 0.60, 1.00, 0.80, 0.55, 0.35, 0.27, 0.20], 0.17)
    for name, x in (("nrow * niso (the old order)", (nrow * niso).astype(float)), ("niso", niso.astype(float)),
                    ("niso / nrow", niso / nrow), ("plan.cpp: w(nrow / niso) * (min(niso, 24) + 18)", w * (np.minimum(niso, 24) + 18))):
        order = np.argsort(-x, kind="stable")
        capped = iters[order] >= 1000
        print("%-44s top 5 / 10 / 20 %% hold %5.1f / %5.1f / %5.1f %% of the capped loci" % (
            name, *[100.0 * capped[:int(len(order) * f)].sum() / capped.sum() for f in (0.05, 0.1, 0.2)]))
    rb = [0, .5, .75, 1.25, 1.5, 2, 3, 5, 10, 1e9]
    nb = [1, 2, 3, 4, 6, 8, 12, 16, 24, 300]
    print("mean iterations / loci; rows: nrow / niso, columns: niso in", nb)
    for a, c in zip(rb[:-1], rb[1:]):
        cells = []
        for x, y in zip(nb[:-1], nb[1:]):
            m = (r >= a) & (r < c) & (niso >= x) & (niso < y)
            cells.append("%4.0f/%-5d" % (iters[m].mean() if m.sum() else 0, m.sum()))
        print("[%4g,%5g) " % (a, c), " ".join(cells))


if __name__ == "__main__":
    main()
