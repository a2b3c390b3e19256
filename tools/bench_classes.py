#!/usr/bin/env python3
"""Per-size-class iteration cost (GPU box): batches of same-shape loci whose counts
are so large that the absolute 1e-2 stop is never met, so every locus runs the
full 1000 iterations; reports kernel time / 1000 for a given number of loci."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from strawberry_amd import em, synth  # noqa: E402


def make(nrow, niso, n_loci, seed=1):
    rng = np.random.Generator(np.random.PCG64(seed))
    loci = []
    for _ in range(n_loci):
        F = np.where(rng.random((nrow, niso)) < 0.6, rng.uniform(1e-3, .3, (nrow, niso)), 0.0)
        F[:, 0] = np.maximum(F[:, 0], 1e-3)
        loci.append((rng.integers(10**8, 2 * 10**8, nrow).astype(np.int32), F))
    return synth.from_loci(loci)


def main():
    ctx = em.default_context(0)
    shapes = [(4, 2), (8, 4), (16, 4), (32, 8), (64, 8), (128, 8), (256, 8), (32, 16), (64, 16), (64, 32), (64, 64),
              (512, 8), (2000, 8), (1000, 20)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
    for nrow, niso in shapes:
        for n_loci in (64, 1024, 4096):
            if nrow * niso * n_loci > 3e7:
                continue
            b = make(nrow, niso, n_loci)
            s = em.EmBatchSolver(b, ctx)
            s.set_timing(True)
            s.run_em()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                s.run_em()
                torch.cuda.synchronize()
                best = min(best, max(s.last_kernel_ms()))
            r = s.results()
            cls = s.plan.classes()[0]
            print("%5dx%-3d n=%5d  kind %d C %2d R %2d G %3d waves %5d | %8.3f ms  iters mean %6.1f  -> %.3f us/iter" % (
                nrow, niso, n_loci, cls["kind"], cls["C"], cls["R"], cls["G"], cls["n_waves"], best,
                r["iters"].mean(), best * 1e3 / max(1, r["iters"].max())), flush=True)


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "mix"):
    main()


def mix():
    """Several shapes in ONE batch (one fused launch): does the mix slow the classes down?"""
    ctx = em.default_context(0)
    shapes = [(4, 2), (8, 4), (16, 4), (32, 8), (64, 8), (128, 8), (32, 16), (64, 16), (64, 32), (12, 3), (24, 6)]
    for per in (16, 128):
        parts = [make(nr, ni, per, seed=7 + i) for i, (nr, ni) in enumerate(shapes)]
        loci = []
        for b in parts:
            for l in range(b.n_loci):
                loci.append(b.locus(l))
        b = synth.from_loci(loci)
        s = em.EmBatchSolver(b, ctx)
        s.set_timing(True)
        s.run_em()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            s.run_em()
            torch.cuda.synchronize()
            best = min(best, max(s.last_kernel_ms()))
        r = s.results()
        print("mix of %d shapes x %d loci: %d classes, %.3f ms, iters mean %.1f max %d" % (
            len(shapes), per, s.plan.info()["n_classes"], best, r["iters"].mean(), r["iters"].max()), flush=True)
        for c in s.plan.classes():
            print("    ", c)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "mix":
    mix()
