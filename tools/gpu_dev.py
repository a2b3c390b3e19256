#!/usr/bin/env python3
"""Developer probe (GPU box): parity + timing of each config, prints one line per case."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from strawberry_amd import em, synth  # noqa: E402
from oracle import OracleLib  # noqa: E402


def bench(b, ctx, o, label, reps=5, check=True):
    s = em.EmBatchSolver(b, ctx)
    print(label, "plan", s.plan.info(), flush=True)
    for c in s.plan.classes():
        print("   class", c)
    s.run_em()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        ev0.record()
        s.run_em()
        ev1.record()
        torch.cuda.synchronize()
        ts.append(ev0.elapsed_time(ev1))
    r = s.results()
    ms = min(ts)
    print("%s: %d loci  best %.3f ms (all %s)  %.3g loci/s  %.1f Mfrag/s  algoGB/s %.1f" % (
        label, b.n_loci, ms, ["%.3f" % t for t in ts], b.n_loci / ms * 1e3, b.n_frags / ms * 1e-3,
        b.algorithmic_bytes() / ms * 1e-6), flush=True)
    print("   status", np.bincount(r["status"], minlength=4), "iters mean %.1f max %d" % (r["iters"].mean(), r["iters"].max()))
    if check:
        t = time.time()
        theta, status, iters = o.em_batch(b.row_off, b.iso_off, b.f_off, b.count, b.F, threads=os.cpu_count())
        dt = time.time() - t
        err = np.abs(r["theta"] - theta) / np.maximum(np.abs(theta), 1e-9)
        print("   oracle %d thr: %.2fs (%.0f loci/s)  status_eq %s iters_eq %s (ndiff %d) max_rel_err %.2e" % (
            os.cpu_count(), dt, b.n_loci / dt, (r["status"] == status).all(), (r["iters"] == iters).all(),
            int((r["iters"] != iters).sum()), err.max()), flush=True)
        bad = np.nonzero((r["status"] != status) | (r["iters"] != iters))[0]
        errl = np.maximum.reduceat(np.nan_to_num(err, nan=1e9), b.iso_off[:-1])
        bad = np.union1d(bad, np.nonzero(errl > 1e-9)[0])
        for l in bad[:20]:
            print("   BAD locus %d nrow %d niso %d: gpu st %d it %d | oracle st %d it %d | err %.2e" % (
                l, b.nrow[l], b.niso[l], r["status"][l], r["iters"][l], status[l], iters[l], errl[l]))


def main():
    ctx = em.default_context(0)
    print(ctx.device_info())
    o = OracleLib()
    which = sys.argv[1:] or ["c2", "c3", "c2u"]
    if "c2" in which:
        bench(synth.make_c2(), ctx, o, "C2")
    if "c3" in which:
        bench(synth.make_c3(), ctx, o, "C3")
    if "c2u" in which:
        bench(synth.make_c2(n_loci=2000, unbinned=True), ctx, o, "C2-U(2000)")


if __name__ == "__main__":
    main()
