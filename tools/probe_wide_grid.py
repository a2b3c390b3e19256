#!/usr/bin/env python3
"""GPU box: single wide loci on a grid of shapes (argv: "nrow x niso" pairs) against the oracle; an error of a shape does
not stop the others."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from strawberry_amd import em
from strawberry_amd.synth import _generate
from oracle import OracleLib
ctx = em.default_context(0)
o = OracleLib()
rng = np.random.Generator(np.random.PCG64(5))
for arg in sys.argv[1:]:
    nrow, niso = [int(x) for x in arg.split("x")]
    b = _generate(rng, np.array([nrow], np.int64), np.array([niso], np.int64), np.array([nrow * 50], np.int64))
    try:
        s = em.EmBatchSolver(b, ctx)
        s.run_em(); s.synchronize()
        r = s.results()
    except Exception as e:
        print("%5d x %3d: ERROR %s" % (nrow, niso, str(e)[:80]), flush=True)
        continue
    theta, status, iters = o.em_batch(b.row_off, b.iso_off, b.f_off, b.count, b.F, threads=8)
    err = float((np.abs(r["theta"] - theta) / np.maximum(np.abs(theta), 1e-9)).max())
    ok = r["status"][0] == status[0] and r["iters"][0] == iters[0] and err < 1e-9
    print("%5d x %3d: gpu status %d iters %4d | oracle status %d iters %4d | theta err %.2e %s" % (
        nrow, niso, r["status"][0], r["iters"][0], status[0], iters[0], err, "" if ok else "  <-- MISMATCH"), flush=True)
