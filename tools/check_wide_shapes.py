#!/usr/bin/env python3
"""GPU box: single wide loci of many shapes (every layout of em_wide_kernel, one and several workgroups, both exchange
forms) against the oracle: status and iteration count exact, theta to 1e-9."""
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
bad = 0
shapes = [(64, 400), (128, 194), (256, 100), (700, 40), (300, 70), (90, 96), (500, 97), (333, 129), (40, 193), (777, 257),
          (100, 385), (64, 512), (1500, 300), (2600, 130), (3000, 500), (5000, 65), (1, 100), (7, 300),
          # round 5: the layouts of 5 and 7 columns per lane, rows in LDS slots only partly filled, the last register block of 1-3 rows
          (300, 75), (801, 80), (900, 110), (450, 112), (1200, 150), (257, 160), (400, 210), (161, 224), (2000, 310),
          (129, 320), (800, 440), (81, 448), (513, 64), (20, 66), (577, 97), (97, 512), (4000, 400),
          (200, 128), (1500, 120), (417, 113), (300, 256), (1300, 240), (209, 225)]
for nrow, niso in shapes:
    b = _generate(rng, np.array([nrow], np.int64), np.array([niso], np.int64), np.array([nrow * 50], np.int64))
    s = em.EmBatchSolver(b, ctx)
    s.run_em(); s.synchronize()
    r = s.results()
    theta, status, iters = o.em_batch(b.row_off, b.iso_off, b.f_off, b.count, b.F, threads=8)
    err = float((np.abs(r["theta"] - theta) / np.maximum(np.abs(theta), 1e-9)).max())
    ok = r["status"][0] == status[0] and r["iters"][0] == iters[0] and err < 1e-9
    bad += not ok
    print("%5d x %3d: gpu status %d iters %4d | oracle status %d iters %4d | theta err %.2e %s" % (
        nrow, niso, r["status"][0], r["iters"][0], status[0], iters[0], err, "" if ok else "  <-- MISMATCH"), flush=True)
sys.exit(1 if bad else 0)
