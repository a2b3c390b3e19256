#!/usr/bin/env python3
"""GPU box: random stress of the bin-sequence kernel (A8) against the oracle: random genomes (GC content in
stretches, lower case, N, odd bytes), bins of 1-300 segments of 1-3000 bases, lengths up to 200 000.
GC ratio and flags exact, entropy within 1e-12 relative.   python tools/stress_binseq.py [seeds]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from strawberry_amd.binseq import bin_sequence_stats
from oracle import OracleLib
o = OracleLib()
bad = tot_bins = tot_bases = 0
worst = 0.0
t0 = time.time()
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(200_000, 2_000_000))
    stretch = int(rng.choice([20, 50, 200, 1000]))
    gcp = np.repeat(rng.choice([0.05, 0.3, 0.5, 0.7, 0.85, 0.95, 1.0], size=n // stretch + 1), stretch)[:n]
    u = rng.random(n)
    g = np.where(u < gcp / 2, ord("C"), np.where(u < gcp, ord("G"), np.where(u < gcp + (1 - gcp) / 2, ord("A"), ord("T")))).astype(np.uint8)
    g[rng.random(n) < rng.random() * 0.3] |= 0x20
    g[rng.random(n) < rng.random() * 0.05] = ord("N")
    g[rng.random(n) < 0.003] = rng.integers(0, 256, dtype=np.uint8)
    start = int(rng.integers(1, 1_000_000))
    off, sl, sr = [0], [], []
    for b in range(int(rng.integers(200, 3000))):
        k = int(rng.choice([1, 1, 2, 3, 5, 9, int(rng.integers(10, 300))]))
        hi = int(rng.choice([3, 60, 400, 3000]))
        lens = rng.integers(1, hi + 1, size=k)
        gaps = rng.integers(0, 200, size=k)
        span = int((lens + gaps).sum())
        if span >= n:
            continue
        a = start + int(rng.integers(0, n - span))
        for ln, gp in zip(lens, gaps):
            sl.append(a); sr.append(a + int(ln) - 1)
            a += int(ln) + int(gp)
        off.append(len(sl))
    if seed % 3 == 0:                                   # one very long bin
        ln = int(rng.integers(60_000, min(200_000, n - 1)))
        sl.append(start); sr.append(start + ln - 1); off.append(len(sl))
    off, sl, sr = np.array(off, np.int64), np.array(sl, np.uint32), np.array(sr, np.uint32)
    gc, ent, fl = bin_sequence_stats(g.tobytes(), off, sl, sr, genome_start=start)
    ogc, oent, ofl = o.binseq_batch(g.tobytes(), start, off, sl, sr)
    rel = np.abs(ent - oent) / np.maximum(1.0, np.abs(oent))
    ok = np.array_equal(gc, ogc, equal_nan=True) and np.array_equal(fl, ofl) and (rel <= 1e-12).all()
    worst = max(worst, float(rel.max()))
    nb = len(off) - 1
    bases = int((sr.astype(np.int64) - sl + 1).sum())
    print("seed %d: genome %d, %d bins, %d segments, %d bases, flags %s  %s (entropy max rel %.1e)" % (
        seed, n, nb, len(sl), bases, np.bincount(ofl, minlength=16)[[0, 1, 3, 5, 7, 15]].tolist(), "ok" if ok else "MISMATCH", rel.max()), flush=True)
    bad += not ok
    tot_bins += nb
    tot_bases += bases
print("total %d bins, %d bases, %d failures, worst entropy error %.1e, %.1f s" % (tot_bins, tot_bases, bad, worst, time.time() - t0))
sys.exit(1 if bad else 0)
