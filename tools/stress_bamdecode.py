#!/usr/bin/env python3
"""GPU box: random stress of the BAM record decoder: sbgpu_bam_decode_device against the oracle (oracle/bamdecode_oracle.c)
over many seeds -- the mixed bag of tests/bam_util.py (every CIGAR operation, tag type, flag), noise behind valid size words,
records too long for the staging buffer, random option sets -- and the host entry on the same bytes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bam_util as B
from strawberry_amd import bam, em
from oracle import OracleLib
o = OracleLib()
ctx = em.default_context(0)
bad = tot = 0
t0 = time.time()
KEYS = ("status", "record", "read_id", "ref", "left", "right", "partner_pos", "flags", "nh", "nm", "read_len", "sam_flag", "block_off", "block_left",
        "block_right")
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    rng = np.random.default_rng(5000 + seed)
    recs = B.random_records(rng, int(rng.integers(2000, 20000)))
    if seed % 3 == 0:
        recs += B.garbage_records(rng, int(rng.integers(500, 5000)))
    if seed % 4 == 0:
        recs += [B.record(int(rng.integers(0, 3)), int(rng.integers(0, 10 ** 6)), 0, "long%d" % k, [("M", int(rng.integers(20000, 60000)))],
                          tags=[("NH", "C", 1)]) for k in range(int(rng.integers(1, 80)))]
    order = rng.permutation(len(recs))
    raw = np.frombuffer(b"".join(recs[int(k)] for k in order), np.uint8)
    kw = dict(min_intron=int(rng.choice([1, 20, 50])), max_intron=int(rng.choice([500, 4000, 300000])), unique_only=bool(rng.integers(0, 2)),
              library=int(rng.integers(0, 3)), n_ref=int(rng.choice([0, 2, 3])))
    off = bam.index(raw)
    want = o.bam_decode(raw, off, **kw)
    d = bam.decode(raw, off, bam.BamOptions(**kw), device=ctx)
    h = bam.decode(raw, off, bam.BamOptions(**kw))
    a = np.flatnonzero(want["status"] == 0)
    ok = (np.array_equal(d.status, want["status"]) and np.array_equal(d.record, a) and np.array_equal(d.read_id, want["read_id"][a])
          and np.array_equal(d.left, want["left"][a]) and np.array_equal(d.right, want["right"][a]) and np.array_equal(d.nh, want["nh"][a])
          and np.array_equal(d.block_left, want["feat_left"][want["feat_code"] == 0]) and np.array_equal(d.block_right, want["feat_right"][want["feat_code"] == 0])
          and all(np.array_equal(getattr(d, k), getattr(h, k)) for k in KEYS) and d.any_paired == bool(want["any_paired"]))
    bad += not ok
    tot += off.size - 1
    d.close(), h.close()
    if seed % 8 == 7 or not ok:
        print("seed %d: %d records (%d accepted) %s  %s" % (seed, off.size - 1, a.size, kw, "ok" if ok else "MISMATCH"))
print("total %d records, %d failures, %.1f s" % (tot, bad, time.time() - t0))
sys.exit(1 if bad else 0)
