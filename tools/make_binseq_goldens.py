#!/usr/bin/env python3
"""Golden vectors for the per-bin sequence statistics from the REFERENCE's own
include/kmer.h templates (oracle/_ref/libstrawberry_ref.so, ref_kmer_stats in oracle/ref_shim.cpp).

Runs only where /root/reference is mounted.  Commits data only: tests/golden/binseq_cases.npz holds
sequences we generated (one byte string, offsets) and the reference's gc / entropy / flags for each.
All sequences are longer than 40 bases: the reference aborts below that (kmer.h:82)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import build  # noqa: E402
from oracle.lib import RefLib  # noqa: E402

ALPHA = np.frombuffer(b"ACGTacgtNn\x01\x02xR-", np.uint8)


def sequences(rng):
    out = []
    for t in range(600):                       # mixtures of GC content, with odd bytes
        n = int(rng.integers(41, 1500)) if t % 7 else int(rng.integers(4000, 9000))
        gc = float(rng.choice([0.1, 0.3, 0.5, 0.7, 0.8, 0.9, 0.97]))
        p = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])
        pp = np.concatenate([p * 0.9, p * 0.07, np.full(7, 0.03 / 7)])
        out.append(rng.choice(ALPHA, size=n, p=pp / pp.sum()).tobytes())
    # windows exactly at the cutoffs: 16 / 20, 17 / 20, 18 / 20, 19 / 20, 32 / 40, 33 / 40, 36 / 40, 37 / 40 G's
    for w, k in ((20, 16), (20, 17), (20, 18), (20, 19), (40, 32), (40, 33), (40, 36), (40, 37)):
        out.append(b"A" * 50 + b"G" * k + b"A" * (w - k) + b"A" * 50)
        out.append(b"AT" * 25 + (b"GA" * w)[:2 * (w - k)] + b"C" * (2 * k - w) + b"TA" * 25 if 2 * k >= w else b"A" * 60)
    out += [b"A" * 41, b"ACGT" * 11, b"G" * 100, b"N" * 64, b"acgtn" * 13, b"ACGTTGCA" * 700,
            (b"ACGT" * 16 + b"G") * 70]          # 4096-base boundary cases are built in the tests by slicing
    return out


def main():
    build(with_ref=True)
    ref = RefLib()
    rng = np.random.Generator(np.random.PCG64(0xA8))
    seqs = sequences(rng)
    off = np.zeros(len(seqs) + 1, np.int64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    res = [ref.kmer_stats(s) for s in seqs]
    path = os.path.join(ROOT, "tests", "golden", "binseq_cases.npz")
    np.savez_compressed(path, seq=np.frombuffer(b"".join(seqs), np.uint8), off=off,
                        gc=np.array([r[0] for r in res]), entropy=np.array([r[1] for r in res]),
                        flags=np.array([r[2] for r in res], np.uint8))
    print(path, os.path.getsize(path), "bytes,", len(seqs), "sequences,", int(off[-1]), "bases; flags histogram",
          np.bincount([r[2] for r in res], minlength=16))


if __name__ == "__main__":
    main()
