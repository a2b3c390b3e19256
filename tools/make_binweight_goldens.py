#!/usr/bin/env python3
"""tests/golden/binweight_pairs.npz from the REFERENCE's own ExonBin::effective_len and
InsertSize::emp_dist_pdf (oracle/_ref, see oracle/ref_shim.cpp::ref_bin_weight).  Runs only
where /root/reference is mounted.  Numbers only: pair descriptions + reference weights."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import RefLib, build  # noqa: E402
from strawberry_amd.binweight import pack_pairs  # noqa: E402


def random_pairs(rng, n):
    segs, imps, lens = [], [], []
    for _ in range(n):
        nseg = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 9, 12], p=[.15, .2, .15, .15, .1, .08, .07, .05, .05]))
        s = rng.integers(8, 400, nseg)
        if nseg <= 2:
            imp = []
        elif nseg == 3:
            imp = [1] if rng.random() < .5 else []
        elif nseg == 4:
            imp = [[], [1], [2], [1, 2]][int(rng.integers(0, 4))]
        else:
            imp = sorted(rng.choice(np.arange(1, nseg - 1), int(rng.integers(0, nseg - 1)), replace=False).tolist())
        segs.append(s)
        imps.append(imp)
        lens.append(int(s.sum() + rng.integers(0, 3000)))
    return segs, imps, np.array(lens, np.int32)


def main():
    build(with_ref=True)
    ref = RefLib()
    rng = np.random.Generator(np.random.PCG64(2024))
    segs, imps, iso_len = random_pairs(rng, 1500)
    seg_off, seg_lens, mask = pack_pairs(segs, imps)
    out = {}
    # Gaussian law (-i 230/35), read length 75
    out["w_gauss"] = np.array([ref.bin_weight(s, i, L, 75, 230.0, 35.0) for s, i, L in zip(segs, imps, iso_len)])
    # empirical law from a fragment-length sample, read length 50
    frag = np.rint(rng.normal(210, 30, 4000)).astype(np.int32)
    frag = frag[(frag > 60) & (frag < 400)]
    out["w_emp"] = np.array([ref.bin_weight(s, i, L, 50, 0.0, 0.0, frag) for s, i, L in zip(segs, imps, iso_len)])
    path = os.path.join(ROOT, "tests", "golden", "binweight_pairs.npz")
    np.savez_compressed(path, seg_off=seg_off, seg_lens=seg_lens, implicit_mask=mask, iso_len=iso_len,
                        frag_lens=frag, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KB;", len(iso_len), "pairs; nonzero gauss",
          int((out["w_gauss"] != 0).sum()), "emp", int((out["w_emp"] != 0).sum()))


if __name__ == "__main__":
    main()
