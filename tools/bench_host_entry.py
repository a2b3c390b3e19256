#!/usr/bin/env python3
"""sbgpu_quantify_host at chain scale: the PCIe-inclusive fragments -> abundances call (hits in pageable host memory,
as a C++ driver holds them) next to sbgpu_quantify_device on the same hits resident in HBM.

    python tools/bench_host_entry.py [n_loci] [n_frags] [reps]          (SBGPU_HOST_TIMING=1: stage times on stderr)
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from strawberry_amd import _lib, chain, em
    n_loci = int(float(sys.argv[1])) if len(sys.argv) > 1 else 60000
    n_frags = float(sys.argv[2]) if len(sys.argv) > 2 else 2e8
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    ctx = em.default_context(0)
    q = chain.ChainQuantifier(ctx, n_loci=n_loci, n_frags=n_frags, seed=31, pin=True)
    for _ in range(2):
        q.step()
    t = time.perf_counter()
    for _ in range(reps):
        q.step()
    dev_ms = (time.perf_counter() - t) / reps * 1e3
    out = q.host_entry(reps)
    out.update({"loci": q.n_loci, "read_pairs": q.n_frags, "unique_hits": q.n_hits, "quantify_device_ms": dev_ms,
                "pcie_inclusive_over_resident": out["ms_per_call"] / dev_ms})
    print(json.dumps(out))
    q.close()


if __name__ == "__main__":
    main()
