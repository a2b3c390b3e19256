#!/usr/bin/env python3
"""Throughput of the per-bin sequence statistics kernel (A8) on device-resident inputs, with its HBM
roofline and the oracle's loop timed on the host beside it.  One JSON line, same shape as bench.py's.

  python tools/bench_binseq.py [--bins 1000000] [--genome 100000000] [--steps 20] [--warmup 3]

Workload: bins of 1-6 segments of 40-400 bases (mean ~ 750 bases per bin) placed at random on a random
genome window -- the shape of a human annotation's exon bins."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bins", type=int, default=1_000_000)
    ap.add_argument("--genome", type=int, default=100_000_000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    import torch
    from strawberry_amd import _lib, em

    rng = np.random.Generator(np.random.PCG64(0xA8))
    genome = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=a.genome, p=[0.29, 0.21, 0.21, 0.29])
    nseg = rng.integers(1, 7, size=a.bins)
    off = np.concatenate([[0], np.cumsum(nseg)]).astype(np.int64)
    n_seg = int(off[-1])
    lens = rng.integers(40, 400, size=n_seg)
    gaps = rng.integers(1, 2000, size=n_seg)
    first = rng.integers(1, a.genome - 7 * 2400, size=a.bins)
    # segment k of bin b starts at first[b] + sum of the (len + gap) of the bin's earlier segments
    step = lens + gaps
    cs = np.cumsum(step) - step
    left = (np.repeat(first, nseg) + cs - np.repeat(cs[off[:-1]], nseg)).astype(np.uint32)
    right = (left + lens - 1).astype(np.uint32)
    n_bases = int(lens.sum())
    # algorithmic bytes: every base once, a segment's two coordinates, a bin's offset and its 17 bytes out
    alg_bytes = n_bases + 8 * n_seg + a.bins * (8 + 17)

    ctx = em.default_context(0)
    dev = torch.device("cuda", 0)
    t = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x).view(dt)).to(dev)   # noqa: E731
    d_genome, d_off, d_left, d_right = t(genome, np.uint8), t(off, np.int64), t(left, np.int32), t(right, np.int32)
    d_gc = torch.empty(a.bins, dtype=torch.float64, device=dev)
    d_ent = torch.empty(a.bins, dtype=torch.float64, device=dev)
    d_fl = torch.empty(a.bins, dtype=torch.uint8, device=dev)
    d_err = torch.zeros(1, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step_():
        _lib.check(ctx.L.sbgpu_binseq_device(ctx.h, d_genome.data_ptr(), 1, a.genome, a.bins, d_off.data_ptr(), d_left.data_ptr(),
                                             d_right.data_ptr(), d_gc.data_ptr(), d_ent.data_ptr(), d_fl.data_ptr(),
                                             d_err.data_ptr(), stream.cuda_stream), "sbgpu_binseq_device")
    for _ in range(a.warmup):
        step_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(a.steps):
        step_()
    e1.record(stream)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.steps
    kern_ms = e0.elapsed_time(e1) / a.steps
    assert int(d_err.item()) == 0
    out = {
        "metric": "bin sequence statistics, bins/s", "value": a.bins / wall, "unit": "bins/s", "n_gpus": 1, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": wall * 1e3, "higher_is_better": True, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "binseq", "bins": a.bins, "segments": n_seg, "bases": n_bases, "genome": a.genome},
        "roofline": {"bound": "hbm", "achieved": alg_bytes / (kern_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": alg_bytes / (kern_ms * 1e-3) / 1e9 / 8000.0, "traffic": None, "kernel_ms": kern_ms,
                     "bases_per_s": n_bases / (kern_ms * 1e-3)},
    }
    if not a.no_cpu_baseline:
        from oracle import OracleLib
        orc = OracleLib()
        n = min(a.bins, 20_000)
        t0 = time.perf_counter()
        ogc, oent, ofl = orc.binseq_batch(genome.tobytes(), 1, off[:n + 1], left[:off[n]], right[:off[n]])
        dt = time.perf_counter() - t0
        gc, ent, fl = d_gc[:n].cpu().numpy(), d_ent[:n].cpu().numpy(), d_fl[:n].cpu().numpy()
        same = bool((gc == ogc).all() and (fl == ofl).all() and (np.abs(ent - oent) <= 1e-12 * np.maximum(1, np.abs(oent))).all())
        out["cpu_baseline"] = {"value": n / dt, "unit": "bins/s", "cores": 1, "kind": "port",
                               "sample": "%d bins of the same batch, oracle/binseq_oracle.c, 1 thread" % n, "parity": same}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
