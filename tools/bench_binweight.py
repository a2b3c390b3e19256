#!/usr/bin/env python3
"""GPU box: throughput of the bin-weight kernel on synthetic (bin, isoform)
pairs, next to the reference's own effective_len/emp_dist_pdf loop (oracle/_ref) or the C
restatement on one host thread.  Prints one JSON line."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from strawberry_amd import _lib, em  # noqa: E402
from strawberry_amd.binweight import InsertSize, pack_pairs  # noqa: E402


def make_pairs(n, seed=9):
    rng = np.random.Generator(np.random.PCG64(seed))
    nseg = rng.choice([1, 2, 3, 4, 5, 6, 7, 9, 12], n, p=[.15, .2, .15, .15, .1, .08, .07, .05, .05])
    seg_off = np.concatenate([[0], np.cumsum(nseg)]).astype(np.int64)
    seg = rng.integers(8, 400, int(seg_off[-1])).astype(np.uint32)
    mask = np.zeros(n, np.uint32)
    r = rng.random(n)
    m3 = nseg == 3
    mask[m3 & (r < .5)] = 2
    m4 = nseg == 4
    mask[m4] = np.array([0, 2, 4, 6], np.uint32)[(r[m4] * 4).astype(int)]
    big = np.nonzero(nseg >= 5)[0]
    for p in big:
        k = int(rng.integers(0, nseg[p] - 1))
        idx = rng.choice(np.arange(1, nseg[p] - 1), k, replace=False)
        mask[p] = np.bitwise_or.reduce((1 << idx).astype(np.uint32)) if k else 0
    tot = np.add.reduceat(seg.astype(np.int64), seg_off[:-1])
    iso_len = (tot + rng.integers(0, 3000, n)).astype(np.int32)
    return seg_off, seg, mask, iso_len, tot


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    ctx = em.default_context(0)
    dev = torch.device("cuda", 0)
    seg_off, seg, mask, iso_len, tot = make_pairs(n)
    ins = InsertSize(230.0, 35.0)
    rl = 75
    pdf = ins.pdf_table(int(tot.max()) + 1)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_off, d_seg, d_mask, d_len, d_pdf = d(seg_off), d(seg.view(np.int32)), d(mask.view(np.int32)), d(iso_len), d(pdf)
    d_out = torch.zeros(n, dtype=torch.float64, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def run():
        _lib.check(ctx.L.sbgpu_binweight_device(ctx.h, n, d_off.data_ptr(), d_seg.data_ptr(), d_mask.data_ptr(),
                                                d_len.data_ptr(), None, d_pdf.data_ptr(), len(pdf), rl, rl, 0,
                                                d_out.data_ptr(), st), "sbgpu_binweight_device")
    run()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        ev0.record()
        run()
        ev1.record()
        torch.cuda.synchronize()
        ts.append(ev0.elapsed_time(ev1))
    ms = min(ts)
    # algorithmic work: one (pdf, effective_len, divide, add) term per fragment length in [lmin, lmax]
    nseg = np.diff(seg_off)
    inner = tot - seg[seg_off[:-1]] - seg[seg_off[1:] - 1]
    lmin = np.where(nseg > 2, np.maximum(rl, inner), rl)
    terms = np.maximum(0, tot - lmin + 1).sum()
    in_bytes = seg.nbytes + seg_off.nbytes + mask.nbytes + iso_len.nbytes + 8 * n
    # CPU: reference loop on a sample, one thread
    from oracle import OracleLib, RefLib, have_ref
    m = 3000
    t = time.perf_counter()
    if have_ref():
        ref = RefLib()
        w = [ref.bin_weight(seg[seg_off[p]:seg_off[p + 1]], [k for k in range(32) if mask[p] >> k & 1],
                            int(iso_len[p]), rl, 230.0, 35.0) for p in range(m)]
        kind = "reference"
    else:
        o = OracleLib()
        oi = o.make_insert(230.0, 35.0)
        w = [o.bin_weight(seg[seg_off[p]:seg_off[p + 1]], [k for k in range(32) if mask[p] >> k & 1],
                          int(iso_len[p]), rl, oi) for p in range(m)]
        kind = "port"
    dt = time.perf_counter() - t
    got = d_out[:m].cpu().numpy()
    w = np.array(w)
    big = np.abs(w) > 1e-280
    err = float((np.abs(got[big] - w[big]) / np.abs(w[big])).max())
    print(json.dumps({
        "metric": "bin-weight pairs/s (A4)", "value": n / ms * 1e3, "unit": "pairs/s", "ms": ms, "pairs": n,
        "terms_per_s": float(terms) / ms * 1e3, "terms": int(terms), "input_bytes": int(in_bytes),
        "hbm_GBps": in_bytes / ms / 1e6, "max_rel_err_vs_cpu_sample": err,
        "cpu_baseline": {"value": m / dt, "unit": "pairs/s", "cores": 1, "kind": kind,
                         "sample": "first %d pairs, %.2f s (includes ctypes call overhead)" % (m, dt)}}))


if __name__ == "__main__":
    main()
