#!/usr/bin/env python3
"""GPU box: the whole fragments -> abundances chain on one GPU, stage by stage (A5 kernel, host
bookkeeping, A4 kernel, EM + epilogue), with the host bookkeeping at several thread counts.
One JSON line.   python tools/bench_chain.py [--hits 4000000]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hits", type=int, default=4_000_000)
    a = ap.parse_args()
    import torch
    import exonbin_util as XU
    from strawberry_amd import em, synth
    from strawberry_amd import exonbin as eb
    from strawberry_amd.quantify import InsertSize, LocusQuantifier

    loci = synth.make_gene_models(500, seed=21)
    hl, pairs = synth.make_fragments(loci, 200, seed=22)
    feats, loc = [], []
    for l, (lb, rb) in zip(hl, pairs):
        f = eb.hit_features(lb, rb)
        if f is not None:
            feats.append(f)
            loc.append(l)
    annot, hits = eb.Annotation(loci), eb.Hits(loc, feats)
    annot, hits = XU.tile(annot, hits, max(1, a.hits // hits.n_hits))
    ctx = em.default_context(0)
    sync = torch.cuda.synchronize

    def timed(fn):
        sync()
        t = time.perf_counter()
        r = fn()
        sync()
        return (time.perf_counter() - t) * 1e3, r

    q = LocusQuantifier(annot, hits, InsertSize(250.0, 30.0), 75, ctx=ctx, device_bins=False)
    q.run(hits.n_hits, min_isoform_frac=0.0)      # warm-up: code objects, allocator
    stages = {}
    host = {}
    for nt in (1, 8, 32, 64):
        os.environ["SBGPU_HOST_THREADS"] = str(nt)
        ms, bins = timed(q.assign_bins)
        host[str(nt)] = ms
    os.environ.pop("SBGPU_HOST_THREADS")
    host_bins = bins
    q = LocusQuantifier(annot, hits, InsertSize(250.0, 30.0), 75, ctx=ctx, device_bins=True)
    q.assign_bins()
    assert q.bins_on_device and (q.bins.count == host_bins.count).all() and (q.bins.pair_out_index == host_bins.pair_out_index).all()
    stages["assign_bins (A5 kernel + grouping on the device + pairs on the host)"], bins = timed(q.assign_bins)
    stages["bin_weights (upload pairs + A4 kernel)"], _ = timed(q.bin_weights)
    stages["solve (plan + EM + epilogue + D2H)"], res = timed(lambda: q.solve(hits.n_hits, min_isoform_frac=0.0))
    total = sum(stages.values())
    # the same chain as ONE C-ABI call with host buffers in and out (what a C/C++ driver uses)
    import ctypes as C
    from strawberry_amd import _lib
    an, ht = annot._struct(), hits._struct()
    ins = InsertSize(250.0, 30.0)._struct(75)
    n_iso = int(annot.iso_off[-1])
    theta = np.zeros(n_iso + 1); status = np.zeros(annot.n_loci + 1, np.int32); iters = np.zeros(annot.n_loci + 1, np.int32)
    used = _lib.sbgpu_insert_t()

    def one_call():
        h = C.c_void_p()
        _lib.check(ctx.L.sbgpu_quantify_host(ctx.h, C.byref(an), C.byref(ht), hits.mass.ctypes.data, C.byref(ins), 75, 0,
                                             theta.ctypes.data, status.ctypes.data, iters.ctypes.data, None, C.byref(used),
                                             C.byref(h)), "sbgpu_quantify_host")
        ctx.L.sbgpu_bins_destroy(h)
    one_call()
    one_ms = min(timed(one_call)[0] for _ in range(3))
    assert np.allclose(theta[:n_iso], res["theta"], rtol=1e-12, atol=1e-12)
    print(json.dumps({
        "metric": "fragments/s, fragments -> abundances chain", "value": hits.n_hits / total * 1e3, "unit": "fragments/s",
        "hits": hits.n_hits, "loci": annot.n_loci, "bins": int(bins.n_bins), "pairs": int(bins.n_pairs),
        "stage_ms": stages, "total_ms": total, "sbgpu_quantify_host_ms": one_ms,
        "sbgpu_quantify_host_fragments_per_s": hits.n_hits / one_ms * 1e3, "assign_bins_ms_with_host_grouping_by_threads": host, "host_cores": os.cpu_count(),
        "em_mean_iters": float(res["iters"].mean())}))


if __name__ == "__main__":
    main()
