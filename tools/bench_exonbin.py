#!/usr/bin/env python3
"""Throughput of the exon-bin kernel (A5) on device-resident inputs, with its HBM roofline and the
oracle's loop timed on the host beside it.  One JSON line, same shape as bench.py's.

  python tools/bench_exonbin.py [--hits 8000000] [--steps 20] [--warmup 3] [--wide]"""
import argparse
import ctypes as C
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
    ap.add_argument("--hits", type=int, default=8_000_000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--only", choices=["compat", "key"], help="diagnostic: run only one of the two tests")
    a = ap.parse_args()
    import torch
    import exonbin_util as XU
    from strawberry_amd import _lib, em, synth
    from strawberry_amd import exonbin as eb
    from strawberry_amd.quantify import LocusQuantifier
    from strawberry_amd.binweight import InsertSize

    loci = synth.make_gene_models(500, seed=21)
    hl, pairs = synth.make_fragments(loci, 200, seed=22)
    feats, loc = [], []
    for l, (lb, rb) in zip(hl, pairs):
        f = eb.hit_features(lb, rb)
        if f is not None:
            feats.append(f)
            loc.append(l)
    annot, hits = eb.Annotation(loci), eb.Hits(loc, feats)
    copies = max(1, a.hits // hits.n_hits)
    annot, hits = XU.tile(annot, hits, copies)
    ctx = em.default_context(0)
    q = LocusQuantifier(annot, hits, InsertSize(250.0, 30.0), 75, ctx=ctx)
    cw, kw = annot.compat_words, annot.key_words
    n_feat = int(hits.feat_off[-1])
    if a.only == "compat":
        kw = 0
    if a.only == "key":
        cw = 0
    # algorithmic bytes per hit: its features (1 + 4 + 4 B each), its offset and locus, its words out
    alg_bytes = n_feat * 9 + hits.n_hits * (8 + 4 + 4 * (cw + kw))
    dev = torch.device("cuda", 0)
    d_compat = torch.zeros((hits.n_hits, max(cw, 1)), dtype=torch.int32, device=dev)
    d_key = torch.zeros((hits.n_hits, max(kw, 1)), dtype=torch.int32, device=dev)
    an = _lib.sbgpu_annotation_t(annot.n_loci, q._p("iso_off"), q._p("exon_off"), q._p("exon_left"), q._p("exon_right"),
                                 q._p("seg_off"), q._p("seg_left"), q._p("seg_right"))
    ht = _lib.sbgpu_hits_t(hits.n_hits, q._p("hit_locus"), q._p("feat_off"), q._p("feat_code"), q._p("feat_left"),
                           q._p("feat_right"))
    stream = torch.cuda.current_stream(dev)

    def step():
        _lib.check(ctx.L.sbgpu_exonbin_device(ctx.h, C.byref(an), C.byref(ht), cw, kw, d_compat.data_ptr(), d_key.data_ptr(),
                                              C.c_void_p(stream.cuda_stream)), "sbgpu_exonbin_device")
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(a.steps):
        step()
    e1.record(stream)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.steps
    kern_ms = e0.elapsed_time(e1) / a.steps
    out = {
        "metric": "exon-bin hits/s", "value": hits.n_hits / wall, "unit": "hits/s", "n_gpus": 1, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": wall * 1e3, "higher_is_better": True, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "exonbin", "hits": hits.n_hits, "loci": annot.n_loci, "features": n_feat,
                   "compat_words": cw, "key_words": kw},
        "roofline": {"bound": "hbm", "achieved": alg_bytes / (kern_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": alg_bytes / (kern_ms * 1e-3) / 1e9 / 8000.0, "traffic": None, "kernel_ms": kern_ms},
    }
    if not a.no_cpu_baseline and not a.only:
        from oracle import OracleLib
        orc = OracleLib()
        n = min(hits.n_hits, 2_000_000)
        sub = eb.Hits.from_arrays(hits.hit_locus[:n], hits.feat_off[:n + 1], hits.feat_code, hits.feat_left, hits.feat_right)
        t0 = time.perf_counter()
        oc, ok = orc.exonbin_batch(annot, sub)
        dt = time.perf_counter() - t0
        same = bool((oc == d_compat[:n].cpu().numpy().view(np.uint32)).all() and (ok == d_key[:n].cpu().numpy().view(np.uint32)).all())
        out["cpu_baseline"] = {"value": n / dt, "unit": "hits/s", "cores": 1, "kind": "port",
                               "sample": "%d hits of the same batch, oracle/exonbin_oracle.c, 1 thread" % n, "parity": same}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
