#!/usr/bin/env python3
"""exonbin on BIG loci (65-128 segments): the 128-bit segment basis (exonbin_seg128_kernel) against the exon walk
(SBGPU_EXONBIN_SEGBASIS=0 in the environment selects the walk everywhere).  usage: bench_exonbin_big.py [n_loci=1500] [hits_per_locus=3000]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from strawberry_amd import em, exonbin as eb
n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
per = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
rng = np.random.default_rng(1)
loci, loc, feats = [], [], []
for l in range(n_loci):
    n_cells, n_iso = int(rng.integers(70, 128)), int(rng.integers(4, 30))
    base = 20000 * (l + 1)
    isoforms = [[(base + 100 * k, base + 100 * k + 79) for k in range(n_cells)]]
    for _ in range(n_iso - 1):
        keep = rng.random(n_cells) < 0.7
        isoforms.append([(base + 100 * k, base + 100 * k + 79) for k in range(n_cells) if keep[k]] or isoforms[0][:1])
    loci.append(isoforms)
    starts = np.sort(rng.integers(0, n_cells - 3, per))
    for s0 in starts.tolist():
        ex = isoforms[0]
        # a pair: left mate in cell s0 (spliced into s0+1 half of the time), gap, right mate in cell s0+2
        if rng.random() < 0.5:
            f = ([0, 1, 0, 2, 0], [ex[s0][0] + 40, ex[s0][1] + 1, ex[s0 + 1][0], ex[s0 + 1][0] + 35, ex[s0 + 2][0] + 5],
                 [ex[s0][1], ex[s0 + 1][0] - 1, ex[s0 + 1][0] + 34, ex[s0 + 2][0] + 4, ex[s0 + 2][0] + 79])
        else:
            f = ([0, 2, 0], [ex[s0][0] + 3, ex[s0][0] + 78, ex[s0 + 2][0] + 2], [ex[s0][0] + 77, ex[s0 + 2][0] + 1, ex[s0 + 2][0] + 76])
        loc.append(l), feats.append(f)
annot, hits = eb.Annotation(loci), eb.Hits(loc, feats)
ctx = em.default_context(0)
ts = []
for _ in range(6):
    torch.cuda.synchronize(); t = time.time()
    compat, key = eb.compat_and_keys(annot, hits, ctx)
    torch.cuda.synchronize(); ts.append(time.time() - t)
print(json.dumps({"loci": n_loci, "hits": hits.n_hits, "segments_per_locus": float(np.diff(annot.seg_off).mean()), "key_words": annot.key_words,
                  "segbasis": os.environ.get("SBGPU_EXONBIN_SEGBASIS", "1"), "call_ms_min": min(ts[1:]) * 1e3,
                  "compatible_frac": float((compat != 0).any(1).mean()), "note": "whole sbgpu_exonbin_host call incl. uploads and downloads"}))
