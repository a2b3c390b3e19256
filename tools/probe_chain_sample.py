#!/usr/bin/env python3
"""GPU box: the chain workload's sample (distinct gene models, fragments drawn on the device) -- generation time, shape,
two timed steps with the library's stage times, and the first loci of the sample against the oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from strawberry_amd import chain, em, exonbin as eb
from strawberry_amd.quantify import quantify_host
n_loci = int(float(sys.argv[1])) if len(sys.argv) > 1 else 60000
n_frags = float(sys.argv[2]) if len(sys.argv) > 2 else 2e8
ctx = em.default_context(0)
t = time.time()
q = chain.ChainQuantifier(ctx, n_loci=n_loci, n_frags=n_frags)
print("sample: %.1f s; %d loci, %d isoforms, %d read pairs in %d unique hits, %.2f features per hit, max memory %.1f GB" % (
    time.time() - t, q.n_loci, q.n_iso, q.n_frags, q.n_hits, q.hits.n_features / max(q.n_hits, 1), torch.cuda.max_memory_allocated() / 2**30), flush=True)
for _ in range(3):
    t = time.time(); q.step(); print("step %.2f ms" % ((time.time() - t) * 1e3), flush=True)
print(q.info, "status counts", np.bincount(q.status[:q.n_loci], minlength=4), "mean iters %.1f max %d" % (q.iters[:q.n_loci].mean(), q.iters[:q.n_loci].max()))
nb = np.diff(np.concatenate([[0]]))  # placeholder
# ---- the first loci against the oracle
from oracle import OracleLib
o = OracleLib()
K = min(150, q.n_loci)
sub = eb.Annotation.__new__(eb.Annotation)
a = q.annot
sub.n_loci = K
i1, s1 = int(a.iso_off[K]), int(a.seg_off[K])
e1 = int(a.exon_off[i1])
sub.iso_off, sub.exon_off, sub.seg_off = a.iso_off[:K + 1].copy(), a.exon_off[:i1 + 1].copy(), a.seg_off[:K + 1].copy()
sub.exon_left, sub.exon_right = a.exon_left[:e1].copy(), a.exon_right[:e1].copy()
sub.seg_left, sub.seg_right = a.seg_left[:s1].copy(), a.seg_right[:s1].copy()
sub.compat_words, sub.key_words = a.compat_words, a.key_words
hh = q.hits.host_hits(K)
r = quantify_host(sub, hh, q.insert, q.read_len, ctx=ctx)
oc, ok = o.exonbin_batch(sub, hh)
print("sample of %d loci, %d hits: compat words equal oracle: %s; hits compatible with some isoform %.3f" % (
    K, hh.n_hits, bool((r["compat"] == oc).all()), float((oc != 0).any(axis=1).mean())))
b = r["bins"]
theta, status, iters = o.em_batch(b.row_off, b.iso_off, b.f_off, b.count, r["F"])
err = np.abs(r["theta"] - theta) / np.maximum(np.abs(theta), 1e-9)
print("EM on the chain's (n, F): status equal %s iters equal %s theta err %.2e" % ((r["status"] == status).all(), (r["iters"] == iters).all(), err.max()))
full = q.theta[:i1]
print("full-run theta of these loci vs sample run: max rel diff %.2e; status equal %s" % (
    float((np.abs(full - r["theta"]) / np.maximum(np.abs(r["theta"]), 1e-9)).max()), (q.status[:K] == r["status"]).all()))
