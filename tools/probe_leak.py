import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import exonbin_util as XU
from strawberry_amd import em, synth, exonbin as eb
from strawberry_amd.quantify import InsertSize, quantify_host, LocusQuantifier
loci = synth.make_gene_models(300, seed=21)
hl, pairs = synth.make_fragments(loci, 150, seed=22)
rows = [(l, eb.hit_features(lb, rb)) for l, (lb, rb) in zip(hl, pairs)]
rows = [(l, f) for l, f in rows if f is not None]
annot, hits = eb.Annotation(loci), eb.Hits([l for l, _ in rows], [f for _, f in rows])
annot, hits = XU.tile(annot, hits, 10)
ctx = em.default_context(0)
frac = eb.Hits.from_arrays(hits.hit_locus, hits.feat_off, hits.feat_code, hits.feat_left, hits.feat_right, np.full(hits.n_hits, 0.5, np.float32))
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
quantify_host(annot, hits, InsertSize(250.0, 30.0), 75, ctx=ctx); quantify_host(annot, frac, None, 75, ctx=ctx)
f0 = free(); ref = None
for i in range(40):
    r = quantify_host(annot, hits if i % 2 == 0 else frac, InsertSize(250.0, 30.0) if i % 3 else None, 75, ctx=ctx)
    if i % 2 == 0 and i % 3:
        if ref is None: ref = r["theta"].copy()
        assert (r["theta"] == ref).all()
f1 = free()
print("free before %.1f MB after %.1f MB (delta %.2f MB) over 40 calls; deterministic theta" % (f0 / 1e6, f1 / 1e6, (f0 - f1) / 1e6))
