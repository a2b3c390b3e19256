"""Debug: which (bin, isoform) weights of the chain sample differ between the device chain and the oracle chain."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_chain_scale_gpu as T
import e2e_util as U
from oracle import OracleLib
from strawberry_amd import chain, em
from strawberry_amd.quantify import InsertSize, quantify_host
o = OracleLib()
ctx = em.default_context(0)
q = chain.ChainQuantifier(ctx, n_loci=T.N_LOCI, n_frags=T.N_FRAGS, seed=41, pin=False)
hits = q.hits.host_hits(q.n_loci); annot = q.annot
r = quantify_host(annot, hits, InsertSize(T.MEAN, T.SD), T.RL, ctx=ctx)
b = r["bins"]; F = r["F"]
f_off, oF, npairs = T.oracle_weights(o, annot, b.row_off, b.bin_key[:, 0], b.bin_compat[:, 0], 16)
nz = oF != 0
err = np.zeros_like(oF); err[nz] = np.abs(F[nz] - oF[nz]) / oF[nz]
bad = np.flatnonzero(err > 1e-12)
print("pairs", npairs, "bad", len(bad), "max", err.max())
ins = o.make_insert(T.MEAN, T.SD)
# the product's own pair descriptions
for e in bad[:12]:
    l = int(np.searchsorted(f_off, e, side="right") - 1)
    niso = int(annot.iso_off[l + 1] - annot.iso_off[l])
    rloc, j = divmod(int(e - f_off[l]), niso)
    rr = int(b.row_off[l]) + rloc
    p = int(np.flatnonzero(b.pair_out_index == e)[0])
    segs = b.pair_seg_lens[b.pair_seg_off[p]:b.pair_seg_off[p + 1]]
    mask = int(b.pair_implicit_mask[p])
    allsegs = [(int(x), int(y)) for x, y in annot.segments(l)]
    coords = [s for k, s in enumerate(allsegs) if (int(b.bin_key[rr, 0]) >> k) & 1]
    jj = int(annot.iso_off[l]) + j
    e0, e1 = int(annot.exon_off[jj]), int(annot.exon_off[jj + 1])
    ex = list(zip(annot.exon_left[e0:e1].tolist(), annot.exon_right[e0:e1].tolist()))
    iso_segs = U.isoform_segments(allsegs, ex)
    got = U.bin_under_iso(coords, iso_segs)
    print("locus", l, "bin", rloc, "iso", j, "F dev %.17g oracle %.17g" % (F[e], oF[e]), "| product segs", segs.tolist(), "mask", bin(mask),
          "iso_len", int(b.pair_iso_len[p]), "| test segs", got, "iso_len", sum(y - x + 1 for x, y in ex))
    imp = [k for k in range(len(segs)) if (mask >> k) & 1]
    print("    oracle on product's description: %.17g" % o.bin_weight(segs, imp, int(b.pair_iso_len[p]), T.RL, ins))
