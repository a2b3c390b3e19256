import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from strawberry_amd import em
from strawberry_amd.binweight import InsertSize, bin_weights, pack_pairs
ctx = em.default_context(0)
print("ctx ok", flush=True)
for segs, imps, L in [([[300]], [[]], [1000]), ([[300, 300]], [[]], [1000]), ([[100, 50, 80]], [[1]], [1000]),
                      ([[100, 50, 80, 60]], [[]], [1000]), ([[100, 50, 80, 60, 70]], [[2]], [1000]),
                      ([[100, 50, 80, 60, 70, 30, 90]], [[]], [1000])]:
    so, sl, m = pack_pairs(segs, imps)
    t = time.time()
    w = bin_weights(so, sl, m, L, InsertSize(200.0, 20.0), 50, ctx=ctx)
    print(segs, imps, w, "%.3fs" % (time.time() - t), flush=True)
z = np.load(os.path.join(ROOT, "tests", "golden", "binweight_pairs.npz"))
for n in (10, 100, 1500):
    t = time.time()
    w = bin_weights(z["seg_off"][:n + 1], z["seg_lens"], z["implicit_mask"][:n], z["iso_len"][:n], InsertSize(230.0, 35.0), 75, ctx=ctx)
    ref = z["w_gauss"][:n]
    nz = ref != 0
    print(n, "pairs %.3fs" % (time.time() - t), "max rel err", (np.abs(w[nz] - ref[nz]) / np.abs(ref[nz])).max() if nz.any() else 0, flush=True)
