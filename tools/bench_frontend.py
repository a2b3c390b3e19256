#!/usr/bin/env python3
"""GPU box: the front of the path on records that are RESIDENT in HBM -- alignment records -> read pairs
(sbgpu_pair_mates_device) -> unique hits (sbgpu_collapse_pairs_device) -- on synthetic clusters whose sizes follow a
log-normal law (so a few clusters are far beyond the LDS kernels' 8192 records / 4096 pairs and take the global-memory
kernels).  Records are made with torch on the device: per cluster, fragments at random starts with duplicates, mates 75
bases long and an insert of 150-330; records in (cluster, position) order, as a BAM gives them.  Prints one JSON line.
usage: bench_frontend.py [n_clusters=20000] [n_pairs=1e7] [sigma=1.3]"""
import ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from strawberry_amd import _lib, em
n_cl = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000
n_pairs = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
sigma = float(sys.argv[3]) if len(sys.argv) > 3 else 1.3
ctx = em.default_context(0)
L, dev = ctx.L, torch.device("cuda", ctx.device)
g = torch.Generator(device=dev); g.manual_seed(5)
# pairs per cluster: log-normal shares of n_pairs
share = torch.exp(sigma * torch.randn(n_cl, generator=g, device=dev))
per = torch.clamp((share / share.sum() * n_pairs).round().long(), min=1)
pair_cl = torch.repeat_interleave(torch.arange(n_cl, device=dev), per)
P = int(pair_cl.numel())
base = (pair_cl + 1) * 4_000_000
span = torch.clamp(per[pair_cl] // 4, min=400)                      # ~4 copies per start: duplicates to collapse
start = base + (torch.rand(P, generator=g, device=dev) * span).long()
ins = 150 + (torch.rand(P, generator=g, device=dev) * 4).long() * 60
lpos, rpos = start, start + ins
# two records per pair, in (cluster, position) order (stable: the left mate first where both start together)
rec_pos = torch.stack([lpos, rpos], 1).reshape(-1)
rec_pair = torch.arange(P, device=dev).repeat_interleave(2)
rec_is_right = torch.tensor([0, 1], device=dev).repeat(P)
key = rec_pos * 2 + rec_is_right                                     # (the cluster is in the position's high part)
order = torch.argsort(key, stable=True)
rec_pos, rec_pair, rec_is_right = rec_pos[order], rec_pair[order], rec_is_right[order]
n_rec = 2 * P
read_id = (rec_pair + 1).to(torch.int64)
block_off = torch.arange(n_rec + 1, device=dev, dtype=torch.int64)
block_left = rec_pos.to(torch.int32)
block_right = (rec_pos + 74).to(torch.int32)
partner = torch.where(rec_is_right == 1, lpos[rec_pair], rpos[rec_pair]).to(torch.int32)
flags = (rec_is_right | (1 << 2)).to(torch.uint8)                    # bit 0: reverse strand; XS +
nh = torch.ones(n_rec, dtype=torch.int32, device=dev)
read_off = torch.zeros(n_cl + 1, dtype=torch.int64)
read_off[1:] = torch.cumsum(2 * per, 0).cpu()
read_off_np = read_off.numpy()
torch.cuda.synchronize()
rs = _lib.sbgpu_reads_t(n_rec, read_id.data_ptr(), block_off.data_ptr(), block_left.data_ptr(), block_right.data_ptr(),
                        partner.data_ptr(), flags.data_ptr(), nh.data_ptr())
stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
big_pair = int((2 * per > 8192).sum()); big_col = int((per > 4096).sum())
times = {"pair_mates": [], "collapse": []}
info_p, info_u = (C.c_int64 * 8)(), (C.c_int64 * 8)()
for it in range(9):
    hp, hu = C.c_void_p(), C.c_void_p()
    torch.cuda.synchronize(); t = time.time()
    _lib.check(L.sbgpu_pair_mates_device(ctx.h, n_cl, C.byref(rs), read_off_np.ctypes.data, stream, C.byref(hp)), "sbgpu_pair_mates_device")
    torch.cuda.synchronize(); t1 = time.time()
    pairs, poff = _lib.sbgpu_pairs_t(), C.c_void_p()
    _lib.check(L.sbgpu_matepairs_pairs(hp, C.byref(pairs), C.byref(poff)), "sbgpu_matepairs_pairs")
    _lib.check(L.sbgpu_matepairs_info(hp, info_p), "sbgpu_matepairs_info")
    torch.cuda.synchronize(); t2 = time.time()
    _lib.check(L.sbgpu_collapse_pairs_device(ctx.h, n_cl, C.byref(pairs), poff, stream, C.byref(hu)), "sbgpu_collapse_pairs_device")
    torch.cuda.synchronize(); t3 = time.time()
    _lib.check(L.sbgpu_uniq_dev_info(hu, info_u), "sbgpu_uniq_dev_info")
    L.sbgpu_uniq_dev_destroy(hu); L.sbgpu_matepairs_destroy(hp)
    if it >= 2:
        times["pair_mates"].append((t1 - t) * 1e3); times["collapse"].append((t3 - t2) * 1e3)
pm, co = float(np.median(times["pair_mates"])), float(np.median(times["collapse"]))
print(json.dumps({"metric": "front end on resident records: records -> pairs -> unique hits", "clusters": n_cl, "records": n_rec,
                  "pairs": int(info_p[0]), "complete_pairs": int(info_p[1]), "unique_hits": int(info_u[0]),
                  "largest_cluster_records": int(2 * per.max()), "clusters_beyond_8192_records": big_pair, "clusters_beyond_4096_pairs": big_col,
                  "pair_mates_ms": pm, "collapse_ms": co, "records_per_s": n_rec / (pm * 1e-3), "pairs_per_s_collapse": int(info_p[0]) / (co * 1e-3),
                  "pair_mates_ms_min": float(np.min(times["pair_mates"])), "collapse_ms_min": float(np.min(times["collapse"])),
                  "note": "whole calls incl. their host synchronisations and scratch allocation; median (and minimum) of 7 after two warm-up rounds"}))
