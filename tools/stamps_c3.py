#!/usr/bin/env python3
"""GPU box, diagnostic build (make -C strawberry_amd/csrc stamps): per-wave cycle
stamps of one EM launch over the wave-kind loci of C3."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from strawberry_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "strawberry_amd", "lib", "libsbgpu_stamps.so")
from strawberry_amd import em, synth

ctx = em.default_context(0)
b = synth.make_c3()
s = em.EmBatchSolver(b, ctx); s.run_em(); r = s.results()
s.set_timing(True)
kinds = s.plan.locus_kinds()
sub = b if os.environ.get("STAMPS_FULL") else b.select(np.nonzero(kinds < 3)[0])
s = em.EmBatchSolver(sub, ctx)
s.set_timing(True)
s.run_em(); torch.cuda.synchronize(); s.run_em(); torch.cuda.synchronize()
print("kernel ms", s.last_kernel_ms())
nw = sum(c["n_waves"] for c in s.plan.classes() if c["kind"] in (0, 1, 2))
buf = np.zeros(65536 * 8, np.uint64)
L = _lib.load()
L.sbgpu_debug_read_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert L.sbgpu_debug_read_stamps(buf.ctypes.data, buf.nbytes) == 0
st = buf.reshape(-1, 8)[:nw].astype(np.float64)
t0 = st[:, 0].min()
def us(x): return x / 100.0   # s_memrealtime ticks at 100 MHz
print("waves", nw)
print("start spread us: min %.1f med %.1f max %.1f" % (us(st[:,0].min()-t0), us(np.median(st[:,0])-t0), us(st[:,0].max()-t0)))
print("lookup us: med %.2f max %.2f" % (us(np.median(st[:,1]-st[:,0])), us((st[:,1]-st[:,0]).max())))
print("first refill done after start us: med %.2f max %.2f" % (us(np.median(st[:,2]-st[:,0])), us((st[:,2]-st[:,0]).max())))
print("end us: med %.1f p90 %.1f max %.1f" % (us(np.median(st[:,3])-t0), us(np.percentile(st[:,3],90)-t0), us(st[:,3].max()-t0)))
print("batches: mean %.2f max %d ; iters mean %.1f max %d" % (st[:,4].mean(), st[:,4].max(), st[:,5].mean(), st[:,5].max()))
key = st[:,6].astype(np.int64)
print("distinct SIMDs seen:", len(np.unique(key)))
life = us(st[:,3]-st[:,0])
upi = life/np.maximum(st[:,5],1)
print("life us: med %.1f max %.1f ; us/iter med %.3f p90 %.3f max %.3f" % (np.median(life), life.max(), np.median(upi), np.percentile(upi,90), upi.max()))
off = 0
print("waves still alive at t (us):", {t: int(((us(st[:,0]-t0) <= t) & (us(st[:,3]-t0) > t)).sum()) for t in (50,100,200,400,600,800,1000,1200,1400,1600,1800)})
print("waves not yet started at t:", {t: int((us(st[:,0]-t0) > t).sum()) for t in (50,100,200,400,600,800,1000)})
long = (st[:,5] >= 900) if st[:,5].max() > 0 else (life >= 500.0)   # no iteration counts in the stamps: by lifetime
print("long waves (>=900 iters or >= 500 us): %d ; their start us: med %.0f p90 %.0f max %.0f ; life med %.0f max %.0f" % (long.sum(), np.median(us(st[long,0]-t0)), np.percentile(us(st[long,0]-t0),90), us(st[long,0]-t0).max(), np.median(us(st[long,3]-st[long,0])), us(st[long,3]-st[long,0]).max()))
lk = key[long]
u, cnt = np.unique(lk, return_counts=True)
print("long waves per SIMD: SIMDs with 1: %d, 2: %d, 3+: %d" % ((cnt==1).sum(), (cnt==2).sum(), (cnt>=3).sum()))
# for every long wave: what fraction of its life does it share its SIMD with another stamped wave?
order = np.argsort(key, kind="stable")
share = []
for i in np.nonzero(long)[0]:
    same = np.nonzero(key == key[i])[0]
    same = same[same != i]
    a0, a1 = st[i,0], st[i,3]
    ov = np.clip(np.minimum(st[same,3], a1) - np.maximum(st[same,0], a0), 0, None).sum()
    share.append(ov / max(a1 - a0, 1))
share = np.array(share)
print("long waves: mean number of co-resident wave-kind waves over their life: med %.2f mean %.2f ; life us by sharing: <0.25: %.0f  0.25-0.75: %.0f  >0.75: %.0f" % (
    np.median(share), share.mean(), np.median(life[long][share<0.25]) if (share<0.25).any() else -1, np.median(life[long][(share>=0.25)&(share<0.75)]) if ((share>=0.25)&(share<0.75)).any() else -1, np.median(life[long][share>=0.75]) if (share>=0.75).any() else -1))
idx = np.nonzero(long)[0]
print("long waves: index in launch order: med %d p90 %d max %d" % (np.median(idx), np.percentile(idx,90), idx.max()))
print("end time of long waves us: med %.0f p90 %.0f max %.0f" % (np.median(us(st[long,3]-t0)), np.percentile(us(st[long,3]-t0),90), us(st[long,3]-t0).max()))
for c in s.plan.classes():
    if c["kind"] not in (0, 1, 2): continue
    n = c["n_waves"]; x = st[off:off+n]; off += n
    lf = us(x[:,3]-x[:,0])
    lw = x[:,5] >= 900
    if lw.any():
        print("  long in C%2d R%2d G%2d: %3d waves, start med %5.0f max %5.0f, life med %5.0f max %5.0f, end max %5.0f" % (c["C"], c["R"], c["G"], lw.sum(), np.median(us(x[lw,0]-t0)), us(x[lw,0]-t0).max(), np.median(lf[lw]), lf[lw].max(), us(x[lw,3]-t0).max()))
    print("  class C%2d R%2d G%2d waves %4d: iters mean %6.1f  wave-us total %8.0f  us/iter med %.3f" % (c["C"], c["R"], c["G"], n, x[:,5].mean(), lf.sum(), np.median(lf/np.maximum(x[:,5],1))))
    continue
    print("  class C%2d R%2d G%2d waves %4d: start med %7.1f  end max %7.1f  iters mean %6.1f max %5d batches max %d  us/iter med %.3f" % (c["C"], c["R"], c["G"], n, us(np.median(x[:,0])-t0), us(x[:,3].max()-t0), x[:,5].mean(), x[:,5].max(), x[:,4].max(), np.median(lf/np.maximum(x[:,5],1))))
