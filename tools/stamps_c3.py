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
sub = b.select(np.nonzero(kinds < 3)[0])
s = em.EmBatchSolver(sub, ctx)
s.set_timing(True)
s.run_em(); torch.cuda.synchronize(); s.run_em(); torch.cuda.synchronize()
print("kernel ms", s.last_kernel_ms())
nw = sum(c["n_waves"] for c in s.plan.classes())
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
print("refill us total per wave: med %.2f max %.2f ; events us: med %.2f max %.2f" % (us(np.median(st[:,6])), us(st[:,6].max()), us(np.median(st[:,7])), us(st[:,7].max())))
life = us(st[:,3]-st[:,0])
upi = life/np.maximum(st[:,5],1)
print("life us: med %.1f max %.1f ; us/iter med %.3f p90 %.3f max %.3f" % (np.median(life), life.max(), np.median(upi), np.percentile(upi,90), upi.max()))
off = 0
print("waves still alive at t (us):", {t: int(((us(st[:,0]-t0) <= t) & (us(st[:,3]-t0) > t)).sum()) for t in (50,100,200,400,600,800,1000,1200,1400,1600,1800)})
print("waves not yet started at t:", {t: int((us(st[:,0]-t0) > t).sum()) for t in (50,100,200,400,600,800,1000)})
long = (st[:,5] >= 900) if st[:,5].max() > 0 else (life >= 500.0)   # no iteration counts in the stamps: by lifetime
print("long waves (>=900 iters or >= 500 us): %d ; their start us: med %.0f p90 %.0f max %.0f ; life med %.0f max %.0f" % (long.sum(), np.median(us(st[long,0]-t0)), np.percentile(us(st[long,0]-t0),90), us(st[long,0]-t0).max(), np.median(us(st[long,3]-st[long,0])), us(st[long,3]-st[long,0]).max()))
for c in s.plan.classes():
    n = c["n_waves"]; x = st[off:off+n]; off += n
    lf = us(x[:,3]-x[:,0])
    print("  class C%2d R%2d G%2d waves %4d: start med %7.1f  end max %7.1f  iters mean %6.1f max %5d batches max %d  us/iter med %.3f" % (c["C"], c["R"], c["G"], n, us(np.median(x[:,0])-t0), us(x[:,3].max()-t0), x[:,5].mean(), x[:,5].max(), x[:,4].max(), np.median(lf/np.maximum(x[:,5],1))))
