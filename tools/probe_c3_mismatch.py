"""GPU box: solve C3 and list the loci whose status or iteration count differs from the oracle (diagnostic)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from strawberry_amd import em, synth
from oracle.lib import OracleLib
ctx = em.default_context(0)
b = synth.make_c3()
s = em.EmBatchSolver(b, ctx); s.run_em(); r = s.results()
o = OracleLib()
ot, os_, oi = o.em_batch(b.row_off, b.iso_off, b.f_off, b.count, b.F, threads=8)
bad = np.nonzero((r["iters"] != oi) | (r["status"] != os_))[0]
print("mismatches", len(bad))
kinds = s.plan.locus_kinds()
nrow = np.diff(b.row_off); niso = np.diff(b.iso_off)
for l in bad[:40]:
    th = r["theta"][b.iso_off[l]:b.iso_off[l+1]]; to = ot[b.iso_off[l]:b.iso_off[l+1]]
    print("locus %6d kind %d nrow %4d niso %3d  iters gpu %4d oracle %4d status %d/%d  theta relerr %.2e" % (l, kinds[l], nrow[l], niso[l], r["iters"][l], oi[l], r["status"][l], os_[l], np.abs(th-to).max()/max(np.abs(to).max(),1e-9)))
