"""Where does a bench step's wall time go?  (diagnostic, GPU box)"""
import sys, time
sys.path.insert(0, '.')
import torch
from strawberry_amd import em, synth, dist as sdist
b = synth.make_c3()
ctx = em.Context(0)
s = em.EmBatchSolver(b, ctx)
q = sdist.ShardQuantifier(s, 200000000, min_isoform_frac=0.0)
def timed(fn, n=20, label=""):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-28s n=%3d submit %.3f  wall %.3f  gpu-event %.3f ms/step" % (label, n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, e0.elapsed_time(e1) / n))
timed(s.run_em, 20, "run_em")
timed(lambda: (s.run_em(), s.run_abundance(**q.kw)), 20, "run_em+abundance")
timed(lambda: (s.run_em(), s.run_abundance(**q.kw), s.run_tpm(s.d_sum_fpkm)), 20, "run_em+abundance+tpm")
timed(q.step, 20, "quant.step")
timed(q.step, 100, "quant.step")
timed(lambda: s.run_abundance(**q.kw), 20, "abundance only")
timed(lambda: s.run_tpm(s.d_sum_fpkm), 20, "tpm only")
