#!/usr/bin/env python3
"""GPU box: wave-kind loci of C3 alone, with and without phased execution."""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from strawberry_amd import em, synth
    ctx = em.default_context(0)
    b = synth.make_c3()
    s = em.EmBatchSolver(b, ctx)
    s.set_timing(True)
    kinds = s.plan.locus_kinds()
    for name, sel in (("wave-kind only", kinds < 3), ("block-kind only", kinds >= 3), ("all", kinds >= 0)):
        sub = b.select(np.nonzero(sel)[0])
        s2 = em.EmBatchSolver(sub, ctx)
        s2.set_timing(True)
        s2.run_em(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); s2.run_em(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print("   %-16s %6d loci  %.3f ms   kernels %s" % (name, sub.n_loci, best, [round(x, 3) for x in s2.last_kernel_ms()]), flush=True)
else:
    for ph in ("", "64,256", "32,128,400", "100,400", "200"):
        for rm in ("1", "2"):
            print("PHASES=%r RMULT=%s" % (ph, rm), flush=True)
            env = dict(os.environ, SBGPU_PHASES=ph, SBGPU_WAVE_RMULT=rm)
            subprocess.run([sys.executable, __file__, "child"], env=env, stderr=subprocess.DEVNULL)
