#!/usr/bin/env python3
"""GPU box: BASELINE config 5 -- the fp32 variant of the EM against the fp64 product path on the C5 batch
(C3 law, 4e8 fragments, bias factors on the weights): how far theta / TPM move, how many loci change status or
iteration count, and what the fp32 kernels buy in time.  Writes a JSON summary (copy it to profiles/).

    python tools/c5_sweep.py [out.json] [n_loci]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from strawberry_amd import em, synth  # noqa: E402


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "c5_sweep.json")
    n_loci = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
    b = synth.make_c5(n_loci=n_loci, total_frags=4e8 * n_loci / 60000)
    ctx = em.default_context(0)
    s = em.EmBatchSolver(b, ctx)
    from strawberry_amd import bias as sbias
    rb, ib, bias_info = sbias.make_c5_bias(ctx, b)        # factors from the bin-sequence kernel, applied at tile load
    s.set_bias(rb, ib)
    total = min(b.n_frags, 2**31 - 1)

    def step64():
        s.run_em()
        s.run_abundance(total_mapped_reads=total, min_isoform_frac=0.0)
        s.run_tpm()

    def step32():
        s.run_em_f32()
        s.theta32_as_f64()
        s.run_abundance(total_mapped_reads=total, min_isoform_frac=0.0)
        s.run_tpm()

    step64()
    r64 = s.results()
    step32()
    r32 = s.results()
    ms64, ms32 = timed(step64), timed(step32)

    th64, th32 = r64["theta"], r32["theta"]
    locus_of = np.repeat(np.arange(b.n_loci), b.niso)
    both_ok = np.isin(r64["status"], (0, 3)) & np.isin(r32["status"], (0, 3))
    m = both_ok[locus_of]
    rel = np.abs(th32 - th64)[m] / np.maximum(np.abs(th64[m]), 1.0)        # per isoform, floor of one fragment
    tpm_abs = np.abs(r32["tpm"] - r64["tpm"])
    tpm_rel = tpm_abs[m] / np.maximum(r64["tpm"][m], 1e-3)
    edges = [0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, np.inf]
    hist = np.histogram(rel, bins=edges)[0]
    tpm_hist = np.histogram(tpm_rel, bins=edges)[0]
    dit = r32["iters"].astype(np.int64) - r64["iters"].astype(np.int64)
    summary = {
        "workload": "C5: C3 law, %d loci, %d fragments, bias factors applied on the device at tile load" % (b.n_loci, b.n_frags),
        "bias": bias_info,
        "ms_per_step": {"f64": ms64, "f32": ms32, "speedup": ms64 / ms32},
        "loci_per_s": {"f64": b.n_loci / ms64 * 1e3, "f32": b.n_loci / ms32 * 1e3},
        "status_f64": np.bincount(r64["status"], minlength=4).tolist(),
        "status_f32": np.bincount(r32["status"], minlength=4).tolist(),
        "loci_status_changed": int((r64["status"] != r32["status"]).sum()),
        "loci_iteration_count_changed": int((dit != 0).sum()),
        "iteration_count_delta_percentiles(50,90,99,100 of |delta|)": np.percentile(np.abs(dit), [50, 90, 99, 100]).tolist(),
        "theta_rel_err (|d theta| / max(theta, 1 fragment), isoforms of loci solved by both)": {
            "bin_edges": [str(e) for e in edges], "counts": hist.tolist(), "max": float(rel.max()), "median": float(np.median(rel)),
            "p99": float(np.percentile(rel, 99))},
        "tpm_rel_err (|d TPM| / max(TPM, 1e-3))": {
            "bin_edges": [str(e) for e in edges], "counts": tpm_hist.tolist(), "max": float(tpm_rel.max()),
            "p99": float(np.percentile(tpm_rel, 99))},
        "tpm_weighted_abs_err (sum |d TPM| / 1e6)": float(tpm_abs.sum() / 1e6),
        "share_of_isoforms_within_1e-4_relative_TPM": float((tpm_rel < 1e-4).mean()),
        "note": "fp32 changes WHEN the absolute test ||next - theta|| < 1e-2 fires: loci near it stop an iteration earlier or "
                "later, and loci whose fp64 run decays a theta_j to exactly 0 (DENOM_ZERO) need not do so in fp32",
    }
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(summary, open(out_path, "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
