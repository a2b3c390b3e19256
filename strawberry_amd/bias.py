"""BASELINE config 5 ("bias-corrected EM: hexamer + positional bias kernel fused into the E-step"): the build's own bias
model, since the reference has none to match (/root/reference/src/bias.cpp is comments; include/estimate.hpp:235,245,
251-254 are dead hooks; its `-b` option only prints the per-bin sequence columns).

    b_ij = 2 ^ (g_i * t_j),   g_i = clamp(4 (GC_i - 1/2), -1, 1),   t_j in [-1, 1]

GC_i is the GC ratio of bin i's sequence as the reference defines it (Kmer<string>::GCRatio, include/kmer.h:67-77),
measured ON THE DEVICE by the bin-sequence kernel (sbgpu_binseq_device, csrc/binseq_device.h) -- the one sequence
statistic pipeline the reference has; t_j is isoform j's sensitivity to it (a per-isoform constant of the model; drawn
at random for the synthetic workload).  The factor multiplies the weight F_ij inside the EM kernels as they load their
tiles (sbgpu_em_run_device_bias): no biased copy of F exists in memory.  Not a parity path.
"""
import numpy as np

from . import binseq


def make_c5_bias(ctx, batch, seed=0xB1A5, len_lo=60, len_hi=260):
    """Synthetic bin sequences for every row of `batch` (drawn on the device: a GC content per bin, bases i.i.d.), their
    GC ratios from the bin-sequence kernel, and the two factor arrays.
    -> (d_row_bias float64[rows], d_iso_bias float64[isoforms], info dict)"""
    import torch
    dev = torch.device("cuda", ctx.device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    n_rows, n_iso = int(batch.row_off[-1]), int(batch.iso_off[-1])
    length = torch.randint(len_lo, len_hi + 1, (n_rows,), device=dev, generator=g)
    gc_target = torch.rand(n_rows, device=dev, generator=g) * 0.6 + 0.2            # 0.2 .. 0.8
    off = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
    torch.cumsum(length, 0, out=off[1:])
    total = int(off[-1].item())
    row_of = torch.repeat_interleave(torch.arange(n_rows, device=dev), length)
    is_gc = torch.rand(total, device=dev, generator=g) < gc_target[row_of]
    second = torch.rand(total, device=dev, generator=g) < 0.5
    del row_of
    # A C G T = 65 67 71 84
    genome = torch.where(is_gc, torch.where(second, 67, 71), torch.where(second, 65, 84)).to(torch.uint8)
    del is_gc, second
    seg_off = torch.arange(n_rows + 1, dtype=torch.int64, device=dev)                # one segment per bin
    seg_left = (off[:-1] + 1).to(torch.int32)                                        # 1-based closed coordinates
    seg_right = off[1:].to(torch.int32)
    gc, ent, flags, err = binseq.bin_sequence_stats_device(genome, 1, seg_off, seg_left, seg_right, device=ctx.device)
    torch.cuda.synchronize(dev)
    if int(err.item()) != 0:
        raise RuntimeError("the bin-sequence kernel rejected a bin")
    row_bias = torch.clamp(4.0 * (gc - 0.5), -1.0, 1.0)
    iso_bias = torch.rand(n_iso, device=dev, generator=g, dtype=torch.float64) * 2.0 - 1.0
    info = {"bins": n_rows, "bases": total, "gc_mean": float(gc.mean().item()), "entropy_mean": float(ent.mean().item()),
            "model": "b_ij = 2^(g_i t_j), g_i = clamp(4 (GC_i - 0.5), -1, 1) from sbgpu_binseq_device, t_j ~ U(-1, 1)"}
    return row_bias, iso_bias, info


def biased_weights(batch, row_bias, iso_bias):
    """The same factors multiplied into a host copy of F (numpy): what the CPU baseline and the checks solve."""
    F = batch.F.copy()
    rl = np.repeat(np.arange(batch.n_loci), batch.nrow)
    el_row = np.repeat(np.arange(int(batch.row_off[-1])), batch.niso[rl])
    el_start = np.concatenate([[0], np.cumsum(batch.niso[rl])])[:-1]
    el_col = np.arange(len(F)) - np.repeat(el_start, batch.niso[rl])
    el_iso = batch.iso_off[rl[el_row]] + el_col
    return F * np.exp2(row_bias[el_row] * iso_bias[el_iso])
