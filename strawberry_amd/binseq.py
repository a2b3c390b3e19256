"""Per-bin sequence statistics of the `-f` table.

Reference: with `-b genome.fa` Sample::printContext appends, for every exon bin, the GC ratio, the
hexamer entropy and four "high GC stretch" flags of the bin's sequence
(/root/reference/src/alignments.cpp:1622-1636, include/kmer.h:14-135, include/isoform.h:173-182).
That is all the reference's bias option computes (src/bias.cpp holds no code).  The arithmetic runs in
the HIP kernel behind sbgpu_binseq_host / sbgpu_binseq_device; nothing here computes results in Python.
"""
import numpy as np

from . import _lib
from .em import default_context


def read_fasta(path):
    """{name: bytes} of a FASTA file, sequence bytes as they stand (either case, N, ...)."""
    out, name, parts = {}, None, []
    with open(path, "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if name is not None:
                    out[name] = b"".join(parts)
                name, parts = line[1:].split()[0].decode(), []
            else:
                parts.append(line.strip())
    if name is not None:
        out[name] = b"".join(parts)
    return out


def bin_segments(bins):
    """(seg_off, seg_left, seg_right) of every bin of an exonbin.LocusBins, bins in batch order: the set bits of
    the bins' key words, as coordinates of the loci's disjoint segments (ascending, like ExonBin::_coords)."""
    annot = bins._annot
    key = np.ascontiguousarray(bins.bin_key, np.uint32).reshape(bins.n_bins, -1)
    bits = np.unpackbits(key.view(np.uint8), axis=1, bitorder="little")      # [n_bins, 32 * key_words]
    b, k = np.nonzero(bits)                                                  # by bin, then by segment
    locus = np.searchsorted(np.asarray(bins.row_off), b, side="right") - 1
    seg = np.asarray(annot.seg_off)[locus] + k
    off = np.concatenate([[0], np.cumsum(bits.sum(axis=1, dtype=np.int64))]).astype(np.int64)
    return off, np.asarray(annot.seg_left)[seg].astype(np.uint32), np.asarray(annot.seg_right)[seg].astype(np.uint32)


def bin_sequence_stats(genome, seg_off, seg_left, seg_right, genome_start=1, device=0):
    """(gc[n], entropy[n], flags[n]) for bins given as CSR segment lists; genome[0] is base
    `genome_start` (1-based) of the chromosome.  Host-buffer form of the C ABI."""
    L = _lib.load()
    ctx = default_context(device)
    g = np.frombuffer(bytes(genome), np.uint8)
    seg_off = np.ascontiguousarray(seg_off, np.int64)
    seg_left = np.ascontiguousarray(seg_left, np.uint32)
    seg_right = np.ascontiguousarray(seg_right, np.uint32)
    n = len(seg_off) - 1
    gc, ent, fl = np.zeros(n), np.zeros(n), np.zeros(n, np.uint8)
    if n:
        _lib.check(L.sbgpu_binseq_host(ctx.h, g.ctypes.data if g.size else None, int(genome_start), g.size, n,
                                       seg_off.ctypes.data, seg_left.ctypes.data if seg_left.size else None,
                                       seg_right.ctypes.data if seg_right.size else None, gc.ctypes.data,
                                       ent.ctypes.data, fl.ctypes.data), "sbgpu_binseq_host")
    return gc, ent, fl


def bin_sequence_stats_device(d_genome, genome_start, d_seg_off, d_seg_left, d_seg_right, device=0, stream=None):
    """Device-resident form over torch tensors (uint8 genome, int64 offsets, int32-viewed uint32
    coordinates): returns (gc, entropy, flags) tensors on the device; raises if the kernel rejected a bin."""
    import torch
    L = _lib.load()
    ctx = default_context(device)
    n = d_seg_off.numel() - 1
    dev = d_genome.device
    gc = torch.empty(n, dtype=torch.float64, device=dev)
    ent = torch.empty(n, dtype=torch.float64, device=dev)
    fl = torch.empty(n, dtype=torch.uint8, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    s = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
    _lib.check(L.sbgpu_binseq_device(ctx.h, d_genome.data_ptr(), int(genome_start), d_genome.numel(), n, d_seg_off.data_ptr(),
                                     d_seg_left.data_ptr(), d_seg_right.data_ptr(), gc.data_ptr(), ent.data_ptr(),
                                     fl.data_ptr(), err.data_ptr(), s), "sbgpu_binseq_device")
    return gc, ent, fl, err
