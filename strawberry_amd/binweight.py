"""Host-side mirror of the reference's bin-weight call surface.

Reference: LocusContext::set_theory_bin_weight, /root/reference/src/estimate.cpp:201-234
(one weight per (exon bin, isoform) pair); InsertSize, include/read.hpp:176-192 and
src/read.cpp:228-297.  All arithmetic runs in the HIP kernel behind
sbgpu_binweight_host / sbgpu_binweight_device; nothing here falls back to the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from .em import default_context


class InsertSize:
    """InsertSize(mean, sd) or InsertSize.from_frag_lens(lens) like the reference's
    three constructors (src/read.cpp:228-272)."""

    def __init__(self, mean=200.0, sd=80.0):
        self.mean, self.sd = float(mean), float(sd)
        self.use_emp = False
        self.start_offset = self.end_offset = self.total_reads = 0
        self.emp_hist = np.zeros(1, np.float64)

    @classmethod
    def from_frag_lens(cls, frag_lens):
        fl = np.asarray(frag_lens, np.int64)
        if len(fl) < 1:
            raise ValueError("Not enough reads")
        self = cls()
        # mean_and_sd_insert_size, src/read.cpp:14-20
        self.mean = float(fl.astype(np.float64).sum() / len(fl))
        self.sd = float(np.sqrt((fl.astype(np.float64) ** 2).sum() / len(fl) - self.mean * self.mean))
        self.use_emp = True
        self.start_offset, self.end_offset, self.total_reads = int(fl.min()), int(fl.max()), len(fl)
        self.emp_hist = np.bincount(fl - fl.min(), minlength=self.end_offset - self.start_offset + 1).astype(np.float64)
        return self

    @classmethod
    def from_hist(cls, start_offset, hist):
        """The same law from its histogram alone (hist[k] = number of fragments of length start_offset + k, first and last
        entries non-zero): what sbgpu_quantify_* builds from the device's pass 1 -- integer sums, one rounding each."""
        h = np.asarray(hist)
        n = int(h.sum())
        if n < 1:
            raise ValueError("Not enough reads")
        self = cls()
        lens = [int(start_offset) + k for k in range(len(h))]
        tot = sum(int(c) * l for c, l in zip(h, lens))
        sq = sum(int(c) * l * l for c, l in zip(h, lens))
        self.mean = float(tot) / float(n)
        self.sd = float(np.sqrt(float(sq) / float(n) - self.mean * self.mean))
        self.use_emp = True
        self.start_offset, self.end_offset, self.total_reads = int(start_offset), int(start_offset) + len(h) - 1, n
        self.emp_hist = np.ascontiguousarray(h, np.float64)
        return self

    def _struct(self, read_len, long_read=False):
        s = _lib.sbgpu_insert_t()
        s.mean, s.sd, s.use_emp = self.mean, self.sd, int(self.use_emp)
        s.start_offset, s.end_offset, s.total_reads = self.start_offset, self.end_offset, self.total_reads
        s.emp_hist = self.emp_hist.ctypes.data_as(C.POINTER(C.c_double))
        s.read_len, s.long_read = int(read_len), int(long_read)
        return s

    def pdf_table(self, n, read_len=0):
        """pdf[fl] = InsertSize::emp_dist_pdf(fl), fl in [0, n)."""
        L = _lib.load()
        out = np.zeros(n, np.float64)
        s = self._struct(read_len)
        _lib.check(L.sbgpu_insert_pdf_table(C.byref(s), n, out.ctypes.data), "sbgpu_insert_pdf_table")
        return out


def pack_pairs(seg_lens_list, implicit_idx_list):
    """(seg_off int64, seg_lens uint32, implicit_mask uint32) from per-pair python lists."""
    n = len(seg_lens_list)
    seg_off = np.zeros(n + 1, np.int64)
    for i, s in enumerate(seg_lens_list):
        seg_off[i + 1] = seg_off[i] + len(s)
    seg = np.zeros(int(seg_off[-1]), np.uint32)
    mask = np.zeros(n, np.uint32)
    for i, (s, imp) in enumerate(zip(seg_lens_list, implicit_idx_list)):
        seg[seg_off[i]:seg_off[i + 1]] = np.asarray(s, np.uint32)
        m = 0
        for k in imp:
            m |= 1 << int(k)
        mask[i] = m
    return seg_off, seg, mask


def bin_weights(seg_off, seg_lens, implicit_mask, iso_len, insert, read_len, long_read=False, ctx=None, device=0):
    """One weight per (bin, isoform) pair through sbgpu_binweight_host."""
    ctx = ctx or default_context(device)
    seg_off = np.ascontiguousarray(seg_off, np.int64)
    seg_lens = np.ascontiguousarray(seg_lens, np.uint32)
    implicit_mask = np.ascontiguousarray(implicit_mask, np.uint32)
    iso_len = np.ascontiguousarray(iso_len, np.int32)
    n = len(seg_off) - 1
    out = np.zeros(max(n, 1), np.float64)
    s = insert._struct(read_len, long_read)
    _lib.check(ctx.L.sbgpu_binweight_host(ctx.h, n, seg_off.ctypes.data, seg_lens.ctypes.data if seg_lens.size else None,
                                          implicit_mask.ctypes.data, iso_len.ctypes.data, C.byref(s),
                                          out.ctypes.data), "sbgpu_binweight_host")
    return out[:n]
