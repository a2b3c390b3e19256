"""Fragments -> abundances for a batch of loci: the device-resident chain behind
LocusContext's constructor + estimate_abundances (/root/reference/include/estimate.hpp:60-103,
src/estimate.cpp:279-355) as Sample::quantifyCluster drives them (src/alignments.cpp:1505-1546).

    hits --exonbin kernel--> compat/key words --host bookkeeping--> bins, counts, (bin, isoform) pairs
         --bin-weight kernel--> F (written straight into the EM batch) --EM kernels--> theta
         --epilogue kernels--> FPKM / Frac / TPM

torch holds the device buffers; every number is produced by libsbgpu.so.
"""
import ctypes as C

import numpy as np

from . import _lib
from .binweight import InsertSize  # noqa: F401  (re-exported: the caller builds one)
from .em import EmBatchSolver, _torch, default_context
from .exonbin import LocusBins
from .synth import LocusBatch


class _LazyHitBin:
    """hit -> bin of the device grouping: stays in HBM until somebody asks for it as an array."""

    def __init__(self, d):
        self._d, self._h = d, None

    def __array__(self, dtype=None, copy=None):
        if self._h is None:
            self._h = self._d.cpu().numpy()
        return self._h if dtype is None else self._h.astype(dtype)

    def __getitem__(self, k):
        return np.asarray(self)[k]


class LocusQuantifier:
    def __init__(self, annot, hits, insert, read_len, long_read=False, ctx=None, device=0, device_bins=True):
        """device_bins: group the hits into bins on the GPU when the input allows it (sorted hits, whole-number
        masses; sbgpu_bins_create_device), else -- or when False -- on the host."""
        self.torch = torch = _torch()
        self.ctx = ctx or default_context(device)
        self.dev = torch.device("cuda", self.ctx.device)
        self.annot, self.hits = annot, hits
        self.insert, self.read_len, self.long_read = insert, int(read_len), bool(long_read)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)  # noqa: E731
        # device copies of the kernel inputs (uint32 travels as int32 bit patterns)
        self._d = {k: up(getattr(annot, k).view(np.int32) if getattr(annot, k).dtype == np.uint32 else getattr(annot, k))
                   for k in ("iso_off", "exon_off", "exon_left", "exon_right", "seg_off", "seg_left", "seg_right")}
        self._d.update({k: up(getattr(hits, k).view(np.int32) if getattr(hits, k).dtype == np.uint32 else getattr(hits, k))
                        for k in ("hit_locus", "feat_off", "feat_code", "feat_left", "feat_right")})
        self._d["mass"] = up(hits.mass)
        self.device_bins = bool(device_bins)
        self.bins_on_device = False
        self.bins = None
        self.solver = None

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def _p(self, name):
        t = self._d[name]
        return t.data_ptr() if t.numel() else None

    def assign_bins(self):
        """A5: kernel for the interval tests, then the host bookkeeping.  -> LocusBins"""
        torch, a, h = self.torch, self.annot, self.hits
        cw, kw = a.compat_words, a.key_words
        self.d_compat = torch.zeros((max(h.n_hits, 1), cw), dtype=torch.int32, device=self.dev)
        self.d_key = torch.zeros((max(h.n_hits, 1), kw), dtype=torch.int32, device=self.dev)
        an = _lib.sbgpu_annotation_t(a.n_loci, self._p("iso_off"), self._p("exon_off"), self._p("exon_left"),
                                     self._p("exon_right"), self._p("seg_off"), self._p("seg_left"), self._p("seg_right"))
        ht = _lib.sbgpu_hits_t(h.n_hits, self._p("hit_locus"), self._p("feat_off"), self._p("feat_code"),
                               self._p("feat_left"), self._p("feat_right"))
        _lib.check(self.ctx.L.sbgpu_exonbin_device(self.ctx.h, C.byref(an), C.byref(ht), cw, kw, self.d_compat.data_ptr(),
                                                   self.d_key.data_ptr(), self._stream()), "sbgpu_exonbin_device")
        self.bins = None
        if self.device_bins and h.n_hits:
            self.d_hit_bin = torch.zeros(h.n_hits, dtype=torch.int64, device=self.dev)
            self.bins = LocusBins.on_device(self.ctx, a, h, ht, self._p("mass"), cw, kw, self.d_compat.data_ptr(),
                                            self.d_key.data_ptr(), self.d_hit_bin.data_ptr(), self._stream())
            if self.bins is not None:
                self.bins.hit_bin = _LazyHitBin(self.d_hit_bin)
        self.bins_on_device = self.bins is not None
        if self.bins is None:   # host form: bring the words back
            compat = self.d_compat[:h.n_hits].cpu().numpy().view(np.uint32)
            key = self.d_key[:h.n_hits].cpu().numpy().view(np.uint32)
            self.bins = LocusBins(a, h, compat, key)
        return self.bins

    def bin_weights(self):
        """A4: one weight per (bin, isoform) pair, scattered into the EM batch's F on the device."""
        torch, b = self.torch, self.bins
        up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(self.dev)  # noqa: E731
        self.d_F = torch.zeros(max(b.n_elem, 1), dtype=torch.float64, device=self.dev)
        if b.n_pairs == 0:
            return self.d_F
        if self.long_read:      # F = 1/L_j: neither the segments nor the pdf are looked at
            pdf = np.zeros(2, np.float64)
        else:
            span = np.diff(np.concatenate([[0], np.cumsum(b.pair_seg_lens.astype(np.int64))])[b.pair_seg_off])
            pdf = self.insert.pdf_table(int(span.max()) + 1, self.read_len)
        d_off, d_seg = up(b.pair_seg_off), up(b.pair_seg_lens.view(np.int32))
        d_mask, d_len = up(b.pair_implicit_mask.view(np.int32)), up(b.pair_iso_len)
        d_idx, d_pdf = up(b.pair_out_index), up(pdf)
        lmin_base = self.insert.start_offset if self.insert.use_emp else self.read_len
        _lib.check(self.ctx.L.sbgpu_binweight_device(
            self.ctx.h, b.n_pairs, d_off.data_ptr(), d_seg.data_ptr(), d_mask.data_ptr(), d_len.data_ptr(),
            d_idx.data_ptr(), d_pdf.data_ptr(), len(pdf), self.read_len, lmin_base, int(self.long_read),
            self.d_F.data_ptr(), self._stream()), "sbgpu_binweight_device")
        self.torch.cuda.current_stream(self.dev).synchronize()  # the inputs above go out of scope
        return self.d_F

    def solve(self, total_mapped_reads, **abundance_kw):
        """A1-A3, A7: EM + abundance epilogue + TPM over all loci.  -> results dict (host arrays)."""
        b = self.bins
        batch = LocusBatch(b.row_off, b.iso_off, b.f_off, b.count, None, b.iso_len, "hits")
        self.solver = s = EmBatchSolver(batch, self.ctx, d_F=self.d_F)
        s.run_em()
        s.run_abundance(total_mapped_reads, **abundance_kw)
        s.run_tpm()
        return s.results()

    def run(self, total_mapped_reads, **abundance_kw):
        self.assign_bins()
        self.bin_weights()
        return self.solve(total_mapped_reads, **abundance_kw)


def quantify_host(annot, hits, insert, read_len, long_read=False, ctx=None, device=0):
    """sbgpu_quantify_host: the whole chain as one C-ABI call on host arrays (what a C / C++ driver uses).
    insert=None: build the empirical insert-size distribution from the hits.
    -> dict(theta, status, iters, compat, bins (LocusBins incl. hit_bin), F, insert (mean, sd, use_emp, ...))"""
    from .exonbin import LocusBins
    ctx = ctx or default_context(device)
    L = ctx.L
    a, h = annot._struct(), hits._struct()
    n_iso = int(annot.iso_off[-1])
    theta = np.zeros(n_iso + 1, np.float64)
    status = np.zeros(annot.n_loci + 1, np.int32)
    iters = np.zeros(annot.n_loci + 1, np.int32)
    cw, kw = annot.compat_words, annot.key_words
    compat = np.zeros((max(hits.n_hits, 1), cw), np.uint32)
    used = _lib.sbgpu_insert_t()
    ins = insert._struct(read_len, long_read) if insert is not None else None
    handle = C.c_void_p()
    _lib.check(L.sbgpu_quantify_host(ctx.h, C.byref(a), C.byref(h), hits.mass.ctypes.data if hits.n_hits else None,
                                     C.byref(ins) if ins is not None else None, int(read_len), int(long_read),
                                     theta.ctypes.data, status.ctypes.data, iters.ctypes.data, compat.ctypes.data,
                                     C.byref(used), C.byref(handle)), "sbgpu_quantify_host")
    info = (C.c_int64 * 8)()
    _lib.check(L.sbgpu_bins_info(handle, info), "sbgpu_bins_info")
    F = np.zeros(max(int(info[3]), 1), np.float64)
    _lib.check(L.sbgpu_bins_export_weights(handle, F.ctypes.data), "sbgpu_bins_export_weights")
    emp = None
    if used.use_emp:
        n = used.end_offset - used.start_offset + 1
        emp = np.ctypeslib.as_array(used.emp_hist, shape=(n,)).copy()
    bins = LocusBins.__new__(LocusBins)
    bins._export(L, annot, handle, hits.n_hits, cw, kw, with_hit_bin=True)   # destroys the handle
    return {"theta": theta[:n_iso], "status": status[:annot.n_loci], "iters": iters[:annot.n_loci],
            "compat": compat[:hits.n_hits], "bins": bins, "F": F[:int(info[3])],
            "insert": {"mean": used.mean, "sd": used.sd, "use_emp": bool(used.use_emp), "start_offset": used.start_offset,
                       "end_offset": used.end_offset, "total_reads": used.total_reads, "emp_hist": emp}}


def quantify_resident(annot, hits, insert, read_len, mapped_reads, long_read=False, ctx=None, device=0, comm=None,
                      min_isoform_frac=0.0, filter_by_expression=True, effective_len_norm=False):
    """sbgpu_quantify_resident on host hits brought to the device first (torch owns the copies): pass 1 (insert=None: the
    empirical insert-size law, built on the device), bins, weights, EM, FPKM / Frac / keep, the FPKM all-reduce over `comm`
    (dist.AbiComm / dist.HostComm; None: a world of one), TPM.  The hits must come grouped by locus.
    -> dict(theta, fpkm, frac, tpm, keep, status, iters, insert, total_fpkm, total_mapped_reads, n_frag_lens, info)"""
    import torch
    ctx = ctx or default_context(device)
    L = ctx.L
    dev = torch.device("cuda", ctx.device)
    if hits.n_hits > 1 and (np.diff(hits.hit_locus) < 0).any():
        raise ValueError("quantify_resident: hits must be grouped by locus")
    off = np.concatenate([[0], np.cumsum(np.bincount(hits.hit_locus, minlength=annot.n_loci))]).astype(np.int64)
    up = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x).view(dt)).to(dev)  # noqa: E731
    d = {"hit_locus": up(hits.hit_locus, np.int32), "feat_off": up(hits.feat_off, np.int64), "feat_code": up(hits.feat_code, np.uint8),
         "feat_left": up(hits.feat_left, np.int32), "feat_right": up(hits.feat_right, np.int32)}
    d_mass = up(hits.mass, np.float32)
    hs = _lib.sbgpu_hits_t()
    hs.n_hits = hits.n_hits
    for k, v in d.items():
        setattr(hs, k, v.data_ptr())
    a = annot._struct()
    n_iso, nl = int(annot.iso_off[-1]), annot.n_loci
    res = {k: np.zeros(n_iso + 1, np.float64) for k in ("theta", "fpkm", "frac", "tpm")}
    res["keep"] = np.zeros(n_iso + 1, np.int32)
    res["status"], res["iters"] = np.zeros(nl + 1, np.int32), np.zeros(nl + 1, np.int32)
    out = _lib.sbgpu_abundances_t()
    for k, v in res.items():
        setattr(out, k, v.ctypes.data)
    par = _lib.sbgpu_abundance_params_t(0, int(effective_len_norm), int(filter_by_expression), 0, 0.0, float(min_isoform_frac))
    used = _lib.sbgpu_insert_t()
    ins = insert._struct(read_len, long_read) if insert is not None else None
    handle = C.c_void_p()
    torch.cuda.synchronize(dev)
    _lib.check(L.sbgpu_quantify_resident(ctx.h, C.byref(a), C.byref(hs), d_mass.data_ptr(), off.ctypes.data,
                                         C.byref(ins) if ins is not None else None, int(read_len), int(long_read), int(mapped_reads),
                                         C.byref(par), comm.h if comm is not None else None, C.byref(used), C.byref(out),
                                         C.byref(handle)), "sbgpu_quantify_resident")
    emp = None
    if used.use_emp:
        emp = np.ctypeslib.as_array(used.emp_hist, shape=(used.end_offset - used.start_offset + 1,)).copy()
    info = (C.c_int64 * 8)()
    _lib.check(L.sbgpu_bins_info(handle, info), "sbgpu_bins_info")
    L.sbgpu_bins_destroy(handle)
    r = {k: (v[:n_iso] if k not in ("status", "iters") else v[:nl]) for k, v in res.items()}
    r.update({"insert": {"mean": used.mean, "sd": used.sd, "use_emp": bool(used.use_emp), "start_offset": used.start_offset,
                         "end_offset": used.end_offset, "total_reads": used.total_reads, "emp_hist": emp},
              "total_fpkm": float(out.total_fpkm), "total_mapped_reads": int(out.total_mapped_reads), "n_frag_lens": int(out.n_frag_lens),
              "info": {"n_bins": int(info[2]), "n_elem": int(info[3]), "n_pairs": int(info[4])}})
    return r
