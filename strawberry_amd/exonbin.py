"""Host-side mirror of the reference's exon-bin assignment.

Reference: LocusContext's constructor and assign_exon_bin, /root/reference/include/estimate.hpp:60-103
and src/estimate.cpp:135-198.  The interval tests (Contig::is_compatible, overlap_exons) run in
the HIP kernel behind sbgpu_exonbin_host / sbgpu_exonbin_device; the map/set bookkeeping around
them is the C++ host code of libsbgpu.so (csrc/locus_bins.cpp).  Nothing here computes results in
Python and nothing falls back to the CPU for the kernel's part.
"""
import ctypes as C

import numpy as np

from . import _lib
from .em import default_context

MATCH, INTRON, GAP = 0, 1, 2


def _ptr(a):
    return a.ctypes.data if a.size else None


class Annotation:
    """Loci -> isoforms -> exons (closed, sorted) and the loci's disjoint exon segments."""

    def __init__(self, loci):
        """loci: list (per locus) of lists (per isoform) of (left, right) exons."""
        iso_off, exon_off, xl, xr = [0], [0], [], []
        for isos in loci:
            for exons in isos:
                for (a, b) in exons:
                    xl.append(a)
                    xr.append(b)
                exon_off.append(len(xl))
            iso_off.append(len(exon_off) - 1)
        self.n_loci = len(loci)
        self.iso_off = np.asarray(iso_off, np.int64)
        self.exon_off = np.asarray(exon_off, np.int64)
        self.exon_left = np.asarray(xl, np.uint32)
        self.exon_right = np.asarray(xr, np.uint32)
        L = _lib.load()
        self.seg_off = np.zeros(self.n_loci + 1, np.int64)
        n = L.sbgpu_segments_host(self.n_loci, _ptr(self.iso_off), _ptr(self.exon_off), _ptr(self.exon_left),
                                  _ptr(self.exon_right), _ptr(self.seg_off), None, None, 0)
        if n < 0:
            _lib.check(int(n), "sbgpu_segments_host")
        self.seg_left = np.zeros(n, np.uint32)
        self.seg_right = np.zeros(n, np.uint32)
        n2 = L.sbgpu_segments_host(self.n_loci, _ptr(self.iso_off), _ptr(self.exon_off), _ptr(self.exon_left),
                                   _ptr(self.exon_right), _ptr(self.seg_off), _ptr(self.seg_left), _ptr(self.seg_right), n)
        assert n2 == n
        self.compat_words = max(1, int(-(-np.diff(self.iso_off).max(initial=0) // 32)))
        self.key_words = max(1, int(-(-np.diff(self.seg_off).max(initial=0) // 32)))

    def prefix(self, n_loci):
        """The first n_loci loci as an annotation of their own (copies)."""
        a = Annotation.__new__(Annotation)
        a.n_loci = int(n_loci)
        ni = int(self.iso_off[n_loci])
        ne, ns = int(self.exon_off[ni]), int(self.seg_off[n_loci])
        a.iso_off, a.exon_off, a.seg_off = self.iso_off[:n_loci + 1].copy(), self.exon_off[:ni + 1].copy(), self.seg_off[:n_loci + 1].copy()
        a.exon_left, a.exon_right = self.exon_left[:ne].copy(), self.exon_right[:ne].copy()
        a.seg_left, a.seg_right = self.seg_left[:ns].copy(), self.seg_right[:ns].copy()
        a.compat_words = max(1, int(-(-np.diff(a.iso_off).max(initial=0) // 32)))
        a.key_words = max(1, int(-(-np.diff(a.seg_off).max(initial=0) // 32)))
        return a

    @classmethod
    def concat(cls, parts):
        """Several annotations one after the other (their loci keep their order; the arrays are joined, the offsets shifted)."""
        a = cls.__new__(cls)
        a.n_loci = sum(p.n_loci for p in parts)

        def join_off(name, count_of):
            out, base = [np.zeros(1, np.int64)], 0
            for p in parts:
                out.append(getattr(p, name)[1:] + base)
                base += count_of(p)
            return np.concatenate(out)
        a.iso_off = join_off("iso_off", lambda p: int(p.iso_off[-1]))
        a.exon_off = join_off("exon_off", lambda p: int(p.exon_off[-1]))
        a.seg_off = join_off("seg_off", lambda p: int(p.seg_off[-1]))
        for name in ("exon_left", "exon_right", "seg_left", "seg_right"):
            setattr(a, name, np.concatenate([getattr(p, name) for p in parts]))
        a.compat_words = max(p.compat_words for p in parts)
        a.key_words = max(p.key_words for p in parts)
        return a

    def segments(self, locus):
        s = slice(self.seg_off[locus], self.seg_off[locus + 1])
        return list(zip(self.seg_left[s].tolist(), self.seg_right[s].tolist()))

    def _struct(self):
        s = _lib.sbgpu_annotation_t()
        s.n_loci = self.n_loci
        s.iso_off, s.exon_off, s.seg_off = _ptr(self.iso_off), _ptr(self.exon_off), _ptr(self.seg_off)
        s.exon_left, s.exon_right = _ptr(self.exon_left), _ptr(self.exon_right)
        s.seg_left, s.seg_right = _ptr(self.seg_left), _ptr(self.seg_right)
        return s


def mate_features(blocks):
    """A mate's aligned blocks [(l, r), ...] -> (code, left, right) with the introns between them -- none between two
    blocks that touch: an insertion in the read (readhit_2_genomicFeats, src/contig.cpp:12-53)."""
    code, left, right = [], [], []
    for k, (a, b) in enumerate(blocks):
        if k and a != blocks[k - 1][1] + 1:
            code.append(INTRON)
            left.append(blocks[k - 1][1] + 1)
            right.append(a - 1)
        code.append(MATCH)
        left.append(a)
        right.append(b)
    return code, left, right


def hit_features(left_blocks, right_blocks):
    """Contig(PairedHit): -> (code, left, right) lists, or None when the reference rejects the pair."""
    L = _lib.load()
    lc, ll, lr = (np.asarray(x, t) for x, t in zip(mate_features(left_blocks), (np.uint8, np.uint32, np.uint32)))
    rc, rl, rr = (np.asarray(x, t) for x, t in zip(mate_features(right_blocks), (np.uint8, np.uint32, np.uint32)))
    cap = len(lc) + len(rc) + 1
    oc, ol, orr = np.zeros(cap, np.uint8), np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
    n = L.sbgpu_hit_features(len(lc), _ptr(lc), _ptr(ll), _ptr(lr), len(rc), _ptr(rc), _ptr(rl), _ptr(rr),
                             oc.ctypes.data, ol.ctypes.data, orr.ctypes.data)
    if n < 0:
        _lib.check(n, "sbgpu_hit_features")
    if n == 0:
        return None
    return oc[:n].tolist(), ol[:n].tolist(), orr[:n].tolist()


class Reads:
    """Alignment records of a batch of clusters (sbgpu_reads_t), a cluster's records in the order the BAM gives them."""

    def __init__(self, read_locus, read_id, blocks, partner_pos, flags, nh):
        self.n_reads = len(blocks)
        self.read_locus = np.asarray(read_locus, np.int32)
        self.read_id = np.ascontiguousarray(read_id, np.uint64)
        off, bl, br = [0], [], []
        for b in blocks:
            for (x, y) in b:
                bl.append(x)
                br.append(y)
            off.append(len(bl))
        self.block_off = np.asarray(off, np.int64)
        self.block_left, self.block_right = np.asarray(bl, np.uint32), np.asarray(br, np.uint32)
        self.partner_pos = np.ascontiguousarray(partner_pos, np.uint32)
        self.flags = np.ascontiguousarray(flags, np.uint8)
        self.nh = np.ascontiguousarray(nh, np.int32)


def assign_reads(c_ref, c_left, c_right, c_strand, r_ref, r_left, r_right, flags, device=None):
    """Which cluster a position-sorted alignment record belongs to (Sample::nextClusterRefDemand's pass):
    sbgpu_assign_reads_host, or -- device: an em.Context -- sbgpu_assign_reads_device on uploaded records.
    -> (read_cluster int32[n_reads], cluster_read_off int64[n_clusters + 1], flags with SBGPU_READ_SKIP set on the others)"""
    L = _lib.load()
    c_ref, c_left, c_right = (np.ascontiguousarray(x, t) for x, t in ((c_ref, np.int32), (c_left, np.uint32), (c_right, np.uint32)))
    c_strand = np.ascontiguousarray(c_strand, np.uint8)
    r_ref, r_left, r_right = (np.ascontiguousarray(x, t) for x, t in ((r_ref, np.int32), (r_left, np.uint32), (r_right, np.uint32)))
    flags = np.ascontiguousarray(flags, np.uint8).copy()
    cl = _lib.sbgpu_clusters_t(len(c_ref), _ptr(c_ref), _ptr(c_left), _ptr(c_right), _ptr(c_strand))
    n = len(r_ref)
    out, off = np.zeros(max(n, 1), np.int32), np.zeros(len(c_ref) + 1, np.int64)
    if device is None:
        _lib.check(L.sbgpu_assign_reads_host(C.byref(cl), n, _ptr(r_ref), _ptr(r_left), _ptr(r_right), _ptr(flags), out.ctypes.data,
                                             off.ctypes.data), "sbgpu_assign_reads_host")
        return out[:n], off, flags
    import torch
    dev = torch.device("cuda", device.device)
    up = lambda x: torch.from_numpy(x.view(np.int32) if x.dtype == np.uint32 else x).to(dev) if x.size else torch.zeros(1, dtype=torch.int32, device=dev)  # noqa: E731
    d_ref, d_left, d_right, d_flags = up(r_ref), up(r_left), up(r_right), up(flags)
    d_out = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    _lib.check(L.sbgpu_assign_reads_device(device.h, C.byref(cl), n, d_ref.data_ptr(), d_left.data_ptr(), d_right.data_ptr(),
                                           d_flags.data_ptr(), d_out.data_ptr(), off.ctypes.data,
                                           C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "sbgpu_assign_reads_device")
    return d_out.cpu().numpy()[:n], off, d_flags.cpu().numpy()[:n]


def pair_mates(n_loci, reads, device=None):
    """Alignment records -> read pairs: HitCluster::addOpenHit + addHit (sbgpu_pair_mates_host; `device`: an
    em.Context -> sbgpu_pair_mates_device on uploaded records, results brought back for comparison).
    -> dict(pair_off [n_loci + 1], mass, left_off, left (code, left, right), right_off, right (...), info)"""
    L = _lib.load()
    r = reads
    off = np.searchsorted(r.read_locus, np.arange(n_loci + 1), side="left").astype(np.int64)
    handle = C.c_void_p()
    if device is None:
        rs = _lib.sbgpu_reads_t(r.n_reads, _ptr(r.read_id), _ptr(r.block_off), _ptr(r.block_left), _ptr(r.block_right),
                                _ptr(r.partner_pos), _ptr(r.flags), _ptr(r.nh))
        _lib.check(L.sbgpu_pair_mates_host(n_loci, C.byref(rs), off.ctypes.data, C.byref(handle)), "sbgpu_pair_mates_host")
        keep = None
    else:
        import torch
        dev = torch.device("cuda", device.device)
        def up(x):
            x = x.view(np.int32) if x.dtype == np.uint32 else (x.view(np.int64) if x.dtype == np.uint64 else x)
            return torch.from_numpy(np.ascontiguousarray(x)).to(dev) if x.size else torch.zeros(1, dtype=torch.int64, device=dev)
        keep = [up(x) for x in (r.read_id, r.block_off, r.block_left, r.block_right, r.partner_pos, r.flags, r.nh)]
        rs = _lib.sbgpu_reads_t(r.n_reads, *[t.data_ptr() for t in keep])
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(L.sbgpu_pair_mates_device(device.h, n_loci, C.byref(rs), off.ctypes.data, stream, C.byref(handle)), "sbgpu_pair_mates_device")
    try:
        info = (C.c_int64 * 8)()
        _lib.check(L.sbgpu_matepairs_info(handle, info), "sbgpu_matepairs_info")
        n_p, n_l, n_r = int(info[0]), int(info[5]), int(info[6])
        p = _lib.sbgpu_pairs_t()
        poff = C.c_void_p()
        _lib.check(L.sbgpu_matepairs_pairs(handle, C.byref(p), C.byref(poff)), "sbgpu_matepairs_pairs")
        pair_off = np.ctypeslib.as_array(C.cast(poff, C.POINTER(C.c_int64)), shape=(n_loci + 1,)).copy()
        mass = np.zeros(n_p)
        lo, ro = np.zeros(n_p + 1, np.int64), np.zeros(n_p + 1, np.int64)
        lc, ll, lr = np.zeros(n_l, np.uint8), np.zeros(n_l, np.uint32), np.zeros(n_l, np.uint32)
        rc, rl, rr = np.zeros(n_r, np.uint8), np.zeros(n_r, np.uint32), np.zeros(n_r, np.uint32)
        _lib.check(L.sbgpu_matepairs_export(handle, _ptr(mass), lo.ctypes.data, _ptr(lc), _ptr(ll), _ptr(lr), ro.ctypes.data,
                                            _ptr(rc), _ptr(rl), _ptr(rr)), "sbgpu_matepairs_export")
    finally:
        L.sbgpu_matepairs_destroy(handle)
    return {"pair_off": pair_off, "mass": mass, "left_off": lo, "left": (lc, ll, lr), "right_off": ro, "right": (rc, rl, rr),
            "info": {"pairs": n_p, "complete": int(info[1]), "single": int(info[2]), "refused": int(info[3]), "orphan": int(info[4]),
                     "on_device": bool(info[7] & 1)},
            # device form: did the positional matching serve the call (no sort)?  else why the sorted form did
            # (1 records not in position order, 2 a read id with several fitting mates)
            "positional": bool(info[7] & 2), "why_sorted": int(info[7]) >> 2}


def collapse_pairs(n_loci, pair_locus, pair_mass, left_blocks, right_blocks, device=None):
    """Aligned read pairs -> unique hits: HitCluster::collapseAndFilterHits + Contig(PairedHit)
    (sbgpu_collapse_pairs_host).  left_blocks / right_blocks: per pair the mate's aligned blocks [(l, r), ...]
    ([] for a missing mate); pair_mass: the pair's raw mass.  -> (Hits, cluster_mass[n_loci], info dict)
    device: an em.Context -- the pairs (grouped by locus here) are uploaded and collapsed on the GPU
    (sbgpu_collapse_pairs_device); the result comes back as host arrays for comparison."""
    L = _lib.load()
    if device is not None:
        # group by locus, keeping the input order inside a locus (what the host form does internally)
        order = np.argsort(np.asarray(pair_locus), kind="stable")
        pair_locus = [pair_locus[i] for i in order]
        pair_mass = [pair_mass[i] for i in order]
        left_blocks = [left_blocks[i] for i in order]
        right_blocks = [right_blocks[i] for i in order]
    def csr(blocks_list):
        off, c, l, r = [0], [], [], []
        for b in blocks_list:
            cc, ll, rr = mate_features(b)
            c += cc
            l += ll
            r += rr
            off.append(len(c))
        return (np.asarray(off, np.int64), np.asarray(c, np.uint8), np.asarray(l, np.uint32), np.asarray(r, np.uint32))
    lo, lc, ll, lr = csr(left_blocks)
    ro, rc, rl, rr = csr(right_blocks)
    loc = np.ascontiguousarray(pair_locus, np.int32)
    mass = np.ascontiguousarray(pair_mass, np.float64)
    p = _lib.sbgpu_pairs_t(len(loc), _ptr(loc), _ptr(mass), _ptr(lo), _ptr(lc), _ptr(ll), _ptr(lr), _ptr(ro), _ptr(rc),
                           _ptr(rl), _ptr(rr))
    handle = C.c_void_p()
    if device is not None:
        import torch
        dev = torch.device("cuda", device.device)
        keep = [torch.from_numpy(x.view(np.int32) if x.dtype == np.uint32 else x).to(dev) if x.size else torch.zeros(1, dtype=torch.int64, device=dev)
                for x in (mass, lo, lc, ll, lr, ro, rc, rl, rr)]
        dp = _lib.sbgpu_pairs_t(len(loc), None, *[t.data_ptr() for t in keep])
        poff = np.searchsorted(loc, np.arange(n_loci + 1), side="left").astype(np.int64)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(L.sbgpu_collapse_pairs_device(device.h, n_loci, C.byref(dp), poff.ctypes.data, stream, C.byref(handle)),
                   "sbgpu_collapse_pairs_device")
        try:
            info = (C.c_int64 * 8)()
            _lib.check(L.sbgpu_uniq_dev_info(handle, info), "sbgpu_uniq_dev_info")
            nh, nf = int(info[0]), int(info[1])
            hit_locus, feat_off = np.zeros(nh, np.int32), np.zeros(nh + 1, np.int64)
            code, left, right = np.zeros(nf, np.uint8), np.zeros(nf, np.uint32), np.zeros(nf, np.uint32)
            hmass, cmass = np.zeros(nh, np.float32), np.zeros(n_loci, np.float64)
            _lib.check(L.sbgpu_uniq_dev_export(handle, _ptr(hit_locus), feat_off.ctypes.data, _ptr(code), _ptr(left), _ptr(right),
                                               _ptr(hmass), _ptr(cmass)), "sbgpu_uniq_dev_export")
        finally:
            L.sbgpu_uniq_dev_destroy(handle)
        hits = Hits.from_arrays(hit_locus, feat_off, code, left, right, hmass)
        hits.total_mapped = int(info[4])
        return hits, cmass, {"filtered": int(info[2]), "rejected": int(info[3]), "total_mapped": int(info[4])}
    _lib.check(L.sbgpu_collapse_pairs_host(n_loci, C.byref(p), C.byref(handle)), "sbgpu_collapse_pairs_host")
    try:
        info = (C.c_int64 * 8)()
        _lib.check(L.sbgpu_uniq_info(handle, info), "sbgpu_uniq_info")
        nh, nf = int(info[0]), int(info[1])
        hit_locus, feat_off = np.zeros(nh, np.int32), np.zeros(nh + 1, np.int64)
        code, left, right = np.zeros(nf, np.uint8), np.zeros(nf, np.uint32), np.zeros(nf, np.uint32)
        hmass, cmass = np.zeros(nh, np.float32), np.zeros(n_loci, np.float64)
        _lib.check(L.sbgpu_uniq_export(handle, _ptr(hit_locus), feat_off.ctypes.data, _ptr(code), _ptr(left), _ptr(right),
                                       _ptr(hmass), _ptr(cmass)), "sbgpu_uniq_export")
    finally:
        L.sbgpu_uniq_destroy(handle)
    hits = Hits.from_arrays(hit_locus, feat_off, code, left, right, hmass)
    hits.total_mapped = int(info[4])
    return hits, cmass, {"filtered": int(info[2]), "rejected": int(info[3]), "total_mapped": int(info[4])}


class Hits:
    """Fragments of a batch of loci, CSR over features; `mass` = (float) collapse_mass."""

    def __init__(self, hit_locus, feats, mass=None):
        """feats: per hit (code, left, right) lists."""
        self.n_hits = len(feats)
        self.hit_locus = np.asarray(hit_locus, np.int32)
        off = np.zeros(self.n_hits + 1, np.int64)
        for h, f in enumerate(feats):
            off[h + 1] = off[h] + len(f[0])
        self.feat_off = off
        self.feat_code = np.asarray([c for f in feats for c in f[0]], np.uint8)
        self.feat_left = np.asarray([c for f in feats for c in f[1]], np.uint32)
        self.feat_right = np.asarray([c for f in feats for c in f[2]], np.uint32)
        self.mass = np.ones(self.n_hits, np.float32) if mass is None else np.asarray(mass, np.float32)

    @classmethod
    def from_arrays(cls, hit_locus, feat_off, feat_code, feat_left, feat_right, mass=None):
        self = cls.__new__(cls)
        self.n_hits = len(hit_locus)
        self.hit_locus = np.ascontiguousarray(hit_locus, np.int32)
        self.feat_off = np.ascontiguousarray(feat_off, np.int64)
        self.feat_code = np.ascontiguousarray(feat_code, np.uint8)
        self.feat_left = np.ascontiguousarray(feat_left, np.uint32)
        self.feat_right = np.ascontiguousarray(feat_right, np.uint32)
        self.mass = np.ones(self.n_hits, np.float32) if mass is None else np.ascontiguousarray(mass, np.float32)
        return self

    def _struct(self):
        s = _lib.sbgpu_hits_t()
        s.n_hits = self.n_hits
        s.hit_locus, s.feat_off = _ptr(self.hit_locus), _ptr(self.feat_off)
        s.feat_code, s.feat_left, s.feat_right = _ptr(self.feat_code), _ptr(self.feat_left), _ptr(self.feat_right)
        return s


def compat_and_keys(annot, hits, ctx=None, device=0, compat_words=None, key_words=None):
    """-> (compat [n_hits, compat_words] uint32, key [n_hits, key_words] uint32) from the HIP kernel."""
    ctx = ctx or default_context(device)
    cw = annot.compat_words if compat_words is None else compat_words
    kw = annot.key_words if key_words is None else key_words
    compat = np.zeros((max(hits.n_hits, 1), max(cw, 1)), np.uint32)
    key = np.zeros((max(hits.n_hits, 1), max(kw, 1)), np.uint32)
    a, h = annot._struct(), hits._struct()
    _lib.check(ctx.L.sbgpu_exonbin_host(ctx.h, C.byref(a), C.byref(h), cw, kw, compat.ctypes.data, key.ctypes.data),
               "sbgpu_exonbin_host")
    return compat[:hits.n_hits, :cw], key[:hits.n_hits, :kw]


def frag_lens(annot, hits, compat):
    """The empirical insert-size sample (Sample::fragLenDist): the exonic span of every hit that is
    compatible with exactly one transcript, in hit order.  -> int array (only those hits)."""
    L = _lib.load()
    compat = np.ascontiguousarray(compat, np.uint32)
    out = np.full(max(hits.n_hits, 1), -1, np.int32)
    a, h = annot._struct(), hits._struct()
    n = L.sbgpu_frag_lens_host(C.byref(a), C.byref(h), compat.shape[1], _ptr(compat), out.ctypes.data)
    if n < 0:
        _lib.check(int(n), "sbgpu_frag_lens_host")
    out = out[:hits.n_hits]
    return out[out >= 0]


class LocusBins:
    """The exon bins of a batch of loci: an EM batch minus F, and the bin-weight kernel's pairs."""

    def __init__(self, annot, hits, compat, key):
        """Host grouping (sbgpu_bins_create) from host copies of the kernel's words."""
        L = _lib.load()
        compat = np.ascontiguousarray(compat, np.uint32)
        key = np.ascontiguousarray(key, np.uint32)
        cw = compat.shape[1] if compat.ndim == 2 else annot.compat_words
        kw = key.shape[1] if key.ndim == 2 else annot.key_words
        a, h = annot._struct(), hits._struct()
        handle = C.c_void_p()
        _lib.check(L.sbgpu_bins_create(C.byref(a), C.byref(h), _ptr(hits.mass), cw, kw, _ptr(compat), _ptr(key),
                                       C.byref(handle)), "sbgpu_bins_create")
        self._export(L, annot, handle, hits.n_hits, cw, kw, with_hit_bin=True)

    @classmethod
    def on_device(cls, ctx, annot, hits, d_hits_struct, d_mass_ptr, cw, kw, d_compat_ptr, d_key_ptr, d_hit_bin_ptr, stream):
        """Device grouping (sbgpu_bins_create_device): the words stay in HBM.  Returns None when the device
        form does not cover the input (unsorted hits, fractional masses, a locus with thousands of bins);
        the caller then uses the host form."""
        L = _lib.load()
        if hits.n_hits and np.any(np.diff(hits.hit_locus) < 0):
            return None
        off = np.searchsorted(hits.hit_locus, np.arange(annot.n_loci + 1), side="left").astype(np.int64)
        a = annot._struct()
        handle = C.c_void_p()
        rc = L.sbgpu_bins_create_device(ctx.h, C.byref(a), C.byref(d_hits_struct), d_mass_ptr, off.ctypes.data, cw, kw,
                                        d_compat_ptr, d_key_ptr, d_hit_bin_ptr, stream, C.byref(handle))
        if rc == _lib.SBGPU_EUNSUPPORTED:
            return None
        _lib.check(rc, "sbgpu_bins_create_device")
        self = cls.__new__(cls)
        self._export(L, annot, handle, hits.n_hits, cw, kw, with_hit_bin=False)
        return self

    def _export(self, L, annot, handle, n_hits, cw, kw, with_hit_bin):
        try:
            info = (C.c_int64 * 8)()
            _lib.check(L.sbgpu_bins_info(handle, info), "sbgpu_bins_info")
            (self.n_loci, self.n_iso, self.n_bins, self.n_elem, self.n_pairs, n_pair_segs, self.n_hits_used, _) = list(info)
            on_dev, why = C.c_int32(0), C.c_char_p()
            _lib.check(L.sbgpu_bins_grouping(handle, C.byref(on_dev), C.byref(why)), "sbgpu_bins_grouping")
            # which grouping made these bins: the device kernels, or the library's host code (and then why)
            self.grouped_on_device, self.host_grouping_reason = bool(on_dev.value), (why.value or b"").decode()
            self.row_off = np.zeros(self.n_loci + 1, np.int64)
            self.iso_off = np.zeros(self.n_loci + 1, np.int64)
            self.f_off = np.zeros(self.n_loci + 1, np.int64)
            self.count = np.zeros(self.n_bins, np.int32)
            self.iso_len = np.zeros(self.n_iso, np.int32)
            self.bin_key = np.zeros((self.n_bins, kw), np.uint32)
            self.bin_compat = np.zeros((self.n_bins, cw), np.uint32)
            self.hit_bin = np.zeros(n_hits, np.int64) if with_hit_bin else None   # device form: see d_hit_bin
            self.pair_seg_off = np.zeros(self.n_pairs + 1, np.int64)
            self.pair_seg_lens = np.zeros(n_pair_segs, np.uint32)
            self.pair_implicit_mask = np.zeros(self.n_pairs, np.uint32)
            self.pair_iso_len = np.zeros(self.n_pairs, np.int32)
            self.pair_out_index = np.zeros(self.n_pairs, np.int64)
            _lib.check(L.sbgpu_bins_export(
                handle, _ptr(self.row_off), _ptr(self.iso_off), _ptr(self.f_off), _ptr(self.count), _ptr(self.iso_len),
                _ptr(self.bin_key), _ptr(self.bin_compat), _ptr(self.hit_bin) if with_hit_bin else None,
                _ptr(self.pair_seg_off), _ptr(self.pair_seg_lens), _ptr(self.pair_implicit_mask), _ptr(self.pair_iso_len),
                _ptr(self.pair_out_index)), "sbgpu_bins_export")
        finally:
            L.sbgpu_bins_destroy(handle)
        self._annot = annot

    def bin_coords(self, locus):
        """Per bin of the locus, the segments it spans [(l, r), ...] (ExonBin::_coords)."""
        segs = self._annot.segments(locus)
        out = []
        for b in range(self.row_off[locus], self.row_off[locus + 1]):
            words = self.bin_key[b]
            out.append([s for k, s in enumerate(segs) if (int(words[k >> 5]) >> (k & 31)) & 1])
        return out
