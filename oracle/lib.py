"""ctypes bindings for the oracle libraries (test infrastructure only)."""
import ctypes as C
import os
import subprocess
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
REF_LIB = os.path.join(HERE, "_ref", "libstrawberry_ref.so")
REF_BIN = os.path.join(HERE, "_ref", "strawberry_ref")
SAM2BAM = os.path.join(HERE, "_ref", "sam2bam")
REFERENCE_ROOT = "/root/reference"

SBO_EM_OK, SBO_EM_INIT_EMPTY, SBO_EM_DENOM_ZERO, SBO_EM_MAXITER = 0, 1, 2, 3

_p = np.ctypeslib.ndpointer
_i32 = _p(np.int32, flags="C")
_i64 = _p(np.int64, flags="C")
_u32 = _p(np.uint32, flags="C")
_f64 = _p(np.float64, flags="C")


def build(with_ref=None):
    """Compile liboracle.so; compile oracle/_ref too when /root/reference is mounted."""
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    if with_ref is None:
        with_ref = os.path.isdir(os.path.join(REFERENCE_ROOT, "src"))
    if with_ref:
        subprocess.check_call(["make", "-s", "-j4", "-C", HERE, "ref"])


def have_ref():
    return os.path.exists(REF_LIB)


class sbo_insert_t(C.Structure):
    _fields_ = [
        ("mean", C.c_double),
        ("sd", C.c_double),
        ("use_emp", C.c_int32),
        ("start_offset", C.c_int32),
        ("end_offset", C.c_int32),
        ("total_reads", C.c_int32),
        ("emp_hist", C.POINTER(C.c_double)),
    ]


def _csr(row_off, iso_off, f_off, count, F):
    return (
        np.ascontiguousarray(row_off, np.int64),
        np.ascontiguousarray(iso_off, np.int64),
        np.ascontiguousarray(f_off, np.int64),
        np.ascontiguousarray(count, np.int32),
        np.ascontiguousarray(F, np.float64),
    )


def _run_threads(fn, n_loci, threads):
    if threads <= 1 or n_loci < 2 * threads:
        fn(0, n_loci)
        return
    # static contiguous partition (the analogue of `-p T`); ctypes
    # releases the GIL for the duration of each foreign call
    bounds = [n_loci * t // threads for t in range(threads + 1)]
    ts = [threading.Thread(target=fn, args=(bounds[t], bounds[t + 1])) for t in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()


class OracleLib:
    """liboracle.so -- the plain-C restatement."""

    def __init__(self, path=LIB):
        if not os.path.exists(path):
            build(with_ref=False)
        L = self.L = C.CDLL(path)
        L.sbo_em_locus.argtypes = [C.c_int, C.c_int, _i32, _f64, _f64, C.POINTER(C.c_int32)]
        L.sbo_em_locus.restype = C.c_int
        L.sbo_em_batch.argtypes = [C.c_int64, C.c_int64, _i64, _i64, _i64, _i32, _f64, _f64, _i32, _i32]
        L.sbo_em_batch.restype = None
        L.sbo_abundance_locus.argtypes = [C.c_int, _f64, _i32, C.c_int32, C.c_int, C.c_double, C.c_int,
                                          C.c_double, _f64, _f64, _i32]
        L.sbo_abundance_locus.restype = C.c_double
        L.sbo_tpm.argtypes = [C.c_int64, _f64, _i32, _f64]
        L.sbo_tpm.restype = C.c_double
        L.sbo_insert_pdf.argtypes = [C.POINTER(sbo_insert_t), C.c_uint32]
        L.sbo_insert_pdf.restype = C.c_double
        L.sbo_no_gap_ef.argtypes = [C.c_int] * 4
        L.sbo_no_gap_ef.restype = C.c_int
        L.sbo_gap_ef.argtypes = [C.c_int] * 5
        L.sbo_gap_ef.restype = C.c_int
        L.sbo_effective_len.argtypes = [C.c_int, _u32, C.c_int, _u32, C.c_int, C.c_int]
        L.sbo_effective_len.restype = C.c_int
        L.sbo_bin_weight.argtypes = [C.c_int, _u32, C.c_int, _u32, C.c_int, C.c_int, C.POINTER(sbo_insert_t)]
        L.sbo_bin_weight.restype = C.c_double
        _u8 = _p(np.uint8, flags="C")
        L.sbo_is_compatible.argtypes = [C.c_int, _u8, _u32, _u32, C.c_int, _u32, _u32]
        L.sbo_is_compatible.restype = C.c_int
        L.sbo_overlap_key.argtypes = [C.c_int, _u8, _u32, _u32, C.c_int, _u32, _u32, _u8]
        L.sbo_overlap_key.restype = None
        L.sbo_exonbin_batch.argtypes = [_i64, _i64, _u32, _u32, _i64, _u32, _u32, C.c_int64, _p(np.int32, flags="C"),
                                        _i64, _u8, _u32, _u32, C.c_int32, C.c_int32, _u32, _u32]
        L.sbo_exonbin_batch.restype = None
        L.sbo_gc_ratio.argtypes = [_u8, C.c_int64]
        L.sbo_gc_ratio.restype = C.c_double
        L.sbo_kmer_entropy.argtypes = [_u8, C.c_int64, C.c_int]
        L.sbo_kmer_entropy.restype = C.c_double
        L.sbo_high_gc_stretch.argtypes = [_u8, C.c_int64, C.c_int, C.c_double]
        L.sbo_high_gc_stretch.restype = C.c_int
        L.sbo_binseq_batch.argtypes = [_u8, C.c_int64, C.c_int64, _i64, _u32, _u32, _f64, _f64, _u8]
        L.sbo_binseq_batch.restype = None
        L.sbo_collapse_cluster.argtypes = [C.c_int, _i64, _u32, _u32, _i64, _u32, _u32, _i32, _i32, _f64, _f64, _i32]
        L.sbo_collapse_cluster.restype = C.c_int
        L.sbo_phi.argtypes = [C.c_double]
        L.sbo_phi.restype = C.c_double
        _u64 = _p(np.uint64, flags="C")
        L.sbo_pair_mates.argtypes = [C.c_int, _u64, _i64, _u32, _u32, _u32, _p(np.uint8, flags="C"), _i32, _i32, _i32, _f64, _i32]
        L.sbo_pair_mates.restype = C.c_int
        L.sbo_assign_reads.argtypes = [C.c_int, _i32, _u32, _u32, _p(np.uint8, flags="C"), C.c_int64, _i32, _u32, _u32, _p(np.uint8, flags="C"),
                                       _i32, _i64]
        L.sbo_assign_reads.restype = None

    def assign_reads(self, c_ref, c_left, c_right, c_strand, r_ref, r_left, r_right, r_xs):
        """Sample::nextClusterRefDemand's pass.  -> (read_cluster int32[n_reads], off int64[n_clusters + 1])"""
        nc, nr = len(c_ref), len(r_ref)
        out, off = np.zeros(max(nr, 1), np.int32), np.zeros(nc + 1, np.int64)
        a = lambda x, t: np.ascontiguousarray(x if len(x) else [0], t)  # noqa: E731
        self.L.sbo_assign_reads(nc, a(c_ref, np.int32), a(c_left, np.uint32), a(c_right, np.uint32), a(c_strand, np.uint8), nr,
                                a(r_ref, np.int32), a(r_left, np.uint32), a(r_right, np.uint32), a(r_xs, np.uint8), out, off)
        return out[:nr].copy(), off

    # ---- BAM records -> ReadHits (BAMHitFactory::getHitFromBuf)
    def bam_index(self, rec_bytes):
        """Record offsets of an uncompressed BAM record stream -> int64[n + 1]."""
        b = np.ascontiguousarray(rec_bytes, np.uint8)
        cap = b.size // 36 + 1
        off = np.zeros(cap + 1, np.int64)
        self.L.sbo_bam_index.argtypes = [_p(np.uint8, flags="C"), C.c_int64, _i64, C.c_int64]
        self.L.sbo_bam_index.restype = C.c_int64
        n = self.L.sbo_bam_index(b if b.size else np.zeros(1, np.uint8), b.size, off, cap)
        if n < 0:
            raise ValueError("sbo_bam_index: the stream ends inside a record")
        return off[:n + 1].copy()

    def bam_decode(self, rec_bytes, rec_off=None, min_intron=20, max_intron=300000, unique_only=True, library=0, n_ref=0):
        """oracle/bamdecode_oracle.c on an uncompressed record stream -> dict of per-record arrays (see the header)."""
        b = np.ascontiguousarray(rec_bytes, np.uint8)
        off = self.bam_index(b) if rec_off is None else np.ascontiguousarray(rec_off, np.int64)
        n = off.size - 1
        m = max(b.size // 4, 1)
        _u8, _u64 = _p(np.uint8, flags="C"), _p(np.uint64, flags="C")
        opts = (C.c_int32 * 5)(int(min_intron), int(max_intron), int(bool(unique_only)), int(library), int(n_ref))
        N = max(n, 1)
        z = dict(status=np.zeros(N, np.uint8), read_id=np.zeros(N, np.uint64), ref=np.zeros(N, np.int32), left=np.zeros(N, np.uint32),
                 right=np.zeros(N, np.uint32), strand=np.zeros(N, np.uint8), partner_same_ref=np.zeros(N, np.uint8),
                 partner_pos=np.zeros(N, np.uint32), nm=np.zeros(N, np.int32), nh=np.zeros(N, np.int32), sam_flag=np.zeros(N, np.uint32),
                 singleton=np.zeros(N, np.uint8), mass=np.zeros(N, np.float64), read_len=np.zeros(N, np.int32),
                 cig_off=np.zeros(N + 1, np.int64), cig_type=np.zeros(m, np.uint8), cig_len=np.zeros(m, np.uint32),
                 feat_off=np.zeros(N + 1, np.int64), feat_code=np.zeros(m, np.uint8), feat_left=np.zeros(m, np.uint32),
                 feat_right=np.zeros(m, np.uint32))
        ap = np.zeros(1, np.int32)
        self.L.sbo_bam_decode.argtypes = [_u8, _i64, C.c_int64, C.c_void_p, _u8, _u64, _i32, _u32, _u32, _u8, _u8, _u32, _i32, _i32, _u32, _u8,
                                          _f64, _i32, _i64, _u8, _u32, _i64, _u8, _u32, _u32, _i32]
        self.L.sbo_bam_decode.restype = None
        self.L.sbo_bam_decode(b if b.size else np.zeros(1, np.uint8), off, n, C.cast(opts, C.c_void_p), z["status"], z["read_id"], z["ref"],
                              z["left"], z["right"], z["strand"], z["partner_same_ref"], z["partner_pos"], z["nm"], z["nh"], z["sam_flag"],
                              z["singleton"], z["mass"], z["read_len"], z["cig_off"], z["cig_type"], z["cig_len"], z["feat_off"],
                              z["feat_code"], z["feat_left"], z["feat_right"], ap)
        for key in list(z):
            if key in ("cig_off", "feat_off"):
                z[key] = z[key][:n + 1].copy()
            elif key in ("cig_type", "cig_len"):
                z[key] = z[key][:z["cig_off"][n]].copy()
            elif key in ("feat_code", "feat_left", "feat_right"):
                z[key] = z[key][:z["feat_off"][n]].copy()
            else:
                z[key] = z[key][:n].copy()
        z["any_paired"] = int(ap[0])
        z["n"] = n
        z["rec_off"] = off
        return z

    # ---- mate pairing (HitCluster::addOpenHit + addHit)
    def pair_mates(self, read_id, blocks, partner_pos, flags, nh):
        """One cluster's records in arrival order (blocks: per record [(l, r), ...]).
        -> (left_rec int32[n_pairs], right_rec, mass float64[n_pairs], counts dict)"""
        off, bl, br = _blocks_csr(blocks)
        n = len(blocks)
        lr, rr, m = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.float64)
        cnt = np.zeros(4, np.int32)
        k = self.L.sbo_pair_mates(n, np.ascontiguousarray(read_id, np.uint64), off, bl, br,
                                  np.ascontiguousarray(partner_pos if n else [0], np.uint32),
                                  np.ascontiguousarray(flags if n else [0], np.uint8), np.ascontiguousarray(nh if n else [1], np.int32),
                                  lr, rr, m, cnt)
        return lr[:k].copy(), rr[:k].copy(), m[:k].copy(), {"complete": int(cnt[0]), "single": int(cnt[1]), "refused": int(cnt[2]),
                                                           "orphan": int(cnt[3])}

    # ---- duplicate collapse (HitCluster::collapseAndFilterHits)
    def collapse_cluster(self, left_blocks, right_blocks, nh):
        """One cluster: per pair the mates' aligned blocks [(l, r), ...] ([] = no such mate) and the NH tag.
        -> (uniq_pair int32[n_uniq], uniq_mass float64[n_uniq], cluster_mass, n_filtered)"""
        return _collapse_call(self.L.sbo_collapse_cluster, left_blocks, right_blocks, nh, with_filtered=True)

    # ---- EM
    def em_locus(self, count, F):
        """-> (theta[niso], status, iters) for one locus; F is (nrow, niso)."""
        F = np.ascontiguousarray(F, np.float64)
        nrow, niso = F.shape
        count = np.ascontiguousarray(count, np.int32)
        assert count.shape == (nrow,)
        theta = np.zeros(niso, np.float64)
        it = C.c_int32(0)
        st = self.L.sbo_em_locus(nrow, niso, count, F.reshape(-1), theta, C.byref(it))
        return theta, st, it.value

    def em_batch(self, row_off, iso_off, f_off, count, F, threads=1):
        row_off, iso_off, f_off, count, F = _csr(row_off, iso_off, f_off, count, F)
        n = len(row_off) - 1
        theta = np.zeros(int(iso_off[-1]), np.float64)
        status = np.zeros(n, np.int32)
        iters = np.zeros(n, np.int32)
        _run_threads(
            lambda lo, hi: self.L.sbo_em_batch(lo, hi, row_off, iso_off, f_off, count, F, theta, status, iters),
            n, threads)
        return theta, status, iters

    # ---- epilogue
    def abundance_locus(self, theta, length, total_mapped, effective_len_norm=False, insert_mean=0.0,
                        filter_by_expression=True, min_isoform_frac=0.01):
        theta = np.ascontiguousarray(theta, np.float64)
        length = np.ascontiguousarray(length, np.int32)
        n = len(theta)
        fpkm, frac, keep = np.zeros(n), np.zeros(n), np.zeros(n, np.int32)
        s = self.L.sbo_abundance_locus(n, theta, length, int(total_mapped), int(effective_len_norm),
                                       float(insert_mean), int(filter_by_expression), float(min_isoform_frac),
                                       fpkm, frac, keep)
        return fpkm, frac, keep, s

    def abundance(self, iso_off, theta, status, length, total_mapped_reads, effective_len_norm=False,
                  insert_mean=0.0, filter_by_expression=True, min_isoform_frac=0.01):
        """The epilogue for a whole batch: sbo_abundance_locus per locus (loci whose init() failed are
        dropped: quantifyCluster returns nothing for them, src/alignments.cpp:1524-1545), then sbo_tpm
        over the survivors.  -> dict(fpkm, frac, keep, tpm, sum_fpkm)."""
        iso_off = np.ascontiguousarray(iso_off, np.int64)
        theta = np.ascontiguousarray(theta, np.float64)
        length = np.ascontiguousarray(length, np.int32)
        n = len(theta)
        fpkm, frac, keep = np.zeros(n), np.zeros(n), np.zeros(n, np.int32)
        for l in range(len(iso_off) - 1):
            if status[l] == 1:
                continue
            j0, j1 = int(iso_off[l]), int(iso_off[l + 1])
            self.L.sbo_abundance_locus(j1 - j0, theta[j0:j1], length[j0:j1], int(total_mapped_reads),
                                       int(effective_len_norm), float(insert_mean), int(filter_by_expression),
                                       float(min_isoform_frac), fpkm[j0:j1], frac[j0:j1], keep[j0:j1])
        tpm, total = self.tpm(fpkm, keep)
        return {"fpkm": fpkm, "frac": frac, "keep": keep, "tpm": tpm, "sum_fpkm": total}

    def tpm(self, fpkm, keep):
        fpkm = np.ascontiguousarray(fpkm, np.float64)
        keep = np.ascontiguousarray(keep, np.int32)
        out = np.zeros(len(fpkm))
        total = self.L.sbo_tpm(len(fpkm), fpkm, keep, out)
        return out, total

    # ---- bin-weight model
    @staticmethod
    def make_insert(mean=0.0, sd=1.0, frag_lens=None):
        """Gaussian when frag_lens is None, else the empirical law built like
        InsertSize(vector<int>) (src/read.cpp:241-272, mean/sd via
        mean_and_sd_insert_size)."""
        ins = sbo_insert_t()
        ins._keep = None
        if frag_lens is None:
            ins.mean, ins.sd, ins.use_emp = float(mean), float(sd), 0
            return ins
        fl = np.asarray(frag_lens, np.int64)
        lo, hi = int(fl.min()), int(fl.max())
        hist = np.bincount(fl - lo, minlength=hi - lo + 1).astype(np.float64)
        ins._keep = hist
        ins.mean, ins.sd, ins.use_emp = float(mean), float(sd), 1
        ins.start_offset, ins.end_offset, ins.total_reads = lo, hi, len(fl)
        ins.emp_hist = hist.ctypes.data_as(C.POINTER(C.c_double))
        return ins

    def insert_pdf(self, ins, fl):
        return self.L.sbo_insert_pdf(C.byref(ins), int(fl))

    def effective_len(self, seg_lens, implicit_idx, fl, rl):
        s = np.ascontiguousarray(seg_lens, np.uint32)
        m = np.ascontiguousarray(implicit_idx, np.uint32)
        if len(m) == 0:
            m = np.zeros(1, np.uint32)
            nimp = 0
        else:
            nimp = len(m)
        return self.L.sbo_effective_len(len(s), s, nimp, m, int(fl), int(rl))

    def bin_weight(self, seg_lens, implicit_idx, iso_len, rl, ins):
        s = np.ascontiguousarray(seg_lens, np.uint32)
        m = np.ascontiguousarray(implicit_idx, np.uint32)
        nimp = len(m)
        if nimp == 0:
            m = np.zeros(1, np.uint32)
        return self.L.sbo_bin_weight(len(s), s, nimp, m, int(iso_len), int(rl), C.byref(ins))


    # ---- exon-bin assignment, integer part
    def is_compatible(self, code, left, right, exon_left, exon_right):
        code = np.ascontiguousarray(code, np.uint8)
        return bool(self.L.sbo_is_compatible(len(code), code, np.ascontiguousarray(left, np.uint32),
                                             np.ascontiguousarray(right, np.uint32), len(exon_left),
                                             np.ascontiguousarray(exon_left, np.uint32),
                                             np.ascontiguousarray(exon_right, np.uint32)))

    def overlap_key(self, code, left, right, seg_left, seg_right):
        code = np.ascontiguousarray(code, np.uint8)
        out = np.zeros(max(1, len(seg_left)), np.uint8)
        self.L.sbo_overlap_key(len(code), code, np.ascontiguousarray(left, np.uint32),
                               np.ascontiguousarray(right, np.uint32), len(seg_left),
                               np.ascontiguousarray(seg_left, np.uint32), np.ascontiguousarray(seg_right, np.uint32), out)
        return out[:len(seg_left)]


    def exonbin_batch(self, annot, hits, compat_words=None, key_words=None):
        """(compat, key) words for every hit; annot / hits carry the arrays of include/sbgpu.h's
        sbgpu_annotation_t / sbgpu_hits_t as numpy attributes of the same names."""
        cw = compat_words or annot.compat_words
        kw = key_words or annot.key_words
        compat = np.zeros((max(hits.n_hits, 1), cw), np.uint32)
        key = np.zeros((max(hits.n_hits, 1), kw), np.uint32)
        pad = lambda a, t: np.ascontiguousarray(a if len(a) else np.zeros(1, t), t)  # noqa: E731
        self.L.sbo_exonbin_batch(
            pad(annot.iso_off, np.int64), pad(annot.exon_off, np.int64), pad(annot.exon_left, np.uint32),
            pad(annot.exon_right, np.uint32), pad(annot.seg_off, np.int64), pad(annot.seg_left, np.uint32),
            pad(annot.seg_right, np.uint32), hits.n_hits, pad(hits.hit_locus, np.int32), pad(hits.feat_off, np.int64),
            pad(hits.feat_code, np.uint8), pad(hits.feat_left, np.uint32), pad(hits.feat_right, np.uint32), cw, kw,
            compat.reshape(-1), key.reshape(-1))
        return compat[:hits.n_hits], key[:hits.n_hits]


    # ---- per-bin sequence statistics (A8)
    def seq_stats(self, seq):
        """(gc, entropy, flags) of one sequence given as bytes."""
        a = np.frombuffer(bytes(seq), np.uint8).copy() if len(seq) else np.zeros(1, np.uint8)
        n = len(seq)
        flags = 0
        for q, (w, cut) in enumerate(((20, 0.8), (20, 0.9), (40, 0.8), (40, 0.9))):
            flags |= self.L.sbo_high_gc_stretch(a, n, w, cut) << q
        return self.L.sbo_gc_ratio(a, n), self.L.sbo_kmer_entropy(a, n, 6), flags

    def binseq_batch(self, genome, genome_start, seg_off, seg_left, seg_right):
        """The six `-f` columns of every bin: (gc[n], entropy[n], flags[n])."""
        seg_off = np.ascontiguousarray(seg_off, np.int64)
        n = len(seg_off) - 1
        gc, ent, fl = np.zeros(max(n, 1)), np.zeros(max(n, 1)), np.zeros(max(n, 1), np.uint8)
        pad = lambda a, t: np.ascontiguousarray(a if len(a) else np.zeros(1, t), t)  # noqa: E731
        self.L.sbo_binseq_batch(pad(np.frombuffer(bytes(genome), np.uint8), np.uint8), genome_start, n, seg_off,
                                pad(seg_left, np.uint32), pad(seg_right, np.uint32), gc, ent, fl)
        return gc[:n], ent[:n], fl[:n]


def write_sam_from_hits(path, hits, lib=None):
    """Unique hits (exonbin.Hits: left mate's features, one GAP, right mate's features; mass = read pairs behind the hit)
    -> the coordinate-sorted SAM of those read pairs (oracle/sam_writer.c).  -> number of records"""
    L = (lib or OracleLib()).L
    cnt = np.asarray(hits.mass).astype(np.int64)
    pair_off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    gap_idx = np.flatnonzero(hits.feat_code == 2).astype(np.int64)
    assert len(gap_idx) == hits.n_hits, "every hit must hold exactly one GAP (paired-end hits)"
    fl = hits.feat_left.astype(np.int64)
    pl, pr = fl[hits.feat_off[:-1]], fl[gap_idx + 1]
    pos = np.stack([np.repeat(pl, cnt), np.repeat(pr, cnt)], 1).ravel()
    order = np.argsort(pos, kind="stable").astype(np.int64)
    L.sbo_write_sam.restype = C.c_int
    L.sbo_write_sam.argtypes = [C.c_char_p, C.c_int64, C.c_int64, _i64, _p(np.uint8, flags="C"), _u32, _u32, _i64, _i64, C.c_int64, _i64]
    rc = L.sbo_write_sam(path.encode(), int(hits.feat_right.max()) + 10000, hits.n_hits, np.ascontiguousarray(hits.feat_off, np.int64),
                         np.ascontiguousarray(hits.feat_code, np.uint8), np.ascontiguousarray(hits.feat_left, np.uint32),
                         np.ascontiguousarray(hits.feat_right, np.uint32), gap_idx, pair_off, len(order), order)
    if rc != 0:
        raise RuntimeError("sbo_write_sam failed for " + path)
    return len(order)


def write_gtf_from_annotation(path, annot, n_loci=None):
    """The first n_loci gene models of an exonbin.Annotation as a GTF on chr1, plus strand (genes G<l>, transcripts G<l>.<j+1>)."""
    a = annot
    n_loci = a.n_loci if n_loci is None else n_loci
    el, er = a.exon_left.tolist(), a.exon_right.tolist()
    with open(path, "w") as f:
        for l in range(n_loci):
            for j, i in enumerate(range(int(a.iso_off[l]), int(a.iso_off[l + 1]))):
                e0, e1 = int(a.exon_off[i]), int(a.exon_off[i + 1])
                attr = 'gene_id "G%d"; transcript_id "G%d.%d";' % (l, l, j + 1)
                f.write("chr1\tsynth\ttranscript\t%d\t%d\t.\t+\t.\t%s\n" % (el[e0], er[e1 - 1], attr))
                for e in range(e0, e1):
                    f.write("chr1\tsynth\texon\t%d\t%d\t.\t+\t.\t%s\n" % (el[e], er[e], attr))


def _blocks_csr(blocks_list):
    off, l, r = [0], [], []
    for b in blocks_list:
        for (x, y) in b:
            l.append(x)
            r.append(y)
        off.append(len(l))
    pad = lambda v: np.asarray(v if v else [0], np.uint32)  # noqa: E731
    return np.asarray(off, np.int64), pad(l), pad(r)


def _collapse_call(fn, left_blocks, right_blocks, nh, with_filtered):
    n = len(left_blocks)
    lo, ll, lr = _blocks_csr(left_blocks)
    ro, rl, rr = _blocks_csr(right_blocks)
    nh = np.ascontiguousarray(nh, np.int32)
    up, um = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.float64)
    cm, nf = np.zeros(1, np.float64), np.zeros(1, np.int32)
    args = [n, lo, ll, lr, ro, rl, rr, nh, up, um, cm] + ([nf] if with_filtered else [])
    k = fn(*args)
    if k < 0:
        raise ValueError("collapse: malformed cluster")
    return up[:k].copy(), um[:k].copy(), float(cm[0]), int(nf[0])


class RefLib:
    """oracle/_ref/libstrawberry_ref.so -- the reference's own objects behind ref_shim.cpp."""

    def __init__(self, path=REF_LIB):
        if not os.path.exists(path):
            raise FileNotFoundError(path + " (run `make -C oracle ref` where /root/reference is mounted)")
        L = self.L = C.CDLL(path)
        L.ref_em_locus.argtypes = [C.c_int, C.c_int, _i32, _f64, _f64]
        L.ref_em_locus.restype = C.c_int
        L.ref_em_batch.argtypes = [C.c_int64, C.c_int64, _i64, _i64, _i64, _i32, _f64, _f64, _i32]
        L.ref_em_batch.restype = None
        L.ref_effective_len.argtypes = [C.c_int, _u32, C.c_int, _u32, C.c_int, C.c_int]
        L.ref_effective_len.restype = C.c_int
        L.ref_insert_pdf.argtypes = [C.c_double, C.c_double, C.c_int, _i32, C.c_int, C.c_int, _f64, _f64]
        L.ref_insert_pdf.restype = None
        L.ref_bin_weight.argtypes = [C.c_int, _u32, C.c_int, _u32, C.c_int, C.c_int, C.c_double, C.c_double,
                                     C.c_int, _i32]
        L.ref_bin_weight.restype = C.c_double
        _u8 = _p(np.uint8, flags="C")
        L.ref_is_compatible.argtypes = [C.c_int, _i32, _u32, _i32, C.c_int, _i32, _u32, _i32]
        L.ref_is_compatible.restype = C.c_int
        L.ref_overlap_key.argtypes = [C.c_int, _i32, _u32, _i32, C.c_int, _u32, _u32, _u8]
        L.ref_overlap_key.restype = None
        L.ref_pairedhit_features.argtypes = [C.c_int, _u32, _u32, C.c_int, _u32, _u32, _i32, _u32, _u32]
        L.ref_pairedhit_features.restype = C.c_int
        L.ref_kmer_stats.argtypes = [C.c_char_p, C.c_int, _f64]
        L.ref_kmer_stats.restype = None
        if hasattr(L, "ref_cluster_from_records"):
            L.ref_cluster_from_records.argtypes = [C.c_int, _p(np.uint64, flags="C"), _i64, _u32, _u32, _u32, _u8, _i32, _i32, _i32,
                                                   _f64, _f64, _i32]
            L.ref_cluster_from_records.restype = C.c_int
        if hasattr(L, "ref_collapse_cluster"):
            L.ref_collapse_cluster.argtypes = [C.c_int, _i64, _u32, _u32, _i64, _u32, _u32, _i32, _i32, _f64, _f64]
            L.ref_collapse_cluster.restype = C.c_int

    def bam_decode(self, bam_path, n_records, n_ops, min_intron=20, max_intron=300000, unique_only=True, library=0):
        """Every record of a BAM file through the reference's own BAMHitFactory::getHitFromBuf (src/read.cpp:480-715)
        with the option globals set as the command line would; n_records / n_ops: array capacities (records, CIGAR
        operations in all).  -> dict of per-record arrays (file order), the kept CIGARs and readhit_2_genomicFeats'
        features as CSR, and `single_end` (the global SINGLE_END_EXP afterwards)."""
        L = self.L
        if not hasattr(L, "ref_bam_decode"):
            raise RuntimeError("oracle/_ref/libstrawberry_ref.so predates ref_bam_decode: `make -C oracle ref`")
        _u8 = _p(np.uint8, flags="C")
        _u64 = _p(np.uint64, flags="C")
        L.ref_bam_decode.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, _u8, _u64, _i32, _u32, _u32, _u8,
                                     _u8, _u32, _i32, _i32, _u32, _f64, _i32, _i64, _u8, _u32, _i64, _u8, _u32, _u32, _i32]
        L.ref_bam_decode.restype = C.c_int
        n, m = max(int(n_records), 1), max(int(n_ops), 1)
        z = dict(accepted=np.zeros(n, np.uint8), read_id=np.zeros(n, np.uint64), ref=np.zeros(n, np.int32), left=np.zeros(n, np.uint32),
                 right=np.zeros(n, np.uint32), strand=np.zeros(n, np.uint8), partner_same_ref=np.zeros(n, np.uint8),
                 partner_pos=np.zeros(n, np.uint32), nm=np.zeros(n, np.int32), nh=np.zeros(n, np.int32), flag_bits=np.zeros(n, np.uint32),
                 mass=np.zeros(n, np.float64), read_len=np.zeros(n, np.int32), cig_off=np.zeros(n + 1, np.int64), cig_type=np.zeros(m, np.uint8),
                 cig_len=np.zeros(m, np.uint32), feat_off=np.zeros(n + 1, np.int64), feat_code=np.zeros(m, np.uint8),
                 feat_left=np.zeros(m, np.uint32), feat_right=np.zeros(m, np.uint32))
        se = np.zeros(1, np.int32)
        k = L.ref_bam_decode(str(bam_path).encode(), int(min_intron), int(max_intron), int(bool(unique_only)), int(library), n, m,
                             z["accepted"], z["read_id"], z["ref"], z["left"], z["right"], z["strand"], z["partner_same_ref"],
                             z["partner_pos"], z["nm"], z["nh"], z["flag_bits"], z["mass"], z["read_len"], z["cig_off"], z["cig_type"],
                             z["cig_len"], z["feat_off"], z["feat_code"], z["feat_left"], z["feat_right"], se)
        if k < 0:
            raise RuntimeError("ref_bam_decode: capacities too small")
        for key in list(z):
            if key in ("cig_off", "feat_off"):
                z[key] = z[key][:k + 1].copy()
            elif key not in ("cig_type", "cig_len", "feat_code", "feat_left", "feat_right"):
                z[key] = z[key][:k].copy()
        for key in ("cig_type", "cig_len"):
            z[key] = z[key][:z["cig_off"][-1]].copy()
        for key in ("feat_code", "feat_left", "feat_right"):
            z[key] = z[key][:z["feat_off"][-1]].copy()
        z["single_end"] = int(se[0])
        z["n"] = k
        return z

    def cluster_from_records(self, read_id, blocks, partner_pos, flags, nh):
        """Records -> the reference's HitCluster::addOpenHit (mate pairing) -> collapseAndFilterHits.
        -> (left_rec, right_rec, uniq_mass, cluster_mass, n_hits_before_collapse)"""
        off, bl, br = _blocks_csr(blocks)
        n = len(blocks)
        lr, rr, m = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.float64)
        cm, nhits = np.zeros(1, np.float64), np.zeros(1, np.int32)
        k = self.L.ref_cluster_from_records(n, np.ascontiguousarray(read_id, np.uint64), off, bl, br,
                                            np.ascontiguousarray(partner_pos if n else [0], np.uint32),
                                            np.ascontiguousarray(flags if n else [0], np.uint8), np.ascontiguousarray(nh if n else [1], np.int32),
                                            lr, rr, m, cm, nhits)
        return lr[:k].copy(), rr[:k].copy(), m[:k].copy(), float(cm[0]), int(nhits[0])

    def collapse_cluster(self, left_blocks, right_blocks, nh):
        """The reference's own HitCluster (addOpenHit for every read, then collapseAndFilterHits).
        -> (uniq_pair, uniq_mass, cluster_mass, None)"""
        up, um, cm, _ = _collapse_call(self.L.ref_collapse_cluster, left_blocks, right_blocks, nh, with_filtered=False)
        return up, um, cm, None

    def kmer_stats(self, seq):
        """(gc, entropy, flags) by the reference's own Kmer<string> templates; len(seq) > 40."""
        assert len(seq) > 40, "the reference asserts w < len (kmer.h:82)"
        out = np.zeros(6)
        self.L.ref_kmer_stats(bytes(seq), len(seq), out)
        return out[0], out[1], int(out[2]) | int(out[3]) << 1 | int(out[4]) << 2 | int(out[5]) << 3

    def is_compatible(self, code, left, right, exon_left, exon_right):
        """Contig::is_compatible on Contigs built from flat features; the isoform from its exons."""
        rc = np.ascontiguousarray(code, np.int32)
        rl = np.ascontiguousarray(left, np.uint32)
        rn = np.ascontiguousarray(np.asarray(right, np.int64) - np.asarray(left, np.int64) + 1, np.int32)
        ic, il, iln = [], [], []
        for k in range(len(exon_left)):
            if k:
                ic.append(1); il.append(exon_right[k - 1] + 1); iln.append(exon_left[k] - exon_right[k - 1] - 1)
            ic.append(0); il.append(exon_left[k]); iln.append(exon_right[k] - exon_left[k] + 1)
        return bool(self.L.ref_is_compatible(len(rc), rc, rl, rn, len(ic), np.ascontiguousarray(ic, np.int32),
                                             np.ascontiguousarray(il, np.uint32), np.ascontiguousarray(iln, np.int32)))

    def pairedhit_features(self, left_blocks, right_blocks):
        """Contig(PairedHit) from the mates' aligned blocks -> (code, left, right) lists or None."""
        def arr(blocks, k):
            return np.ascontiguousarray([b[k] for b in blocks] or [0], np.uint32)
        cap = 2 * (len(left_blocks) + len(right_blocks)) + 2
        oc, ol, orr = np.zeros(cap, np.int32), np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
        n = self.L.ref_pairedhit_features(len(left_blocks), arr(left_blocks, 0), arr(left_blocks, 1), len(right_blocks),
                                          arr(right_blocks, 0), arr(right_blocks, 1), oc, ol, orr)
        return None if n == 0 else (oc[:n].tolist(), ol[:n].tolist(), orr[:n].tolist())

    def overlap_key(self, code, left, right, seg_left, seg_right):
        rc = np.ascontiguousarray(code, np.int32)
        rl = np.ascontiguousarray(left, np.uint32)
        rn = np.ascontiguousarray(np.asarray(right, np.int64) - np.asarray(left, np.int64) + 1, np.int32)
        out = np.zeros(max(1, len(seg_left)), np.uint8)
        self.L.ref_overlap_key(len(rc), rc, rl, rn, len(seg_left), np.ascontiguousarray(seg_left, np.uint32),
                               np.ascontiguousarray(seg_right, np.uint32), out)
        return out[:len(seg_left)]

    def em_locus(self, count, F):
        """-> (theta, init_ok, run_ok) from the reference's EmSolver."""
        F = np.ascontiguousarray(F, np.float64)
        nrow, niso = F.shape
        count = np.ascontiguousarray(count, np.int32)
        theta = np.zeros(niso, np.float64)
        fl = self.L.ref_em_locus(nrow, niso, count, F.reshape(-1), theta)
        return theta, bool(fl & 1), bool(fl & 2)

    def em_batch(self, row_off, iso_off, f_off, count, F, threads=1):
        row_off, iso_off, f_off, count, F = _csr(row_off, iso_off, f_off, count, F)
        n = len(row_off) - 1
        theta = np.zeros(int(iso_off[-1]), np.float64)
        flags = np.zeros(n, np.int32)
        _run_threads(
            lambda lo, hi: self.L.ref_em_batch(lo, hi, row_off, iso_off, f_off, count, F, theta, flags),
            n, threads)
        return theta, flags

    def effective_len(self, seg_lens, implicit_idx, fl, rl):
        s = np.ascontiguousarray(seg_lens, np.uint32)
        m = np.ascontiguousarray(implicit_idx, np.uint32)
        nimp = len(m)
        if nimp == 0:
            m = np.zeros(1, np.uint32)
        return self.L.ref_effective_len(len(s), s, nimp, m, int(fl), int(rl))

    def insert_pdf(self, mean, sd, frag_lens, fl_lo, fl_hi):
        fl = np.zeros(1, np.int32) if frag_lens is None else np.ascontiguousarray(frag_lens, np.int32)
        n = 0 if frag_lens is None else len(fl)
        out = np.zeros(fl_hi - fl_lo + 1)
        info = np.zeros(4)
        self.L.ref_insert_pdf(float(mean), float(sd), n, fl, int(fl_lo), int(fl_hi), out, info)
        return out, info

    def bin_weight(self, seg_lens, implicit_idx, iso_len, rl, mean, sd, frag_lens=None):
        s = np.ascontiguousarray(seg_lens, np.uint32)
        m = np.ascontiguousarray(implicit_idx, np.uint32)
        nimp = len(m)
        if nimp == 0:
            m = np.zeros(1, np.uint32)
        fl = np.zeros(1, np.int32) if frag_lens is None else np.ascontiguousarray(frag_lens, np.int32)
        n = 0 if frag_lens is None else len(fl)
        return self.L.ref_bin_weight(len(s), s, nimp, m, int(iso_len), int(rl), float(mean), float(sd), n, fl)
