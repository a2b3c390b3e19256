"""Records -> theta in one resident pass: the BAM alignment records of the chain sample in HBM, then

    sbgpu_bam_decode_device      BAMHitFactory::getHitFromBuf            /root/reference/src/read.cpp:480-715
    sbgpu_assign_reads_device    Sample::nextClusterRefDemand            /root/reference/src/alignments.cpp:1145-1187
    sbgpu_pair_mates_device      HitCluster::addOpenHit / addHit         /root/reference/src/alignments.cpp:423-655
    sbgpu_collapse_pairs_device  HitCluster::collapseAndFilterHits       /root/reference/src/alignments.cpp:656-703
    sbgpu_quantify_device        Sample::quantifyCluster (bins, weights, EM)  alignments.cpp:1510-1546

every stage handing its device arrays to the next; only per-cluster offsets and theta come back to the host.

The record stream is synthetic (bench.py: `data: synthetic`): every read pair behind the chain sample's unique hits
(strawberry_amd/chain.py::DeviceSample) gets its two BAM records -- the uncompressed records behind the BAM header, as
samtools' bam_read1 sees them after BGZF inflate -- packed on the device with torch index arithmetic (4e8 records are
~70 GB that never exist in host memory), sorted by position like a coordinate-sorted BAM.  Plumbing, not product: the
library never sees how the bytes were made.
"""
import ctypes as C
import time

import numpy as np

from . import _lib
from . import bam
from . import exonbin as eb
from .chain import ChainQuantifier

_FIXED = 36          # block_size + the 32 bytes of fixed fields behind it
_NAME = 11           # "r%09d" + NUL


def pack_bam_records(torch, s, read_len=75, chunk=1 << 21):
    """DeviceSample -> (bytes uint8 [total], rec_off int64 [n_records + 1]) on the sample's device.

    A unique hit of mass m stands for m read pairs: 2 m records (flag 99 at the left mate's start, 147 at the right
    mate's), named r%09d by pair, CIGAR = the mate's blocks as M with N between, NH:i:1, XS:A:+, 75 bases of sequence and
    qualities (never looked at by the decoder, but they are what a record is mostly made of); records in (position, left
    before right) order, stable."""
    dev = s.feat_off.device
    H = s.n_hits
    code = s.feat_code
    gap_idx = torch.nonzero(code == eb.GAP).reshape(-1)
    assert int(gap_idx.numel()) == H, "every hit of the chain sample holds exactly one GAP (paired-end hits)"
    f0 = s.feat_off[:-1]
    nbl = (gap_idx - f0 + 1) // 2                          # MATCH blocks of the left mate (M, I, M, ... : 2 nb - 1 features)
    nbr = (s.feat_off[1:] - gap_idx) // 2
    fleft, fright = s.feat_left.to(torch.int64), s.feat_right.to(torch.int64)
    cnt = s.mass.to(torch.int64)
    P = int(cnt.sum().item())
    pair_hit = torch.repeat_interleave(torch.arange(H, device=dev), cnt)
    lpos, rpos = fleft[f0][pair_hit], fleft[gap_idx + 1][pair_hit]
    # the fragment's last base (the right mate's last block's end) for TLEN
    frag_end = fright[s.feat_off[1:] - 1][pair_hit]
    n = 2 * P
    rec_pos = torch.stack([lpos, rpos], 1).reshape(-1)
    order = torch.argsort(rec_pos * 2 + torch.tensor([0, 1], device=dev).repeat(P), stable=True)
    del rec_pos
    rec_pair = order >> 1
    rec_right = (order & 1).to(torch.bool)
    del order
    rec_hit = pair_hit[rec_pair]
    nb = torch.where(rec_right, nbr[rec_hit], nbl[rec_hit])
    first_feat = torch.where(rec_right, gap_idx[rec_hit] + 1, f0[rec_hit])
    tail = (read_len + 1) // 2 + read_len + 8               # sequence, qualities, NH:i:1 (4 bytes as NH C 1), XS:A:+ (4 bytes)
    rec_len = _FIXED + _NAME + 4 * (2 * nb - 1) + tail
    rec_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(rec_len, 0, out=rec_off[1:])
    total = int(rec_off[-1].item())
    buf = torch.empty(total, dtype=torch.uint8, device=dev)
    max_nb = int(nb.max().item()) if n else 1

    def le32(rows, col, v):
        for k in range(4):
            rows[:, col + k] = ((v >> (8 * k)) & 0xFF).to(torch.uint8)

    templates = {}
    for b in range(1, max_nb + 1):
        L = _FIXED + _NAME + 4 * (2 * b - 1) + tail
        t = np.zeros(L, np.uint8)
        t[0:4] = np.frombuffer(np.int32(L - 4).tobytes(), np.uint8)
        t[12] = _NAME
        t[13] = 255                                        # MAPQ
        t[20:24] = np.frombuffer(np.int32(read_len).tobytes(), np.uint8)
        t[36] = ord("r")
        c = _FIXED + _NAME + 4 * (2 * b - 1)
        t[c:c + (read_len + 1) // 2] = 0x11                # AAAA...
        c += (read_len + 1) // 2
        t[c:c + read_len] = 40
        c += read_len
        t[c:c + 8] = np.frombuffer(b"NHC\x01XSA+", np.uint8)
        templates[b] = torch.from_numpy(t).to(dev)
    for a in range(0, n, chunk):
        z = min(n, a + chunk)
        nb_c = nb[a:z]
        for b in range(1, max_nb + 1):
            sel = torch.nonzero(nb_c == b).reshape(-1) + a
            m = int(sel.numel())
            if m == 0:
                continue
            L = int(templates[b].numel())
            rows = templates[b].repeat(m, 1)
            right = rec_right[sel]
            pr = rec_pair[sel]
            ff = first_feat[sel]
            le32(rows, 8, fleft[ff] - 1)                                                   # pos (0-based)
            le32(rows, 16, torch.where(right, 147, 99).to(torch.int64) * 65536 + (2 * b - 1))   # flag << 16 | n_cigar_op
            le32(rows, 28, torch.where(right, lpos[pr], rpos[pr]) - 1)                     # next_pos
            tlen = frag_end[pr] - lpos[pr] + 1
            le32(rows, 32, torch.where(right, -tlen, tlen))
            d = pr.clone()
            for k in range(9):                                                             # r%09d
                rows[:, 36 + 9 - k] = (d % 10 + 48).to(torch.uint8)
                d = d // 10
            for k in range(b):                                                             # M (N M)*
                le32(rows, _FIXED + _NAME + 8 * k, (fright[ff + 2 * k] - fleft[ff + 2 * k] + 1) * 16)
                if k + 1 < b:
                    le32(rows, _FIXED + _NAME + 8 * k + 4, (fleft[ff + 2 * k + 2] - fright[ff + 2 * k] - 1) * 16 + 3)
            idx = rec_off[sel].unsqueeze(1) + torch.arange(L, device=dev).unsqueeze(0)
            buf[idx.reshape(-1)] = rows.reshape(-1)
            del rows, idx
    return buf, rec_off


class FrontQuantifier(ChainQuantifier):
    """The chain sample's alignment records in HBM -> theta.  step() runs the five stages; stage_wall_ms keeps the last
    step's per-stage wall times (every stage ends synchronised: its totals decide the next one's allocations)."""

    STAGES = ("bam_decode", "assign_reads", "pair_mates", "collapse_pairs", "quantify")

    def __init__(self, ctx, n_loci=60000, n_frags=2e8, seed=31, read_len=75, loci_subset=None, **resident_kw):
        """loci_subset = (rank, world): this rank's loci of ONE sample (locus l on rank l mod world, as the chain shards) --
        its records only; the clusters, like the reference's, are the shard's own gene models.  resident_kw: ChainQuantifier's
        resident / empirical / comm / min_isoform_frac (the last stage is then sbgpu_quantify_resident: records -> TPM)."""
        super().__init__(ctx, n_loci=n_loci, n_frags=n_frags, seed=seed, read_len=read_len, loci_subset=loci_subset, pin=True, **resident_kw)
        torch = self.torch
        t = time.perf_counter()
        self.d_bytes, self.d_rec_off = pack_bam_records(torch, self.sample, read_len)
        torch.cuda.synchronize(self.dev)
        self.pack_s = time.perf_counter() - t
        self.n_records, self.n_bytes = int(self.d_rec_off.numel()) - 1, int(self.d_bytes.numel())
        a = self.annot
        # the clusters of quant mode: one per gene model, [first exon's start, last exon's end] (addRef2Cluster), strand +
        first_exon = a.exon_off[a.iso_off[:-1]]
        lefts = np.minimum.reduceat(a.exon_left, a.exon_off[:-1])       # per isoform
        rights = np.maximum.reduceat(a.exon_right, a.exon_off[:-1])
        self._c_left = np.ascontiguousarray(np.minimum.reduceat(lefts, a.iso_off[:-1]), np.uint32)
        self._c_right = np.ascontiguousarray(np.maximum.reduceat(rights, a.iso_off[:-1]), np.uint32)
        del first_exon
        self._c_ref = np.zeros(self.n_loci, np.int32)
        self._c_strand = np.ones(self.n_loci, np.uint8)
        self._clusters = _lib.sbgpu_clusters_t(self.n_loci, self._c_ref.ctypes.data, self._c_left.ctypes.data, self._c_right.ctypes.data,
                                               self._c_strand.ctypes.data)
        self._opts = bam.BamOptions(n_ref=1).c()
        self._read_off = np.zeros(self.n_loci + 1, np.int64)
        self.stage_wall_ms = {}
        self.counts = {}

    def step(self, keep=False):
        torch, L, ctx = self.torch, self.ctx.L, self.ctx
        sync = lambda: torch.cuda.synchronize(self.dev)  # noqa: E731
        ms = {}
        sync()
        t = time.perf_counter()
        hb = C.c_void_p()
        _lib.check(L.sbgpu_bam_decode_device(ctx.h, self.d_bytes.data_ptr(), self.n_bytes, self.d_rec_off.data_ptr(), self.n_records,
                                             C.byref(self._opts), None, C.byref(hb)), "sbgpu_bam_decode_device")
        rs = _lib.sbgpu_reads_t()
        d_ref, d_left, d_right = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.check(L.sbgpu_bamreads_reads(hb, C.byref(rs), C.byref(d_ref), C.byref(d_left), C.byref(d_right)), "sbgpu_bamreads_reads")
        sync()
        t1 = time.perf_counter()
        ms["bam_decode"] = (t1 - t) * 1e3
        n_reads = int(rs.n_reads)
        d_cluster = torch.empty(max(n_reads, 1), dtype=torch.int32, device=self.dev)
        _lib.check(L.sbgpu_assign_reads_device(ctx.h, C.byref(self._clusters), n_reads, d_ref, d_left, d_right, rs.flags, d_cluster.data_ptr(),
                                               self._read_off.ctypes.data, None), "sbgpu_assign_reads_device")
        sync()
        t2 = time.perf_counter()
        ms["assign_reads"] = (t2 - t1) * 1e3
        hm = C.c_void_p()
        _lib.check(L.sbgpu_pair_mates_device(ctx.h, self.n_loci, C.byref(rs), self._read_off.ctypes.data, None, C.byref(hm)), "sbgpu_pair_mates_device")
        dp, poff = _lib.sbgpu_pairs_t(), C.c_void_p()
        _lib.check(L.sbgpu_matepairs_pairs(hm, C.byref(dp), C.byref(poff)), "sbgpu_matepairs_pairs")
        sync()
        t3 = time.perf_counter()
        ms["pair_mates"] = (t3 - t2) * 1e3
        del d_cluster
        L.sbgpu_bamreads_destroy(hb)           # (the pairs hold what they need of the reads)
        hu = C.c_void_p()
        _lib.check(L.sbgpu_collapse_pairs_device(ctx.h, self.n_loci, C.byref(dp), poff, None, C.byref(hu)), "sbgpu_collapse_pairs_device")
        dh = _lib.sbgpu_hits_t()
        d_mass, hoff = C.c_void_p(), C.c_void_p()
        _lib.check(L.sbgpu_uniq_dev_hits(hu, C.byref(dh), C.byref(d_mass), C.byref(hoff)), "sbgpu_uniq_dev_hits")
        sync()
        t4 = time.perf_counter()
        ms["collapse_pairs"] = (t4 - t3) * 1e3
        self.front_hit_off = np.ctypeslib.as_array(C.cast(hoff, C.POINTER(C.c_int64)), shape=(self.n_loci + 1,)).copy()
        L.sbgpu_matepairs_destroy(hm)
        h = C.c_void_p()
        if self.resident:
            ui = (C.c_int64 * 8)()
            _lib.check(L.sbgpu_uniq_dev_info(hu, ui), "sbgpu_uniq_dev_info")
            self._resident_call(L, dh, d_mass, hoff, int(ui[4]), h)     # ui[4]: sum over the clusters of (int) weighted_mass()
        else:
            _lib.check(L.sbgpu_quantify_device(ctx.h, C.byref(self._an), C.byref(dh), d_mass, hoff, C.byref(self._ins), self.read_len, 0,
                                               self.theta.ctypes.data, self.status.ctypes.data, self.iters.ctypes.data, C.byref(h)),
                       "sbgpu_quantify_device")
        sync()
        ms["quantify"] = (time.perf_counter() - t4) * 1e3
        if not self.counts:
            pi, ui = (C.c_int64 * 8)(), (C.c_int64 * 8)()
            # (the handles are gone for the pairs; the unique hits' handle still lives)
            _lib.check(L.sbgpu_uniq_dev_info(hu, ui), "sbgpu_uniq_dev_info")
            self.counts = {"records": self.n_records, "accepted_records": n_reads, "unique_hits": int(ui[0]), "features": int(ui[1]),
                           "pairs_dropped_by_the_span_filter": int(ui[2]), "mapped_reads": int(ui[4])}
            info = (C.c_int64 * 8)()
            _lib.check(L.sbgpu_bins_info(h, info), "sbgpu_bins_info")
            self.info = {"n_bins": int(info[2]), "n_elem": int(info[3]), "n_pairs": int(info[4]), "hits_in_bins": int(info[6])}
        L.sbgpu_bins_destroy(h)
        L.sbgpu_uniq_dev_destroy(hu)
        self.stage_wall_ms = ms

    # ---- the same pass for a caller that holds the records in HOST memory: sbgpu_front_stream_* (chunks, bounded footprint)
    def to_host(self, chunk_bytes, pinned=True, ref_id=0):
        """Bring the packed record stream to host memory (page-locked when `pinned` and the box allows it) and cut it into chunks
        of whole records of at most chunk_bytes: what a driver that inflates BGZF blocks holds -- the bytes, and the records'
        offsets noted on the way.  The device copy is released.  ref_id: the reference this sample's records (and their mates)
        lie on -- several samples on references 0, 1, ... pushed one after the other are ONE coordinate-sorted stream
        (stream_parts).  -> dict(bytes moved, pinned or not, chunks)."""
        torch = self.torch
        n = self.n_bytes
        if ref_id:
            at = self.d_rec_off[:-1]
            self.d_bytes[at + 4] = int(ref_id)       # refID (the template's is 0; ids below 256: one byte)
            self.d_bytes[at + 24] = int(ref_id)      # next_refID
            self._c_ref[:] = int(ref_id)
        rec_off = self.d_rec_off.cpu().numpy()
        note = None
        try:
            host = torch.empty(n, dtype=torch.uint8, pin_memory=bool(pinned))
        except RuntimeError as e:       # (not enough lockable memory)
            host, pinned, note = torch.empty(n, dtype=torch.uint8), False, "page-locked allocation failed: %s" % str(e)[:80]
        step = 1 << 30
        for a in range(0, n, step):
            host[a:a + step].copy_(self.d_bytes[a:a + step])
        torch.cuda.synchronize(self.dev)
        self.d_bytes = self.d_rec_off = None
        torch.cuda.empty_cache()
        self.h_bytes, self.h_pinned, self.h_rec_off = host, bool(pinned), rec_off
        self.cut(chunk_bytes)
        return {"bytes": n, "pinned": bool(pinned), "chunks": len(self.h_chunks), "note": note}

    def cut(self, chunk_bytes):
        """The host copy's chunks: as many whole records as fit chunk_bytes, with the records' offsets inside the chunk."""
        torch = self.torch
        rec_off, cuts, i = self.h_rec_off, [], 0
        while i < self.n_records:
            j = int(np.searchsorted(rec_off, rec_off[i] + chunk_bytes, side="right")) - 1
            if j <= i:
                raise ValueError("a record is longer than a chunk")
            # (the offsets as the inflating driver notes them: inside the chunk, in memory of the same kind as the bytes)
            off = torch.from_numpy(rec_off[i:j + 1] - rec_off[i])
            if self.h_pinned:
                try:
                    off = off.pin_memory()
                except RuntimeError:
                    pass
            cuts.append((i, j, int(rec_off[i]), int(rec_off[j]), off))
            i = j
        self.h_chunks, self.chunk_bytes = cuts, int(chunk_bytes)

    def stream_step(self):
        """One pass records (host) -> TPM through sbgpu_front_stream_begin / push / end; the results land where step() puts
        them (resident mode).  -> the stream's info (sbgpu_front_stream_info) as a dict."""
        L, ctx = self.ctx.L, self.ctx
        fs = C.c_void_p()
        _lib.check(L.sbgpu_front_stream_begin(ctx.h, C.byref(self._clusters), C.byref(self._opts), self.chunk_bytes, C.byref(fs)), "sbgpu_front_stream_begin")
        try:
            base = self.h_bytes.data_ptr()
            for (i, j, b0, b1, off) in self.h_chunks:
                _lib.check(L.sbgpu_front_stream_push(fs, base + b0, b1 - b0, off.data_ptr(), j - i), "sbgpu_front_stream_push")
            h = C.c_void_p()
            _lib.check(L.sbgpu_front_stream_end(fs, C.byref(self._an), None if self.empirical else C.byref(self._ins), self.read_len, 0,
                                                C.byref(self._par), self.comm.h if self.comm is not None else None, C.byref(self._used),
                                                C.byref(self._out), C.byref(h)), "sbgpu_front_stream_end")
            u = self._used
            self.law = {"mean": u.mean, "sd": u.sd, "use_emp": int(u.use_emp), "start_offset": int(u.start_offset),
                        "end_offset": int(u.end_offset), "total_reads": int(u.total_reads)}
            if u.use_emp:
                self.law["emp_hist"] = np.ctypeslib.as_array(u.emp_hist, shape=(u.end_offset - u.start_offset + 1,)).copy()
            self.total_fpkm, self.total_mapped_reads = float(self._out.total_fpkm), int(self._out.total_mapped_reads)
            hoff = C.c_void_p()
            _lib.check(L.sbgpu_front_stream_hits(fs, None, None, C.byref(hoff)), "sbgpu_front_stream_hits")
            self.front_hit_off = np.ctypeslib.as_array(C.cast(hoff, C.POINTER(C.c_int64)), shape=(self.n_loci + 1,)).copy()
            info = (C.c_int64 * 16)()
            _lib.check(L.sbgpu_front_stream_info(fs, info), "sbgpu_front_stream_info")
            L.sbgpu_bins_destroy(h)
        finally:
            L.sbgpu_front_stream_destroy(fs)
        keys = ("records", "accepted_records", "pairs", "unique_hits", "features", "pairs_dropped_by_the_span_filter", "mapped_reads", "chunks",
                "clusters_finished", "most_bytes_carried", "records_decoded_twice", "least_free_device_bytes", "chunk_bytes", "ended",
                "free_device_bytes_at_begin")
        self.stream_info = {k: int(info[i]) for i, k in enumerate(keys)}
        self.counts = dict(self.counts or {}, pairs_dropped_by_the_span_filter=self.stream_info["pairs_dropped_by_the_span_filter"],
                           unique_hits=self.stream_info["unique_hits"])
        return self.stream_info

    def check_filtered_loci(self, oracle, max_pairs=3_000_000):
        """After a resident step(): the loci whose unique hits differ from the sample's -- the reference's span filter
        (HitCluster::collapseAndFilterHits, alignments.cpp:666-682) dropped pairs there, which compare_with_chain can only count --
        against the ORACLE: the sample's read pairs of each such cluster go through oracle/collapse_oracle.c (pinned on the
        reference's own HitCluster), its unique hits must be as many as the front's, and the chain on exactly those hits -- the
        affected loci alone, under the law and the mapped-read total the pass ended with -- must give the front's theta, status,
        iterations, FPKM, Frac and keep bit for bit.  Test infrastructure: `oracle` is oracle.OracleLib (bench.py's parity leg,
        tests).  -> dict."""
        from .binweight import InsertSize
        from .quantify import quantify_resident
        a, smp = self.annot, self.sample
        loci = np.nonzero(np.diff(self.front_hit_off) != np.diff(smp.locus_hit_off))[0]
        out = {"loci": int(len(loci)), "pairs_through_the_oracle": 0, "pairs_the_oracle_dropped": 0, "ok": True}
        if not len(loci):
            return out
        sub_loci, hit_locus, feats, masses, taken = [], [], [], [], []
        for l in loci.tolist():
            h0, h1 = int(smp.locus_hit_off[l]), int(smp.locus_hit_off[l + 1])
            foff = smp.feat_off[h0:h1 + 1].cpu().numpy()
            f0, f1 = int(foff[0]), int(foff[-1])
            code = smp.feat_code[f0:f1].cpu().numpy()
            fl, fr = smp.feat_left[f0:f1].cpu().numpy().view(np.uint32), smp.feat_right[f0:f1].cpu().numpy().view(np.uint32)
            mass = smp.mass[h0:h1].cpu().numpy().astype(np.int64)
            if out["pairs_through_the_oracle"] + int(mass.sum()) > max_pairs:
                out["skipped_for_size"] = out.get("skipped_for_size", 0) + 1
                continue
            lb, rb, owner = [], [], []
            for h in range(h1 - h0):
                s0, s1 = int(foff[h] - f0), int(foff[h + 1] - f0)
                gap = s0 + int(np.nonzero(code[s0:s1] == eb.GAP)[0][0])
                lm = [(int(fl[i]), int(fr[i])) for i in range(s0, gap) if code[i] == eb.MATCH]
                rm = [(int(fl[i]), int(fr[i])) for i in range(gap + 1, s1) if code[i] == eb.MATCH]
                m = int(mass[h])
                lb += [lm] * m
                rb += [rm] * m
                owner += [h] * m
            uniq_pair, uniq_mass, _, n_filtered = oracle.collapse_cluster(lb, rb, [1] * len(lb))
            out["pairs_through_the_oracle"] += len(lb)
            out["pairs_the_oracle_dropped"] += int(n_filtered)
            kept = [owner[int(p)] for p in uniq_pair]
            if len(kept) != int(self.front_hit_off[l + 1] - self.front_hit_off[l]):
                out["ok"] = False
                out.setdefault("hit_count_mismatch", []).append(int(l))
            j0 = int(a.iso_off[l])
            taken.append(int(l))
            sub_loci.append([[(int(a.exon_left[e]), int(a.exon_right[e])) for e in range(int(a.exon_off[j]), int(a.exon_off[j + 1]))]
                             for j in range(j0, int(a.iso_off[l + 1]))])
            for h, um in zip(kept, uniq_mass):
                s0, s1 = int(foff[h] - f0), int(foff[h + 1] - f0)
                hit_locus.append(len(sub_loci) - 1)
                feats.append((code[s0:s1].tolist(), fl[s0:s1].tolist(), fr[s0:s1].tolist()))
                masses.append(float(um))
        if not sub_loci:
            return out
        sub = eb.Annotation(sub_loci)
        hits = eb.Hits(hit_locus, feats, masses)
        law = InsertSize.from_hist(self.law["start_offset"], self.law["emp_hist"]) if self.law["use_emp"] else self.insert
        r = quantify_resident(sub, hits, law, self.read_len, self.total_mapped_reads, ctx=self.ctx)
        j = 0
        for used, l in enumerate(taken):
            n, g0 = int(a.iso_off[l + 1] - a.iso_off[l]), int(a.iso_off[l])
            same = all(np.array_equal(getattr(self, k)[g0:g0 + n], r[k][j:j + n]) for k in ("theta", "fpkm", "frac", "keep"))
            same = same and int(self.status[l]) == int(r["status"][used]) and int(self.iters[l]) == int(r["iters"][used])
            if not same:
                out["ok"] = False
                out.setdefault("theta_mismatch", []).append(int(l))
            j += n
        out["loci_checked"] = len(taken)
        return out

    @staticmethod
    def stream_parts(parts, empirical=True, comm=None, min_isoform_frac=0.0):
        """Several samples (FrontQuantifiers brought to_host with ref_id 0, 1, ...) pushed one after the other as ONE stream: a
        sample too large to be packed on the device in one piece (BASELINE config 5: 4e8 read pairs).  -> dict of results over
        all parts' loci (theta, fpkm, frac, tpm, keep, status, iters, law, totals, info)."""
        from .exonbin import Annotation
        p0 = parts[0]
        L, ctx = p0.ctx.L, p0.ctx
        annot = Annotation.concat([p.annot for p in parts])
        n_iso, n_loci = int(annot.iso_off[-1]), annot.n_loci
        c_ref = np.ascontiguousarray(np.concatenate([p._c_ref for p in parts]), np.int32)
        c_left = np.ascontiguousarray(np.concatenate([p._c_left for p in parts]), np.uint32)
        c_right = np.ascontiguousarray(np.concatenate([p._c_right for p in parts]), np.uint32)
        c_strand = np.ascontiguousarray(np.concatenate([p._c_strand for p in parts]), np.uint8)
        clusters = _lib.sbgpu_clusters_t(n_loci, c_ref.ctypes.data, c_left.ctypes.data, c_right.ctypes.data, c_strand.ctypes.data)
        opts = bam.BamOptions(n_ref=len(parts)).c()
        res = {k: np.zeros(n_iso + 1, np.float64) for k in ("theta", "fpkm", "frac", "tpm")}
        res["keep"] = np.zeros(n_iso + 1, np.int32)
        res["status"], res["iters"] = np.zeros(n_loci + 1, np.int32), np.zeros(n_loci + 1, np.int32)
        out = _lib.sbgpu_abundances_t()
        for k, v in res.items():
            setattr(out, k, v.ctypes.data)
        par = _lib.sbgpu_abundance_params_t(0, 0, 1, 0, 0.0, float(min_isoform_frac))
        used, an = _lib.sbgpu_insert_t(), annot._struct()
        fs, h = C.c_void_p(), C.c_void_p()
        _lib.check(L.sbgpu_front_stream_begin(ctx.h, C.byref(clusters), C.byref(opts), max(p.chunk_bytes for p in parts), C.byref(fs)), "sbgpu_front_stream_begin")
        try:
            for p in parts:
                base = p.h_bytes.data_ptr()
                for (i, j, b0, b1, off) in p.h_chunks:
                    _lib.check(L.sbgpu_front_stream_push(fs, base + b0, b1 - b0, off.data_ptr(), j - i), "sbgpu_front_stream_push")
            _lib.check(L.sbgpu_front_stream_end(fs, C.byref(an), None if empirical else C.byref(p0._ins), p0.read_len, 0, C.byref(par),
                                                comm.h if comm is not None else None, C.byref(used), C.byref(out), C.byref(h)), "sbgpu_front_stream_end")
            law = {"mean": used.mean, "sd": used.sd, "use_emp": int(used.use_emp), "start_offset": int(used.start_offset),
                   "end_offset": int(used.end_offset), "total_reads": int(used.total_reads)}
            info = (C.c_int64 * 16)()
            _lib.check(L.sbgpu_front_stream_info(fs, info), "sbgpu_front_stream_info")
            L.sbgpu_bins_destroy(h)
        finally:
            L.sbgpu_front_stream_destroy(fs)
        keys = ("records", "accepted_records", "pairs", "unique_hits", "features", "pairs_dropped_by_the_span_filter", "mapped_reads", "chunks",
                "clusters_finished", "most_bytes_carried", "records_decoded_twice", "least_free_device_bytes", "chunk_bytes", "ended",
                "free_device_bytes_at_begin")
        r = {k: (v[:n_iso] if k not in ("status", "iters") else v[:n_loci]) for k, v in res.items()}
        r.update({"law": law, "total_fpkm": float(out.total_fpkm), "total_mapped_reads": int(out.total_mapped_reads),
                  "info": {k: int(info[i]) for i, k in enumerate(keys)}, "annot": annot})
        return r

    def chain_step(self):
        """The same sample through the chain alone (its unique hits as DeviceSample made them): what step() must reproduce.
        (In -i mode; an empirical law is the whole sample's, span-filtered pairs included, so the two laws may differ.)"""
        ChainQuantifier.step(self)

    def compare_with_chain(self):
        """After a step(): run the chain on the sample's own unique hits and compare locus by locus.  The front end applies
        the reference's span filter (HitCluster::collapseAndFilterHits drops a pair whose mate's span is an outlier of its
        cluster's spans, alignments.cpp:666-682: a spliced mate in a cluster of thousands of unspliced ones), which the
        sample generator -- it draws unique hits directly -- does not: a locus that lost a pair to it is compared on its hit
        count only, every other locus must agree bit for bit.  Resident mode: the chain is given the law and the mapped-read
        total the records -> TPM pass ended with (an empirical law is the WHOLE sample's, all ranks', the filtered pairs left out)
        and runs without the exchange: theta, status, iterations, FPKM, Frac and keep are compared (TPM divides by all ranks'
        FPKM total).  -> dict; the front's results are put back in place."""
        names = ("theta", "status", "iters") + (("fpkm", "frac", "keep") if self.resident else ())
        front = {k: getattr(self, k).copy() for k in names + (("tpm",) if self.resident else ())}
        saved = None
        if self.resident:
            saved = (self.insert, self.empirical, self._ins, self.comm, self.mapped_override, dict(self.law), self.total_fpkm, self.total_mapped_reads)
            if self.law["use_emp"]:
                from .binweight import InsertSize
                self.set_law(InsertSize.from_hist(self.law["start_offset"], self.law["emp_hist"]))
            self.comm, self.mapped_override = None, self.total_mapped_reads
        self.chain_step()
        same_hits = np.diff(self.front_hit_off) == np.diff(self.hits.locus_hit_off)
        a = self.annot
        iso_same = np.repeat(same_hits, np.diff(a.iso_off))
        ok = True
        for k in names:
            n, m = (self.n_loci, same_hits) if k in ("status", "iters") else (self.n_iso, iso_same)
            ok = ok and bool(np.array_equal(front[k][:n][m], getattr(self, k)[:n][m]))
        lost = int((np.diff(self.hits.locus_hit_off) - np.diff(self.front_hit_off)).sum())
        out = {"loci_with_the_samples_hits": int(same_hits.sum()), "bitwise_equal_there": ok, "compared": list(names),
               "loci_that_lost_pairs_to_the_span_filter": int((~same_hits).sum()), "unique_hits_lost_there": lost,
               "pairs_dropped_by_the_span_filter": self.counts.get("pairs_dropped_by_the_span_filter"),
               "ok": bool(ok and lost >= 0 and lost <= max(self.counts.get("pairs_dropped_by_the_span_filter", 0), 0))}
        for k, v in front.items():
            getattr(self, k)[:] = v
        if saved is not None:
            self.insert, self.empirical, self._ins, self.comm, self.mapped_override, self.law, self.total_fpkm, self.total_mapped_reads = saved
        return out
