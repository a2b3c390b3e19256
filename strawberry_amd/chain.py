"""The fragments -> abundances chain on hits that are resident in HBM (sbgpu_quantify_device), and the synthetic
human-scale input of bench.py's chain workload.

The sample: `n_loci` DISTINCT gene models (synth.make_gene_models: 1-12 exons, 1-6 alternatively spliced isoforms,
every locus drawn on its own -- no model is laid out twice) and ~2e8 read pairs sampled from them ON THE DEVICE with
torch index arithmetic (2e8 fragments are ~10 GB of features that never exist in host memory): per locus a
log-normal share of the fragments, per isoform a Dirichlet-like expression level, fragment length N(250, 30), start
uniform along the transcript, mates of 75 bases mapped through the isoform's exon table, plus pairs that fit fewer
isoforms or none (a mate shifted by a few bases; an unspliced left mate that runs into the intron).  Hits come out
sorted by (locus, left end, right end) -- HitCluster::collapseAndFilterHits' order -- in the layout of sbgpu_hits_t.
"""
import ctypes as C

import numpy as np

from . import _lib
from . import exonbin as eb
from . import synth
from .binweight import InsertSize


class DeviceSample:
    """Annotation (host arrays, it is small) + hits on the device."""

    def __init__(self, torch, dev, n_loci=60000, n_frags=2e8, seed=31, read_len=75, mean=250.0, sd=30.0, noise=0.10,
                 loci_subset=None):
        """loci_subset: (rank, world) -- keep only this rank's share of the SAME sample's loci (strong scaling:
        every rank draws the same annotation and the same per-locus fragment counts, then generates its own loci's
        fragments only)."""
        loci = synth.make_gene_models(n_loci, seed=seed)
        rng = np.random.Generator(np.random.PCG64(seed ^ 0x5EED))
        # per-locus share of the fragments: log-normal (sigma 1), like C3's law
        w = rng.lognormal(0.0, 1.0, n_loci)
        per_locus = np.maximum(0, np.rint(w / w.sum() * n_frags)).astype(np.int64)
        self.world_loci = n_loci
        if loci_subset is not None:
            rank, world = loci_subset
            mine = np.arange(rank, n_loci, world)          # locus l belongs to rank l mod world (examples/quantify_fragments.cpp)
            loci = [loci[l] for l in mine]
            per_locus = per_locus[mine]
            n_loci = len(loci)
        self.annot = eb.Annotation(loci)
        a = self.annot
        n_iso, n_exon = int(a.iso_off[-1]), int(a.exon_off[-1])
        # ---- isoform tables on the host (small), then to the device
        ex_len = (a.exon_right.astype(np.int64) - a.exon_left.astype(np.int64) + 1)
        iso_len = np.add.reduceat(ex_len, a.exon_off[:-1]) if n_iso else np.zeros(0, np.int64)
        ex_iso = np.repeat(np.arange(n_iso), np.diff(a.exon_off))
        cum_incl = np.cumsum(ex_len) - np.repeat(np.concatenate([[0], np.cumsum(iso_len)[:-1]]), np.diff(a.exon_off))   # within the isoform, incl. this exon
        cum_excl = cum_incl - ex_len
        BIG = 1 << 20                                         # > any transcript length
        assert iso_len.max(initial=0) < BIG
        iso_locus = np.repeat(np.arange(n_loci), np.diff(a.iso_off))
        # expression of an isoform inside its locus: Gamma(0.8) shares; isoforms too short for a pair get none
        expr = rng.gamma(0.8, 1.0, n_iso) * (iso_len >= 2 * read_len + 1)
        tot = np.bincount(iso_locus, weights=expr, minlength=n_loci)
        per_locus = np.where(tot > 0, per_locus, 0)           # a locus of short isoforms only has no pairs
        share = np.cumsum(expr) - np.repeat(np.concatenate([[0], np.cumsum(tot)[:-1]]), np.diff(a.iso_off))
        # cumulative within the locus, last = 1.  (The two running sums above cancel to within 1e-11 only: a locus without
        # expression must not turn that residue into a huge number -- `key` below has to stay sorted.)
        share = np.clip(np.where(tot[iso_locus] > 0, share / np.maximum(tot[iso_locus], 1e-300), 0.0), 0.0, 1.0)
        last = np.zeros(n_iso, bool)
        last[a.iso_off[1:][np.diff(a.iso_off) > 0] - 1] = True
        share[last] = 1.0
        up = lambda x, dt=None: torch.from_numpy(np.ascontiguousarray(x if dt is None else x.astype(dt))).to(dev)  # noqa: E731
        g = torch.Generator(device=dev)
        g.manual_seed(seed * 7919 + (0 if loci_subset is None else 1 + loci_subset[0]))
        n = int(per_locus.sum())
        d_locus = torch.repeat_interleave(torch.arange(n_loci, device=dev, dtype=torch.int32), up(per_locus))
        # isoform of each fragment: first isoform of the locus whose cumulative share reaches u
        key = up(iso_locus.astype(np.float64) + share * (1.0 - 1e-9))     # sorted: locus + cumulative share in (0, 1)
        u = torch.rand(n, device=dev, generator=g, dtype=torch.float64) * (1.0 - 3e-9) + 1e-9   # in (0, 1 - 2e-9)
        d_iso = torch.searchsorted(key, d_locus.to(torch.float64) + u, right=False).to(torch.int64)
        del u, key
        d_L = up(iso_len)[d_iso]
        fl = torch.round(torch.randn(n, device=dev, generator=g) * sd + mean).to(torch.int64)
        fl = torch.minimum(torch.clamp(fl, min=2 * read_len + 1), d_L)
        st = torch.floor(torch.rand(n, device=dev, generator=g, dtype=torch.float64) * (d_L - fl + 1).to(torch.float64)).to(torch.int64)
        st = torch.minimum(st, d_L - fl)
        del d_L
        kind = torch.rand(n, device=dev, generator=g)
        shift = torch.where((kind >= noise / 2) & (kind < noise), torch.randint(1, 9, (n,), device=dev, generator=g), 0).to(torch.int64)
        unspliced = kind < noise / 2
        del kind
        # ---- transcript coordinates -> genomic blocks through the isoform's exon table
        d_cum_key = up(ex_iso.astype(np.int64) * BIG + cum_incl)          # sorted: (isoform, end of exon in transcript)
        d_ex_left, d_cum_excl, d_ex_len = up(a.exon_left, np.int64), up(cum_excl), up(ex_len)

        def exon_of(t):   # global exon index holding transcript base t of the fragment's isoform
            return torch.searchsorted(d_cum_key, d_iso * BIG + t, right=True)

        MAXB = 4

        def mate(t0, t1):
            """blocks of transcript interval [t0, t1): (count, left[MAXB], right[MAXB]) -- int64 [n] tensors"""
            e0, e1 = exon_of(t0), exon_of(t1 - 1)
            nb = (e1 - e0 + 1)
            ls, rs = [], []
            for k in range(MAXB):
                e = torch.minimum(e0 + k, e1)
                base = d_ex_left[e] - d_cum_excl[e]
                ls.append(base + torch.maximum(t0, d_cum_excl[e]))
                rs.append(base + torch.minimum(t1, d_cum_excl[e] + d_ex_len[e]) - 1)
            return nb, ls, rs

        nbl, ll, lr = mate(st, st + read_len)
        nbr, rl, rr = mate(st + fl - read_len, st + fl)
        del st, fl
        # unspliced left mate: one block of read_len bases from its start (runs into the intron when it was spliced)
        nbl = torch.where(unspliced, 1, nbl)
        lr[0] = torch.where(unspliced, ll[0] + read_len - 1, lr[0])
        rl = [x + shift for x in rl]
        rr = [x + shift for x in rr]
        del shift, unspliced
        ok = (nbl <= MAXB) & (nbr <= MAXB)
        left_end = lr[0]
        for k in range(1, MAXB):
            left_end = torch.where(nbl > k, lr[k], left_end)
        right_end = rr[0]
        for k in range(1, MAXB):
            right_end = torch.where(nbr > k, rr[k], right_end)
        ok &= rl[0] > left_end + 1                                       # a GAP of at least one base between the mates
        # ---- HitCluster's order: (left end, right end); the loci lie along the genome in locus order.  Equal fragments
        # are ONE unique hit whose mass is their number (HitCluster::collapseAndFilterHits merges them, alignments.cpp:685-696):
        # inside a (left, right) run the fragments are ordered by a hash of their blocks
        span = torch.where(ok, ll[0] * (1 << 31) + right_end, torch.iinfo(torch.int64).max)
        sig = torch.zeros(n, dtype=torch.int64, device=dev)
        for k in range(MAXB):
            for x, nbx in ((ll[k], nbl), (lr[k], nbl), (rl[k], nbr), (rr[k], nbr)):
                sig = (sig ^ torch.where(nbx > k, x, -1 - k)) * (-7046029254386353131) + 7145426229640650777   # (int64 arithmetic wraps)
                sig = sig ^ (sig >> 29)
        order = torch.argsort(sig)
        order = order[torch.argsort(span[order], stable=True)]
        n_ok = int(ok.sum().item())
        order = order[:n_ok]
        s_span, s_sig = span[order], sig[order]
        # one fragment per (left, right): the reference sorts its pairs by the two ends only (std::sort, src/read.cpp:917-923)
        # and merges equal NEIGHBOURS, so different fragments with equal ends would make its own output depend on its
        # sort's tie order (A B A: the second A becomes a unique hit of its own, which the bins' std::set then drops).
        # Of the fragments sharing their ends the first kind stays; its copies are its mass.
        first = torch.ones(n_ok, dtype=torch.bool, device=dev)
        first[1:] = s_span[1:] != s_span[:-1]
        pos = torch.arange(n_ok, device=dev)
        run_start = torch.cummax(torch.where(first, pos, 0), 0).values
        keep = s_sig == s_sig[run_start]
        run_id = torch.cumsum(first, 0) - 1
        mass = torch.bincount(run_id[keep], minlength=int(run_id[-1].item()) + 1 if n_ok else 0).to(torch.float32)
        del pos, run_start, keep
        order = order[first]
        self.n_pairs_drawn = n_ok
        n_ok = int(order.numel())
        del ok, right_end, span, sig, s_span, s_sig, first, run_id
        take = lambda x: x[order]  # noqa: E731
        d_locus, nbl, nbr, left_end = take(d_locus), take(nbl), take(nbr), take(left_end)
        ll, lr, rl, rr = [take(x) for x in ll], [take(x) for x in lr], [take(x) for x in rl], [take(x) for x in rr]
        del order, d_iso
        # ---- features: left mate (MATCH, INTRON between), GAP, right mate
        nf = 2 * nbl - 1 + 1 + 2 * nbr - 1
        feat_off = torch.zeros(n_ok + 1, dtype=torch.int64, device=dev)
        torch.cumsum(nf, 0, out=feat_off[1:])
        total = int(feat_off[-1].item())
        code = torch.zeros(total, dtype=torch.uint8, device=dev)
        fleft = torch.zeros(total, dtype=torch.int64, device=dev)
        fright = torch.zeros(total, dtype=torch.int64, device=dev)
        base = feat_off[:-1]

        def put(mask, pos, c, l, r):
            p = pos[mask]
            code[p] = c
            fleft[p] = l[mask]
            fright[p] = r[mask]

        every = torch.ones(n_ok, dtype=torch.bool, device=dev)
        for k in range(MAXB):
            m = nbl > k
            put(m, base + 2 * k, eb.MATCH, ll[k], lr[k])
            if k:
                put(m, base + 2 * k - 1, eb.INTRON, lr[k - 1] + 1, ll[k] - 1)
        gap_at = base + 2 * nbl - 1
        put(every, gap_at, eb.GAP, left_end + 1, rl[0] - 1)
        for k in range(MAXB):
            m = nbr > k
            put(m, gap_at + 1 + 2 * k, eb.MATCH, rl[k], rr[k])
            if k:
                put(m, gap_at + 2 * k, eb.INTRON, rr[k - 1] + 1, rl[k] - 1)
        del ll, lr, rl, rr, nbl, nbr, left_end, every, gap_at, base, nf
        self.n_hits, self.n_features = n_ok, total
        self.hit_locus = d_locus
        self.feat_off = feat_off
        self.feat_code = code
        self.feat_left = fleft.to(torch.int32)        # uint32 coordinates travel as int32 bit patterns (all < 2^31 here)
        self.feat_right = fright.to(torch.int32)
        del fleft, fright
        self.mass = mass
        self.n_fragments = int(mass.sum().item())     # read pairs behind the unique hits
        if n_ok > 1 and not bool((d_locus[1:] >= d_locus[:-1]).all().item()):
            raise AssertionError("DeviceSample: the hits are not grouped by locus")
        cnt = torch.bincount(d_locus.to(torch.int64), minlength=n_loci)
        self.locus_hit_off = np.concatenate([[0], np.cumsum(cnt.cpu().numpy())]).astype(np.int64)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    def struct(self):
        s = _lib.sbgpu_hits_t()
        s.n_hits = self.n_hits
        for k in ("hit_locus", "feat_off", "feat_code", "feat_left", "feat_right"):
            setattr(s, k, getattr(self, k).data_ptr())
        return s

    def host_hits(self, n_loci):
        """The hits of the first n_loci loci as host arrays (eb.Hits): what the oracle / the reference gets of the sample."""
        h1 = int(self.locus_hit_off[n_loci])
        f1 = int(self.feat_off[h1].item())
        return eb.Hits.from_arrays(self.hit_locus[:h1].cpu().numpy(), self.feat_off[:h1 + 1].cpu().numpy(),
                                   self.feat_code[:f1].cpu().numpy(), self.feat_left[:f1].cpu().numpy().view(np.uint32),
                                   self.feat_right[:f1].cpu().numpy().view(np.uint32), self.mass[:h1].cpu().numpy())


class ChainQuantifier:
    """step(): fragments (in HBM) -> compat / key words -> bins -> weights -> EM -> theta, one C-ABI call;
    then FPKM / TPM on the host arrays the call returns (the caller's own epilogue, as in the reference)."""

    def __init__(self, ctx, n_loci=60000, n_frags=2e8, seed=31, read_len=75, loci_subset=None, pin=True, resident=False,
                 empirical=False, comm=None, min_isoform_frac=0.0):
        """resident=True: step() is sbgpu_quantify_resident -- the chain with the reference's pass 1 in front (empirical=True: no
        insert-size law is given, the device builds it from the hits; Strawberry's default mode) and the FPKM / Frac / TPM
        epilogue behind it, the collectives over `comm` (dist.AbiComm / dist.HostComm; None: a world of one) inside the call;
        theta, FPKM, Frac, keep and TPM land in the object's host arrays, `law` holds the insert-size law that was used."""
        import torch
        self.torch, self.ctx = torch, ctx
        self.dev = torch.device("cuda", ctx.device)
        self.sample = DeviceSample(torch, self.dev, n_loci, n_frags, seed, read_len, loci_subset=loci_subset)
        self.annot, self.hits = self.sample.annot, self.sample
        self.insert = InsertSize(250.0, 30.0)
        self.read_len = read_len
        # n_hits: unique hits (what the kernels touch); n_frags: the read pairs they stand for (their masses)
        self.n_loci, self.n_hits, self.n_frags = self.annot.n_loci, self.hits.n_hits, self.hits.n_fragments
        self.n_iso = int(self.annot.iso_off[-1])
        self.theta = np.zeros(self.n_iso + 1)
        self.status = np.zeros(self.n_loci + 1, np.int32)
        self.iters = np.zeros(self.n_loci + 1, np.int32)
        self._an = self.annot._struct()
        self._ht = self.hits.struct()
        self._ins = self.insert._struct(read_len)
        self.info = None
        self.resident, self.empirical, self.comm = bool(resident), bool(empirical), comm
        if resident:
            self.fpkm, self.frac, self.tpm = np.zeros(self.n_iso + 1), np.zeros(self.n_iso + 1), np.zeros(self.n_iso + 1)
            self.keep = np.zeros(self.n_iso + 1, np.int32)
            self._par = _lib.sbgpu_abundance_params_t(0, 0, 1, 0, 0.0, float(min_isoform_frac))
            self._out = _lib.sbgpu_abundances_t()
            for k in ("theta", "fpkm", "frac", "tpm", "keep", "status", "iters"):
                setattr(self._out, k, getattr(self, k).ctypes.data)
            self._used = _lib.sbgpu_insert_t()
            self.law = None
            self.mapped_override = None     # tests: a mapped-read total other than this object's own
        # the annotation is read once and the reads stream past it (Strawberry.cpp:245-275, :359): kept resident
        self.pinned = bool(pin)
        if pin:
            _lib.check(ctx.L.sbgpu_annotation_pin(ctx.h, C.byref(self._an)), "sbgpu_annotation_pin")

    def step(self, keep=False):
        """keep=True: -> the LocusBins of this step's handle, exported to the host (tests compare them with the oracle's;
        the weights of the device entry stay in HBM -- sbgpu_quantify_host on the same hits returns them)."""
        h = C.c_void_p()
        L = self.ctx.L
        if self.resident:
            self._resident_call(L, self._ht, self.hits.mass.data_ptr(), self.hits.locus_hit_off.ctypes.data, self.n_frags, h)
        else:
            _lib.check(L.sbgpu_quantify_device(self.ctx.h, C.byref(self._an), C.byref(self._ht), self.hits.mass.data_ptr(),
                                               self.hits.locus_hit_off.ctypes.data, C.byref(self._ins), self.read_len, 0,
                                               self.theta.ctypes.data, self.status.ctypes.data, self.iters.ctypes.data,
                                               C.byref(h)), "sbgpu_quantify_device")
        if self.info is None:
            info = (C.c_int64 * 8)()
            _lib.check(L.sbgpu_bins_info(h, info), "sbgpu_bins_info")
            self.info = {"n_bins": int(info[2]), "n_elem": int(info[3]), "n_pairs": int(info[4]), "hits_in_bins": int(info[6])}
        if keep:
            bins = eb.LocusBins.__new__(eb.LocusBins)
            bins._export(L, self.annot, h, self.n_hits, self.annot.compat_words, self.annot.key_words, with_hit_bin=False)   # destroys the handle
            return bins
        L.sbgpu_bins_destroy(h)

    def set_law(self, insert):
        """Quantify under a GIVEN insert-size law from now on (an InsertSize: -i mean/sd, or an empirical law made elsewhere)."""
        self.insert, self.empirical = insert, False
        self._ins = insert._struct(self.read_len)

    def _resident_call(self, L, hits_struct, d_mass, hit_off, mapped_reads, h):
        """sbgpu_quantify_resident on device hits; mapped_reads: this rank's part of Sample::total_mapped_reads()."""
        if self.mapped_override is not None:
            mapped_reads = self.mapped_override
        _lib.check(L.sbgpu_quantify_resident(self.ctx.h, C.byref(self._an), C.byref(hits_struct), d_mass, hit_off,
                                             None if self.empirical else C.byref(self._ins), self.read_len, 0, int(mapped_reads),
                                             C.byref(self._par), self.comm.h if self.comm is not None else None,
                                             C.byref(self._used), C.byref(self._out), C.byref(h)), "sbgpu_quantify_resident")
        u = self._used
        self.law = {"mean": u.mean, "sd": u.sd, "use_emp": int(u.use_emp), "start_offset": int(u.start_offset),
                    "end_offset": int(u.end_offset), "total_reads": int(u.total_reads)}
        if u.use_emp:   # (emp_hist points into the handle: copied while it lives)
            self.law["emp_hist"] = np.ctypeslib.as_array(u.emp_hist, shape=(u.end_offset - u.start_offset + 1,)).copy()
        self.total_fpkm, self.total_mapped_reads = float(self._out.total_fpkm), int(self._out.total_mapped_reads)

    def stage_ms(self):
        """HIP-event times of the kernel stages of one more (untimed) step -> {stage: ms}."""
        L = self.ctx.L
        _lib.check(L.sbgpu_set_timing(self.ctx.h, 1), "sbgpu_set_timing")
        try:
            self.step()
            ms, names = (C.c_float * 16)(), (C.c_char_p * 16)()
            n = L.sbgpu_last_stage_ms(self.ctx.h, 16, ms, names)
            if n < 0:
                _lib.check(n, "sbgpu_last_stage_ms")
            return {names[i].decode(): float(ms[i]) for i in range(n)}
        finally:
            L.sbgpu_set_timing(self.ctx.h, 0)

    def host_entry(self, reps=2):
        """The PCIe-inclusive form of the same call: the sample's hits brought to PAGEABLE host memory (where a C++ driver holds
        them) and handed to sbgpu_quantify_host, which uploads them, runs the same kernels and returns theta.  -> dict with
        the median wall time of `reps` calls (after one warm-up call), the bytes of hits uploaded per call, and whether theta /
        status / iterations equal the resident call's bit for bit."""
        import time
        L = self.ctx.L
        self.step()
        want = (self.theta[:self.n_iso].copy(), self.status[:self.n_loci].copy(), self.iters[:self.n_loci].copy())
        hits = self.hits.host_hits(self.n_loci)
        a, h = self.annot._struct(), hits._struct()
        nbytes = sum(getattr(hits, k).nbytes for k in ("hit_locus", "feat_off", "feat_code", "feat_left", "feat_right", "mass"))
        theta, status, iters = np.zeros(self.n_iso + 1), np.zeros(self.n_loci + 1, np.int32), np.zeros(self.n_loci + 1, np.int32)
        used = _lib.sbgpu_insert_t()
        times, on_dev = [], C.c_int32(0)
        for _ in range(reps + 1):
            handle = C.c_void_p()
            t = time.perf_counter()
            _lib.check(L.sbgpu_quantify_host(self.ctx.h, C.byref(a), C.byref(h), hits.mass.ctypes.data,
                                             None if self.empirical else C.byref(self._ins), self.read_len, 0,
                                             theta.ctypes.data, status.ctypes.data, iters.ctypes.data, None, C.byref(used),
                                             C.byref(handle)), "sbgpu_quantify_host")
            times.append(time.perf_counter() - t)
            _lib.check(L.sbgpu_bins_grouping(handle, C.byref(on_dev), None), "sbgpu_bins_grouping")
            L.sbgpu_bins_destroy(handle)
        ms = float(np.median(times[1:])) * 1e3
        same = bool((theta[:self.n_iso] == want[0]).all() and (status[:self.n_loci] == want[1]).all() and (iters[:self.n_loci] == want[2]).all())
        return {"ms_per_call": ms, "first_call_ms": times[0] * 1e3, "hit_bytes": int(nbytes), "GBps_end_to_end": nbytes / ms / 1e6,
                "grouped_on_device": bool(on_dev.value), "equals_resident_call_bitwise": same}

    def finish(self):
        """(bench.py's timed_steps calls this when the timed steps are issued: every step() has synchronised already)"""

    def close(self):
        """Release what this object keeps with the (shared) context: the pinned annotation.  The pin is keyed on the
        annotation arrays' addresses (+ a sampled fingerprint); it must not outlive the arrays.  Also on __exit__ / __del__."""
        self.unpin()

    def unpin(self):
        """Releases THIS object's pin only: another quantifier on the same context may have pinned its own annotation since
        (one pin per context, the later replaces the earlier) and must keep it."""
        if getattr(self, "pinned", False):
            self.pinned = False
            self.ctx.L.sbgpu_annotation_unpin_matching(self.ctx.h, C.byref(self._an), None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.unpin()
        except Exception:       # interpreter shutdown: the library may be gone already
            pass
