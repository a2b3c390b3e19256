"""Deterministic synthetic locus batches (C2, C2-U, C3).

A batch is the (n, F) input of the reference's ``EmSolver::init``
(/root/reference/src/estimate.cpp:366) for many loci at once, in the
CSR-of-loci layout of include/sbgpu.h.  The generative model is the reference's
own latent-class model with column-normalised F:

    mask_ij ~ Bernoulli(density), at least one 1 per row and per column
    F_ij    = mask_ij * U(1e-3, 0.3)
    pi      ~ Dirichlet(0.5 * 1)
    p_i     = sum_j pi_j F_ij / c_j,  c_j = sum_i F_ij
    n       ~ Multinomial(N_locus, p)

numpy only; no GPU, no reference code.  Everything is drawn from one
``np.random.Generator(PCG64(seed))`` in a fixed order, so a (workload, seed) pair
names one batch exactly.
"""
from dataclasses import dataclass

import numpy as np


@dataclass
class LocusBatch:
    """Host-side CSR-of-loci arrays (all C-contiguous)."""
    row_off: np.ndarray   # int64[n_loci+1]  first bin (row) of each locus
    iso_off: np.ndarray   # int64[n_loci+1]  first isoform of each locus
    f_off: np.ndarray     # int64[n_loci+1]  first element of each locus' row-major F block
    count: np.ndarray     # int32[total_rows]  per-bin fragment count n_i
    F: np.ndarray         # float64[total_elems]  bin weights
    length: np.ndarray    # int32[total_isoforms]  exonic length L_j (for FPKM)
    name: str = "custom"

    @property
    def n_loci(self):
        return len(self.row_off) - 1

    @property
    def nrow(self):
        return np.diff(self.row_off)

    @property
    def niso(self):
        return np.diff(self.iso_off)

    @property
    def n_frags(self):
        """Fragments represented by the batch (sum of all n_i)."""
        return int(self.count.astype(np.int64).sum())

    def algorithmic_bytes(self, elem_size=8):
        """B_locus = nrow*niso*s + nrow*4 + niso*s + 24, summed."""
        nrow, niso = self.nrow, self.niso
        return int((nrow * niso * elem_size + nrow * 4 + niso * elem_size + 24).sum())

    def algorithmic_flops(self, iters):
        """Fl_locus = iters*(5*nrow*niso + nrow + 3*niso), summed."""
        nrow, niso = self.nrow, self.niso
        return int((np.asarray(iters, np.int64) * (5 * nrow * niso + nrow + 3 * niso)).sum())

    def locus(self, l):
        """-> (count[nrow], F[nrow, niso]) views of locus l."""
        r0, r1 = self.row_off[l], self.row_off[l + 1]
        k = int(self.iso_off[l + 1] - self.iso_off[l])
        return self.count[r0:r1], self.F[self.f_off[l]:self.f_off[l + 1]].reshape(int(r1 - r0), k)

    def select(self, idx):
        """Sub-batch made of the loci in ``idx`` (in that order)."""
        idx = np.asarray(idx, np.int64)
        nrow, niso = self.nrow[idx], self.niso[idx]
        row_off = np.concatenate([[0], np.cumsum(nrow)]).astype(np.int64)
        iso_off = np.concatenate([[0], np.cumsum(niso)]).astype(np.int64)
        f_off = np.concatenate([[0], np.cumsum(nrow * niso)]).astype(np.int64)
        count = np.concatenate([self.count[self.row_off[l]:self.row_off[l + 1]] for l in idx]) \
            if len(idx) else np.zeros(0, np.int32)
        F = np.concatenate([self.F[self.f_off[l]:self.f_off[l + 1]] for l in idx]) if len(idx) else np.zeros(0)
        length = np.concatenate([self.length[self.iso_off[l]:self.iso_off[l + 1]] for l in idx]) \
            if len(idx) else np.zeros(0, np.int32)
        return LocusBatch(row_off, iso_off, f_off, np.ascontiguousarray(count, np.int32),
                          np.ascontiguousarray(F, np.float64), np.ascontiguousarray(length, np.int32),
                          self.name + "[sel]")


def from_loci(loci, lengths=None, name="custom"):
    """Build a batch from a list of (count[nrow], F[nrow, niso]) pairs."""
    nrow = np.array([np.asarray(F).reshape(len(c), -1).shape[0] for c, F in loci], np.int64)
    niso = np.array([np.asarray(F).reshape(len(c), -1).shape[1] if len(c) else np.asarray(F).shape[-1]
                     for c, F in loci], np.int64)
    row_off = np.concatenate([[0], np.cumsum(nrow)]).astype(np.int64)
    iso_off = np.concatenate([[0], np.cumsum(niso)]).astype(np.int64)
    f_off = np.concatenate([[0], np.cumsum(nrow * niso)]).astype(np.int64)
    count = np.concatenate([np.asarray(c, np.int32).reshape(-1) for c, _ in loci]) if loci else np.zeros(0, np.int32)
    F = np.concatenate([np.asarray(f, np.float64).reshape(-1) for _, f in loci]) if loci else np.zeros(0)
    if lengths is None:
        length = np.full(int(iso_off[-1]), 1000, np.int32)
    else:
        length = np.concatenate([np.asarray(x, np.int32).reshape(-1) for x in lengths])
    return LocusBatch(row_off, iso_off, f_off, np.ascontiguousarray(count, np.int32),
                      np.ascontiguousarray(F, np.float64), np.ascontiguousarray(length, np.int32), name)


def _segment_sum(values, seg_id, n_seg):
    return np.bincount(seg_id, weights=values, minlength=n_seg)


def _generate(rng, nrow, niso, n_frags, density=0.4, name="synth"):
    """Vectorised draw of a ragged batch. nrow, niso, n_frags: int64[n_loci]."""
    n_loci = len(nrow)
    row_off = np.concatenate([[0], np.cumsum(nrow)]).astype(np.int64)
    iso_off = np.concatenate([[0], np.cumsum(niso)]).astype(np.int64)
    f_off = np.concatenate([[0], np.cumsum(nrow * niso)]).astype(np.int64)
    n_rows, n_isos, n_el = int(row_off[-1]), int(iso_off[-1]), int(f_off[-1])

    # per-row / per-element index helpers
    row_locus = np.repeat(np.arange(n_loci), nrow)                 # locus of each row
    row_niso = niso[row_locus]
    el_row = np.repeat(np.arange(n_rows), row_niso)                # global row of each element
    el_start = np.concatenate([[0], np.cumsum(row_niso)])[:-1]
    el_col = np.arange(n_el) - np.repeat(el_start, row_niso)       # column j of each element
    el_locus = row_locus[el_row]
    el_iso = iso_off[el_locus] + el_col                            # global isoform of each element

    mask = rng.random(n_el) < density
    # >= 1 per row: rows with none get one random column
    row_has = np.bincount(el_row, weights=mask, minlength=n_rows) > 0
    fix_rows = np.nonzero(~row_has)[0]
    if len(fix_rows):
        j = (rng.random(len(fix_rows)) * row_niso[fix_rows]).astype(np.int64)
        mask[el_start[fix_rows] + j] = True
    # >= 1 per column: columns with none get one random row of their locus
    col_has = np.bincount(el_iso, weights=mask, minlength=n_isos) > 0
    fix_cols = np.nonzero(~col_has)[0]
    if len(fix_cols):
        iso_locus = np.repeat(np.arange(n_loci), niso)
        lc = iso_locus[fix_cols]
        i = (rng.random(len(fix_cols)) * nrow[lc]).astype(np.int64)
        jj = fix_cols - iso_off[lc]
        mask[f_off[lc] + i * niso[lc] + jj] = True

    F = np.where(mask, rng.uniform(1e-3, 0.3, n_el), 0.0)
    # pi ~ Dirichlet(0.5): normalised Gamma(0.5)
    g = rng.gamma(0.5, 1.0, n_isos) + 1e-300
    iso_locus = np.repeat(np.arange(n_loci), niso)
    pi = g / _segment_sum(g, iso_locus, n_loci)[iso_locus]
    c = _segment_sum(F, el_iso, n_isos)
    p = _segment_sum(pi[el_iso] * F / c[el_iso], el_row, n_rows)
    # n ~ Multinomial(N_locus, p) by sequential conditional binomials, vectorised over loci
    count = np.zeros(n_rows, np.int64)
    remaining = n_frags.astype(np.int64).copy()
    rem_p = np.ones(n_loci)
    order = np.argsort(-nrow, kind="stable")
    nrow_sorted = nrow[order]
    for r in range(int(nrow.max()) if n_loci else 0):
        k = int(np.searchsorted(-nrow_sorted, -(r + 1), side="right"))  # loci with nrow > r
        if k == 0:
            break
        ls = order[:k]
        pr = p[row_off[ls] + r]
        q = np.clip(pr / np.maximum(rem_p[ls], 1e-300), 0.0, 1.0)
        last = nrow[ls] == r + 1
        q[last] = 1.0
        x = rng.binomial(remaining[ls], q)
        count[row_off[ls] + r] = x
        remaining[ls] -= x
        rem_p[ls] -= pr
    length = rng.integers(500, 5001, n_isos).astype(np.int32)
    return LocusBatch(row_off, iso_off, f_off, count.astype(np.int32), F, length, name)


def make_c2(n_loci=10000, niso=8, nrow=32, n_frags=1000, seed=0x5742, unbinned=False):
    """Config C2: ``n_loci`` loci x ``niso`` isoforms x ``n_frags`` fragments each, collapsed
    into ``nrow`` exon bins.  ``unbinned=True`` gives C2-U: one row per fragment (n_i = 1)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    b = _generate(rng, np.full(n_loci, nrow, np.int64), np.full(n_loci, niso, np.int64),
                  np.full(n_loci, n_frags, np.int64), name="C2")
    if not unbinned:
        return b
    # C2-U: expand every bin row into n_i identical rows with count 1
    reps = b.count.astype(np.int64)
    Fm = b.F.reshape(-1, niso)
    F = np.repeat(Fm, reps, axis=0).reshape(-1)
    nrow_u = np.add.reduceat(reps, b.row_off[:-1])
    row_off = np.concatenate([[0], np.cumsum(nrow_u)]).astype(np.int64)
    f_off = row_off * niso
    return LocusBatch(row_off, b.iso_off, f_off.astype(np.int64), np.ones(int(row_off[-1]), np.int32),
                      np.ascontiguousarray(F), b.length, "C2-U")


def make_c3(n_loci=60000, total_frags=2e8, seed=0x5743, max_niso=200, max_nrow=2000):
    """Config C3, human-scale: niso ~ 1+Geom(0.25) clipped to 200 (mean ~4),
    nrow ~ round(LogNormal(ln 30, 1)) clipped to [1, 2000], fragments per locus
    LogNormal scaled so that the batch holds ``total_frags``."""
    rng = np.random.Generator(np.random.PCG64(seed))
    niso = np.minimum(rng.geometric(0.25, n_loci), max_niso).astype(np.int64)
    nrow = np.clip(np.rint(rng.lognormal(np.log(30.0), 1.0, n_loci)), 1, max_nrow).astype(np.int64)
    w = rng.lognormal(0.0, 1.0, n_loci)
    n_frags = np.maximum(1, np.rint(w / w.sum() * total_frags)).astype(np.int64)
    return _generate(rng, nrow, niso, n_frags, name="C3")


def make_c3t(n_loci=60000, total_frags=2e8, seed=0x5743, n_tail=300):
    """C3-T: C3 plus a human-annotation-shaped TAIL the C3 law (niso ~ 1 + Geom(0.25), at most ~36) never draws --
    `n_tail` loci of 65..400 isoforms and 200..3000 bins (both log-uniform), 50 fragments per bin.  These are the
    loci of a GENCODE-size annotation that do not fit a workgroup's registers (em_wide_kernel / em_stream_kernel).
    The tail loci are spread evenly through the batch."""
    base = make_c3(n_loci=n_loci, total_frags=total_frags, seed=seed)
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x7A11))
    niso = np.exp(rng.uniform(np.log(65), np.log(400), n_tail)).astype(np.int64)
    nrow = np.exp(rng.uniform(np.log(200), np.log(3000), n_tail)).astype(np.int64)
    tail = _generate(rng, nrow, niso, nrow * 50, name="tail")
    # interleave: tail locus k goes in front of base locus pos[k]
    pos = np.linspace(0, n_loci, n_tail, endpoint=False).astype(np.int64)
    parts, prev = [], 0
    for k in range(n_tail):
        if pos[k] > prev:
            parts.append(base.select(np.arange(prev, pos[k])))
        parts.append(tail.select(np.array([k])))
        prev = pos[k]
    if prev < n_loci:
        parts.append(base.select(np.arange(prev, n_loci)))
    return concat_batches(parts, "C3-T")


def concat_batches(parts, name="batch"):
    """Concatenate LocusBatches (loci keep their order)."""
    row_off = np.concatenate([[0]] + [p.row_off[1:] + o for p, o in zip(parts, np.cumsum([0] + [int(p.row_off[-1]) for p in parts[:-1]]))])
    iso_off = np.concatenate([[0]] + [p.iso_off[1:] + o for p, o in zip(parts, np.cumsum([0] + [int(p.iso_off[-1]) for p in parts[:-1]]))])
    f_off = np.concatenate([[0]] + [p.f_off[1:] + o for p, o in zip(parts, np.cumsum([0] + [int(p.f_off[-1]) for p in parts[:-1]]))])
    return LocusBatch(row_off.astype(np.int64), iso_off.astype(np.int64), f_off.astype(np.int64),
                      np.concatenate([p.count for p in parts]), np.concatenate([p.F for p in parts]),
                      np.concatenate([p.length for p in parts]), name)


def make_c5(n_loci=60000, total_frags=4e8, seed=0x5745):
    """Config C5: the C3 law at 4e8 fragments.  The bias factors b_ij in [0.5, 2] are NOT multiplied in
    here: they are applied on the device, inside the EM kernels, as the tiles are loaded (strawberry_amd/bias.py,
    sbgpu_em_run_device_bias)."""
    b = make_c3(n_loci=n_loci, total_frags=total_frags, seed=seed)
    b.name = "C5"
    return b


def make_random(n_loci=256, max_nrow=64, max_niso=12, density=0.4, max_count=60, seed=12345):
    """The survey's ``em_random`` law: nrow in [1,max_nrow], niso in [1,max_niso],
    weights mask*U(1e-3,0.3) (no >=1 guarantees, so dropped rows / zero columns occur),
    counts U{0..max_count-1}."""
    rng = np.random.Generator(np.random.PCG64(seed))
    loci = []
    for _ in range(n_loci):
        nrow = int(rng.integers(1, max_nrow + 1))
        niso = int(rng.integers(1, max_niso + 1))
        F = np.where(rng.random((nrow, niso)) < density, rng.uniform(1e-3, 0.3, (nrow, niso)), 0.0)
        n = rng.integers(0, max_count, nrow).astype(np.int32)
        loci.append((n, F))
    return from_loci(loci, name="random")


# ---------------------------------------------------------------------------------------------
# Synthetic annotation + fragments for the exon-bin path (A5).  Gene models and read pairs of our
# own making, shaped like the reference's inputs: loci of alternatively spliced isoforms, paired
# reads sampled from them, plus fragments that fit no isoform (unspliced, shifted, novel junctions).
def make_gene_models(n_loci, seed=7, max_exons=12, max_isoforms=6, ex_lo=60, ex_hi=400, in_lo=80, in_hi=600):
    """-> list (per locus) of lists (per isoform) of sorted closed exons [(l, r), ...]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    loci = []
    pos = 1000
    for _ in range(n_loci):
        n_ex = int(rng.integers(1, max_exons + 1))
        exons = []
        for _ in range(n_ex):
            ln = int(rng.integers(ex_lo, ex_hi + 1))
            exons.append((pos, pos + ln - 1))
            pos += ln + int(rng.integers(in_lo, in_hi + 1))
        n_iso = int(rng.integers(1, max_isoforms + 1))
        isos = []
        seen = set()
        for k in range(n_iso):
            if k == 0 or n_ex < 3:
                use = list(range(n_ex))
            else:
                use = [e for e in range(n_ex) if e in (0, n_ex - 1) or rng.random() < 0.7]
            ex = [exons[e] for e in use]
            # alternative 5'/3' ends: shorten an exon now and then (creates sub-exon segments)
            if k and rng.random() < 0.5:
                e = int(rng.integers(0, len(ex)))
                a, b = ex[e]
                cut = int(rng.integers(5, max(6, (b - a) // 2)))
                ex[e] = (a + cut, b) if rng.random() < 0.5 else (a, b - cut)
            if tuple(ex) in seen:
                continue
            seen.add(tuple(ex))
            isos.append(ex)
        loci.append(isos)
        pos += 5000
    return loci


def _tx_blocks(ex, t0, t1):
    """transcript interval [t0, t1) -> genomic blocks."""
    out, off = [], 0
    for (a, b) in ex:
        ln = b - a + 1
        lo, hi = max(t0, off), min(t1, off + ln)
        if lo < hi:
            out.append((a + lo - off, a + hi - off - 1))
        off += ln
    return out


def make_fragments(loci, hits_per_locus, seed=11, read_len=75, mean=250.0, sd=30.0, noise=0.15, single=0.05):
    """-> (hit_locus list, [(left_blocks, right_blocks), ...]) in (locus, left, right) order.
    `noise`: share of pairs that are shifted / unspliced so that they fit fewer (or no) isoforms;
    `single`: share of single-end fragments (right mate missing)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    for l, isos in enumerate(loci):
        for _ in range(hits_per_locus):
            ex = isos[int(rng.integers(0, len(isos)))]
            L = sum(b - a + 1 for a, b in ex)
            rl = min(read_len, L)
            fl = int(np.clip(np.rint(rng.normal(mean, sd)), rl, L))
            st = int(rng.integers(0, L - fl + 1))
            left = _tx_blocks(ex, st, st + rl)
            right = _tx_blocks(ex, st + fl - rl, st + fl)
            u = rng.random()
            if u < noise / 3:            # genomic (unspliced) read: runs into the intron
                left = [(left[0][0], left[0][0] + rl - 1)]
            elif u < 2 * noise / 3:      # shifted by a few bases: junctions no longer match
                d = int(rng.integers(1, 9))
                right = [(a + d, b + d) for a, b in right]
            elif u < noise:              # novel junction: skip from the first block to the last
                if len(right) > 1:
                    right = [right[0], right[-1]]
            if rng.random() < single:
                right = []
            out.append((l, left[0][0], (right or left)[-1][1], left, right))
    out.sort(key=lambda r: (r[0], r[1], r[2]))
    return [r[0] for r in out], [(r[3], r[4]) for r in out]
