"""The fragments -> abundances chain on hits that are resident in HBM (sbgpu_quantify_device), and the synthetic
human-scale input of bench.py's `c3-chain` workload.

A small sample (a few hundred gene models with their read pairs, made on the host like the tests' inputs) is
laid along the genome `copies` times: the annotation on the host (it is small), the hits ON THE DEVICE with
torch index arithmetic -- 2e8 fragments are ~10 GB of features that never exist in host memory.
"""
import ctypes as C

import numpy as np

from . import _lib
from . import exonbin as eb
from . import synth
from .binweight import InsertSize


class DeviceHits:
    """Hits of a tiled sample as torch tensors on the device (layout of sbgpu_hits_t)."""

    def __init__(self, torch, dev, base_hits, n_base_loci, copies, stride):
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        n, nf = base_hits.n_hits, int(base_hits.feat_off[-1])
        k_hit = torch.arange(copies, device=dev, dtype=torch.int64).repeat_interleave(n)
        k_feat = torch.arange(copies, device=dev, dtype=torch.int64).repeat_interleave(nf)
        self.n_hits = n * copies
        self.hit_locus = (up(base_hits.hit_locus).to(torch.int64).repeat(copies) + k_hit * n_base_loci).to(torch.int32)
        self.feat_off = torch.cat([up(base_hits.feat_off[:-1]).repeat(copies) + k_hit * nf,
                                   torch.tensor([nf * copies], device=dev, dtype=torch.int64)])
        self.feat_code = up(base_hits.feat_code).repeat(copies)
        # uint32 coordinates travel as int32 bit patterns
        self.feat_left = (up(base_hits.feat_left.astype(np.int64)).repeat(copies) + k_feat * stride).to(torch.int32)
        self.feat_right = (up(base_hits.feat_right.astype(np.int64)).repeat(copies) + k_feat * stride).to(torch.int32)
        self.mass = up(base_hits.mass).repeat(copies)
        del k_hit, k_feat
        self.n_features = nf * copies
        base_off = np.searchsorted(base_hits.hit_locus, np.arange(n_base_loci + 1), side="left").astype(np.int64)
        self.locus_hit_off = np.concatenate([base_off[:-1] + k * n for k in range(copies)] + [[n * copies]]).astype(np.int64)

    def struct(self):
        s = _lib.sbgpu_hits_t()
        s.n_hits = self.n_hits
        for k in ("hit_locus", "feat_off", "feat_code", "feat_left", "feat_right"):
            setattr(s, k, getattr(self, k).data_ptr())
        return s


def tile_annotation(a, copies, stride):
    """`copies` copies of the annotation `stride` bases apart (host arrays)."""
    big = eb.Annotation.__new__(eb.Annotation)
    n_iso, n_exon, n_seg = int(a.iso_off[-1]), int(a.exon_off[-1]), int(a.seg_off[-1])
    rep = lambda off, total: np.concatenate([[0]] + [off[1:] + k * total for k in range(copies)]).astype(np.int64)  # noqa: E731
    shift = lambda x: np.concatenate([x.astype(np.int64) + k * stride for k in range(copies)]).astype(np.uint32)  # noqa: E731
    big.n_loci = a.n_loci * copies
    big.iso_off, big.exon_off, big.seg_off = rep(a.iso_off, n_iso), rep(a.exon_off, n_exon), rep(a.seg_off, n_seg)
    big.exon_left, big.exon_right = shift(a.exon_left), shift(a.exon_right)
    big.seg_left, big.seg_right = shift(a.seg_left), shift(a.seg_right)
    big.compat_words, big.key_words = a.compat_words, a.key_words
    return big


class ChainQuantifier:
    """step(): fragments (in HBM) -> compat / key words -> bins -> weights -> EM -> theta, one C-ABI call;
    then FPKM / TPM on the host arrays the call returns (the caller's own epilogue, as in the reference)."""

    def __init__(self, ctx, n_loci=60000, n_frags=2e8, base_loci=100, seed=31, read_len=75):
        import torch
        self.torch, self.ctx = torch, ctx
        self.dev = torch.device("cuda", ctx.device)
        copies = max(1, int(round(n_loci / base_loci)))
        per_locus = max(1, int(round(n_frags / (copies * base_loci))))
        loci = synth.make_gene_models(base_loci, seed=seed)
        hl, pairs = synth.make_fragments(loci, per_locus, seed=seed + 1, single=0.0)
        feats, loc = [], []
        for l, (lb, rb) in zip(hl, pairs):
            f = eb.hit_features(lb, rb)
            if f is not None:
                feats.append(f)
                loc.append(l)
        base_annot, base_hits = eb.Annotation(loci), eb.Hits(loc, feats)
        stride = int(max(base_annot.exon_right.max(), base_hits.feat_right.max()) + 100000)
        if stride * copies >= 2 ** 32:
            raise ValueError("the tiled sample does not fit 32-bit coordinates")
        self.annot = tile_annotation(base_annot, copies, stride)
        self.hits = DeviceHits(torch, self.dev, base_hits, base_annot.n_loci, copies, stride)
        self.insert = InsertSize(250.0, 30.0)
        self.read_len = read_len
        self.n_loci, self.n_frags = self.annot.n_loci, self.hits.n_hits
        self.n_iso = int(self.annot.iso_off[-1])
        self.theta = np.zeros(self.n_iso + 1)
        self.status = np.zeros(self.n_loci + 1, np.int32)
        self.iters = np.zeros(self.n_loci + 1, np.int32)
        self._an = self.annot._struct()
        self._ht = self.hits.struct()
        self._ins = self.insert._struct(read_len)
        self.info = None

    def step(self):
        h = C.c_void_p()
        L = self.ctx.L
        _lib.check(L.sbgpu_quantify_device(self.ctx.h, C.byref(self._an), C.byref(self._ht), self.hits.mass.data_ptr(),
                                           self.hits.locus_hit_off.ctypes.data, C.byref(self._ins), self.read_len, 0,
                                           self.theta.ctypes.data, self.status.ctypes.data, self.iters.ctypes.data,
                                           C.byref(h)), "sbgpu_quantify_device")
        if self.info is None:
            info = (C.c_int64 * 8)()
            _lib.check(L.sbgpu_bins_info(h, info), "sbgpu_bins_info")
            self.info = {"n_bins": int(info[2]), "n_elem": int(info[3]), "n_pairs": int(info[4]), "hits_in_bins": int(info[6])}
        L.sbgpu_bins_destroy(h)

    def finish(self):
        pass
