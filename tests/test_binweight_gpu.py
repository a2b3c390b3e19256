"""GPU parity tests of the bin-weight kernel through the C ABI:
integer effective lengths are exact by construction of the sum test, weights agree
with the reference's goldens and with the oracle to 1e-12 relative."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "binweight_pairs.npz")
RTOL = 1e-12


@pytest.fixture(scope="module")
def ctx():
    from strawberry_amd import em
    return em.default_context(0)


def check(w, ref, rtol=RTOL):
    """Every weight, the far tail included: the reference is built -Ofast (FTZ / DAZ), so a term whose product or
    quotient comes out subnormal is 0 there; the kernel flushes the same terms (bw_flush).  A weight that is exactly 0 in
    the reference is exactly 0 here; the others agree to `rtol`."""
    nz = ref != 0
    assert ((w != 0) == nz).all(), int(np.nonzero((w != 0) != nz)[0][0])
    err = np.abs(w[nz] - ref[nz]) / np.abs(ref[nz])
    assert err.max() < rtol, (err.max(), int(np.nonzero(nz)[0][err.argmax()]))


def test_binweight_far_tail_is_flushed_like_the_reference(ctx, oracle):
    """Bins over long segments: fragment lengths thirty and more standard deviations out, densities from 1e-280 down to the
    smallest normal and below.  The oracle runs with FTZ (liboracle.so is built -Ofast like the reference); the kernel
    agrees on every weight to 1e-12 and on which weights are exact zeros."""
    from strawberry_amd.binweight import InsertSize, bin_weights, pack_pairs
    rng = np.random.Generator(np.random.PCG64(77))
    segs, imps, lens = [], [], []
    for _ in range(3000):
        nseg = int(rng.integers(2, 9))
        s = rng.integers(50, 420, nseg)
        # the inner segments are implicit (the mates' gap swallows them): lmin = their sum, hundreds to thousands of bases
        imp = list(range(1, nseg - 1)) if rng.random() < 0.8 else sorted(rng.choice(np.arange(1, nseg - 1), max(0, nseg - 3), replace=False).tolist()) if nseg > 3 else []
        segs.append(s)
        imps.append(imp)
        lens.append(int(s.sum() + rng.integers(0, 1500)))
    seg_off, seg_lens, mask = pack_pairs(segs, imps)
    for rl, mean, sd in ((75, 250.0, 30.0), (50, 200.0, 20.0)):
        w = bin_weights(seg_off, seg_lens, mask, lens, InsertSize(mean, sd), rl, ctx=ctx)
        ins = oracle.make_insert(mean, sd)
        ref = np.array([oracle.bin_weight(s, i, L, rl, ins) for s, i, L in zip(segs, imps, lens)])
        tiny = (ref != 0) & (ref < 1e-280)
        assert tiny.sum() >= 5 and (ref == 0).sum() >= 50 and (ref > 1e-200).sum() >= 50, (int(tiny.sum()), int((ref == 0).sum()))
        check(w, ref, 1e-12)


def test_binweight_matches_reference_goldens(ctx):
    from strawberry_amd.binweight import InsertSize, bin_weights
    z = np.load(GOLD)
    w = bin_weights(z["seg_off"], z["seg_lens"], z["implicit_mask"], z["iso_len"], InsertSize(230.0, 35.0), 75, ctx=ctx)
    check(w, z["w_gauss"])
    ins = InsertSize.from_frag_lens(z["frag_lens"])
    w = bin_weights(z["seg_off"], z["seg_lens"], z["implicit_mask"], z["iso_len"], ins, 50, ctx=ctx)
    check(w, z["w_emp"])


def test_binweight_known_answers(ctx):
    """SURVEY.md bin-weight KAT (reference ctx.tsv, -i 200/20, read length 50)."""
    from strawberry_amd.binweight import InsertSize, bin_weights, pack_pairs
    from test_binweight_oracle import KAT
    segs = [k[0] for k in KAT]
    imps = [k[1] for k in KAT]
    seg_off, seg_lens, mask = pack_pairs(segs, imps)
    w = bin_weights(seg_off, seg_lens, mask, [k[2] for k in KAT], InsertSize(200.0, 20.0), 50, ctx=ctx)
    for got, k in zip(w, KAT):
        assert abs(got - k[3]) / k[3] < 5e-11  # the TSV prints 12 significant digits


def test_binweight_vs_oracle_wide_sweep(ctx, oracle):
    """Random pairs incl. many-segment bins (the brute-force >= 5-segment scan), short and long
    read lengths, degenerate ranges (lmin > lmax -> weight 0)."""
    from strawberry_amd.binweight import InsertSize, bin_weights, pack_pairs
    rng = np.random.Generator(np.random.PCG64(31))
    segs, imps, lens = [], [], []
    for _ in range(4000):
        nseg = int(rng.integers(1, 17))
        s = rng.integers(1, 350, nseg)
        if nseg <= 2:
            imp = []
        elif nseg == 3:
            imp = [1] if rng.random() < .5 else []
        elif nseg == 4:
            imp = [[], [1], [2], [1, 2]][int(rng.integers(0, 4))]
        else:
            imp = sorted(rng.choice(np.arange(1, nseg - 1), int(rng.integers(0, nseg - 1)), replace=False).tolist())
        segs.append(s)
        imps.append(imp)
        lens.append(int(s.sum() + rng.integers(0, 2000)))
    seg_off, seg_lens, mask = pack_pairs(segs, imps)
    for rl, mean, sd in ((36, 180.0, 25.0), (100, 320.0, 60.0)):
        w = bin_weights(seg_off, seg_lens, mask, lens, InsertSize(mean, sd), rl, ctx=ctx)
        ins = oracle.make_insert(mean, sd)
        ref = np.array([oracle.bin_weight(s, i, L, rl, ins) for s, i, L in zip(segs, imps, lens)])
        check(w, ref, 1e-11)


def test_binweight_long_read_and_pdf_table(ctx, oracle):
    from strawberry_amd.binweight import InsertSize, bin_weights, pack_pairs
    seg_off, seg_lens, mask = pack_pairs([[100, 200], [50]], [[], []])
    w = bin_weights(seg_off, seg_lens, mask, [1234, 777], InsertSize(), 50, long_read=True, ctx=ctx)
    np.testing.assert_array_equal(w, [1.0 / 1234, 1.0 / 777])  # estimate.cpp:236-247
    # the host-built pdf table == InsertSize::emp_dist_pdf (oracle)
    rng = np.random.Generator(np.random.PCG64(3))
    fl = np.rint(rng.normal(220, 25, 500)).astype(np.int32)
    ins = InsertSize.from_frag_lens(fl)
    tab = ins.pdf_table(500)
    o = oracle.make_insert(ins.mean, ins.sd, fl)
    ref = np.array([oracle.insert_pdf(o, x) for x in range(500)])
    np.testing.assert_allclose(tab, ref, rtol=1e-14, atol=0)


def test_binweight_writes_into_em_batch_and_feeds_em(ctx, oracle):
    """Device-resident pipeline: the kernel scatters weights straight into an EM batch's F
    (out_index), then the EM runs on it -- same theta as oracle weights + oracle EM."""
    import ctypes as C
    import torch
    from strawberry_amd import em, synth
    from strawberry_amd.binweight import InsertSize, pack_pairs
    rng = np.random.Generator(np.random.PCG64(8))
    n_loci, niso, nrow = 40, 3, 6
    segs, imps, lens, out_index = [], [], [], []
    F = np.zeros(n_loci * nrow * niso)
    iso_len = rng.integers(800, 3000, (n_loci, niso))
    for l in range(n_loci):
        for i in range(nrow):
            for j in range(niso):
                if rng.random() < 0.6 or j == i % niso:
                    nseg = int(rng.integers(1, 5))
                    s = rng.integers(60, 300, nseg)
                    imp = [] if nseg < 3 else ([1] if (nseg == 3 and rng.random() < .5) else [])
                    segs.append(s)
                    imps.append(imp)
                    lens.append(int(max(iso_len[l, j], s.sum())))
                    out_index.append((l * nrow + i) * niso + j)
    seg_off, seg_lens, mask = pack_pairs(segs, imps)
    ins = InsertSize(200.0, 30.0)
    dev = torch.device("cuda", 0)
    d = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt)).to(dev)
    d_off, d_seg, d_mask = d(seg_off, np.int64), d(seg_lens, np.uint32).view(torch.int32) if False else d(seg_lens.view(np.int32), np.int32), d(mask.view(np.int32), np.int32)
    d_len, d_idx = d(lens, np.int32), d(out_index, np.int64)
    pdf = ins.pdf_table(int(max(s.sum() for s in segs)) + 1)
    d_pdf = d(pdf, np.float64)
    count = rng.integers(0, 80, n_loci * nrow).astype(np.int32)
    b = synth.from_loci([(count[l * nrow:(l + 1) * nrow], np.zeros((nrow, niso))) for l in range(n_loci)])
    s = em.EmBatchSolver(b, ctx)
    s.d_F.zero_()
    from strawberry_amd import _lib
    _lib.check(ctx.L.sbgpu_binweight_device(ctx.h, len(lens), d_off.data_ptr(), d_seg.data_ptr(), d_mask.data_ptr(),
                                            d_len.data_ptr(), d_idx.data_ptr(), d_pdf.data_ptr(), len(pdf), 50, 50, 0,
                                            s.d_F.data_ptr(), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
               "sbgpu_binweight_device")
    s.run_em()
    r = s.results()
    o_ins = oracle.make_insert(200.0, 30.0)
    for p, idx in enumerate(out_index):
        F[idx] = oracle.bin_weight(segs[p], imps[p], lens[p], 50, o_ins)
    np.testing.assert_allclose(s.d_F.cpu().numpy(), F, rtol=1e-12, atol=0)
    theta, status, iters = oracle.em_batch(b.row_off, b.iso_off, b.f_off, b.count, F)
    np.testing.assert_array_equal(r["status"], status)
    np.testing.assert_array_equal(r["iters"], iters)
    assert (np.abs(r["theta"] - theta) / np.maximum(np.abs(theta), 1e-9)).max() < 1e-9
