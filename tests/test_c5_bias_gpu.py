"""BASELINE config 5: the bias factors 2^(row_bias[i] * iso_bias[j]) applied by the tile kernels as they load their tiles
(sbgpu_em_run_device_bias) give the result of the unbiased entry on the pre-multiplied weights -- status and
iteration counts equal, theta to 1e-9 (exp2 on the device against numpy's) -- and that result is the oracle's on those
weights; the fp32 form stays close.  Not a parity path of the reference (it has no bias arithmetic)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_device_bias_equals_premultiplied_weights(oracle):
    import torch
    from strawberry_amd import em, synth
    ctx = em.default_context(0)
    b = synth.make_c3(n_loci=3000, total_frags=2e6, seed=0x5745)
    rng = np.random.default_rng(5)
    row_bias = rng.uniform(-1, 1, int(b.row_off[-1]))
    iso_bias = rng.uniform(-1, 1, int(b.iso_off[-1]))
    s = em.EmBatchSolver(b, ctx)
    s.set_bias(torch.from_numpy(row_bias).cuda(), torch.from_numpy(iso_bias).cuda())
    s.run_em(); s.synchronize()
    got = s.results()
    # the same problem with the factors multiplied in on the host
    Fb = b.F.copy()
    for l in range(b.n_loci):
        r0, r1, j0, j1 = b.row_off[l], b.row_off[l + 1], b.iso_off[l], b.iso_off[l + 1]
        Fb[b.f_off[l]:b.f_off[l + 1]] *= np.exp2(np.outer(row_bias[r0:r1], iso_bias[j0:j1])).reshape(-1)
    theta, status, iters = oracle.em_batch(b.row_off, b.iso_off, b.f_off, b.count, Fb)
    assert (got["status"] == status).mean() > 0.999 and (got["iters"] == iters).mean() > 0.995   # exp2's last bit may move a count
    same = (got["status"] == status) & (got["iters"] == iters)
    m = np.repeat(same, b.niso)
    err = np.abs(got["theta"][m] - theta[m]) / np.maximum(np.abs(theta[m]), 1e-9)
    assert err.max() < 1e-9
    # the bias changes the answer (it is not a no-op), and without it the plain entry is untouched
    s.set_bias(None, None)
    s.run_em(); s.synchronize()
    plain = s.results()
    t0, st0, it0 = oracle.em_batch(b.row_off, b.iso_off, b.f_off, b.count, b.F)
    assert (plain["status"] == st0).all() and (plain["iters"] == it0).all()
    assert np.abs(plain["theta"] - got["theta"]).max() > 1.0
    # fp32 with the same factors
    s.set_bias(torch.from_numpy(row_bias).cuda(), torch.from_numpy(iso_bias).cuda())
    s.run_em_f32(); s.synchronize()
    t32 = s.d_theta32.cpu().numpy().astype(np.float64)
    ok = np.repeat(np.isin(got["status"], (0, 3)), b.niso)
    rel = np.abs(t32[ok] - got["theta"][ok]) / np.maximum(got["theta"][ok], 1.0)
    assert np.percentile(rel, 95) < 1e-2


def test_device_bias_on_wide_loci_equals_premultiplied_weights(oracle):
    """Round 5: the multi-workgroup kernel of the loci beyond 64 isoforms applies the factors at ITS tile load too (register
    rows and LDS rows alike): a batch with wide loci -- one to several workgroups each, every column-lane width -- under
    sbgpu_em_run_device_bias against the oracle on the pre-multiplied weights."""
    import torch
    from strawberry_amd import em, synth
    from strawberry_amd.synth import _generate
    ctx = em.default_context(0)
    rng = np.random.Generator(np.random.PCG64(77))
    niso = np.array([70, 100, 150, 200, 330, 420, 65], np.int64)
    nrow = np.array([300, 900, 250, 1300, 500, 120, 40], np.int64)
    wide = _generate(rng, nrow, niso, (nrow * 40).astype(np.int64), name="wide")
    small = synth.make_random(n_loci=100, seed=8)
    b = synth.from_loci([wide.locus(l) for l in range(wide.n_loci)] + [small.locus(l) for l in range(small.n_loci)])
    row_bias = rng.uniform(-1, 1, int(b.row_off[-1]))
    iso_bias = rng.uniform(-1, 1, int(b.iso_off[-1]))
    s = em.EmBatchSolver(b, ctx)
    assert (s.plan.locus_kinds()[:7] == 5).all()
    s.set_bias(torch.from_numpy(row_bias).cuda(), torch.from_numpy(iso_bias).cuda())
    s.run_em(); s.synchronize()
    got = s.results()
    Fb = b.F.copy()
    for l in range(b.n_loci):
        r0, r1, j0, j1 = b.row_off[l], b.row_off[l + 1], b.iso_off[l], b.iso_off[l + 1]
        Fb[b.f_off[l]:b.f_off[l + 1]] *= np.exp2(np.outer(row_bias[r0:r1], iso_bias[j0:j1])).reshape(-1)
    theta, status, iters = oracle.em_batch(b.row_off, b.iso_off, b.f_off, b.count, Fb, threads=8)
    np.testing.assert_array_equal(got["status"][:7], status[:7])
    assert (np.abs(got["iters"][:7].astype(int) - iters[:7]) <= 1).all()     # (exp2's last bit may move a count)
    same = (got["status"] == status) & (got["iters"] == iters)
    assert same[:7].sum() >= 5 and same.mean() > 0.97
    m = np.repeat(same, b.niso)
    err = np.abs(got["theta"][m] - theta[m]) / np.maximum(np.abs(theta[m]), 1e-9)
    assert err.max() < 1e-9
    s.set_bias(None, None)
    s.run_em(); s.synchronize()
    plain = s.results()
    assert np.abs(plain["theta"][:int(b.iso_off[7])] - got["theta"][:int(b.iso_off[7])]).max() > 1.0     # not a no-op on the wide loci
