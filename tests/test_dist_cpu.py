"""CPU tests of the N > 1 path: locus sharding and the single all-reduce, with two
gloo processes.  The per-locus arithmetic in these tests comes from the oracle
(tests may use it); what is under test is the host logic of strawberry_amd.dist."""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_loci_partition_and_balance():
    from strawberry_amd import dist, synth
    b = synth.make_c3(n_loci=3000, total_frags=1e6)
    for world in (1, 2, 4, 8):
        parts = dist.shard_loci(b.nrow, b.niso, world)
        allidx = np.sort(np.concatenate(parts))
        np.testing.assert_array_equal(allidx, np.arange(b.n_loci))       # every locus exactly once
        # the cost shard_loci balances: elements x the iteration count predicted from the shape (x 50 for wide loci)
        cost = b.nrow * b.niso * np.where(b.niso > 64, 50, 1) * np.maximum(dist.predicted_iterations(b.nrow, b.niso), 1) + 1
        loads = np.array([cost[p].sum() for p in parts])
        assert loads.max() <= loads.mean() + cost.max()                 # LPT bound
        for p in parts:
            assert (np.diff(p) > 0).all()                                # locus order kept inside a rank
    # deterministic
    a = dist.shard_loci(b.nrow, b.niso, 4)
    c = dist.shard_loci(b.nrow, b.niso, 4)
    for x, y in zip(a, c):
        np.testing.assert_array_equal(x, y)


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch
    sys.path.insert(0, %(root)r)
    from strawberry_amd import dist, synth
    from oracle import OracleLib
    rank, world, _ = dist.init_process_group("gloo")
    o = OracleLib()
    b = synth.make_c3(n_loci=600, total_frags=3e5, seed=77)
    parts = dist.shard_loci(b.nrow, b.niso, world)
    mine = b.select(parts[rank])
    # pass-1 normaliser: global mapped fragments (alignments.cpp:1372)
    tot = torch.tensor([mine.n_frags], dtype=torch.int64)
    dist.allreduce_sum_(tot)
    total_mapped = int(tot.item())
    theta, status, iters = o.em_batch(mine.row_off, mine.iso_off, mine.f_off, mine.count, mine.F)
    fpkm = np.zeros_like(theta); keep = np.zeros(len(theta), np.int32)
    for l in range(mine.n_loci):
        j0, j1 = mine.iso_off[l], mine.iso_off[l + 1]
        if status[l] == 1:
            continue
        f, fr, k, _ = o.abundance_locus(theta[j0:j1], mine.length[j0:j1], total_mapped, min_isoform_frac=0.0)
        fpkm[j0:j1], keep[j0:j1] = f, k
    s = torch.tensor([fpkm[keep != 0].sum()], dtype=torch.float64)
    dist.allreduce_sum_(s)                       # the one collective of the path
    tpm = np.where(keep != 0, 1e6 * fpkm / s.item(), 0.0)
    np.savez(os.path.join(%(out)r, "rank%%d.npz" %% rank), idx=parts[rank], iso_off=mine.iso_off, tpm=tpm,
             total_mapped=total_mapped, sum_fpkm=s.item())
    # the per-rank table of bench.py's N > 1 line: every rank hands in its numbers, every rank gets all of them
    tab = dist.gather_values([rank + 0.5, mine.n_loci, int((status == 3).sum())], rank, world)
    np.save(os.path.join(%(out)r, "table%%d.npy" %% rank), tab)
    dist.barrier()
""")


def test_two_rank_gloo_tpm_equals_single_process(tmp_path, oracle):
    from strawberry_amd import synth
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "out": str(tmp_path)})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29631", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    # single-process answer
    b = synth.make_c3(n_loci=600, total_frags=3e5, seed=77)
    theta, status, _ = oracle.em_batch(b.row_off, b.iso_off, b.f_off, b.count, b.F)
    fpkm = np.zeros_like(theta)
    keep = np.zeros(len(theta), np.int32)
    for l in range(b.n_loci):
        j0, j1 = b.iso_off[l], b.iso_off[l + 1]
        if status[l] == 1:
            continue
        f, fr, k, _ = oracle.abundance_locus(theta[j0:j1], b.length[j0:j1], b.n_frags, min_isoform_frac=0.0)
        fpkm[j0:j1], keep[j0:j1] = f, k
    tpm, total = oracle.tpm(fpkm, keep)
    got = np.full(len(tpm), np.nan)
    for rank in range(2):
        z = np.load(tmp_path / ("rank%d.npz" % rank))
        assert int(z["total_mapped"]) == b.n_frags
        assert abs(float(z["sum_fpkm"]) - total) / total < 1e-12     # up to summation order
        for pos, l in enumerate(z["idx"]):
            n = b.iso_off[l + 1] - b.iso_off[l]
            got[b.iso_off[l]:b.iso_off[l + 1]] = z["tpm"][z["iso_off"][pos]:z["iso_off"][pos] + n]
    assert not np.isnan(got).any()
    np.testing.assert_allclose(got, tpm, rtol=1e-12, atol=0)
    # dist.gather_values: both ranks hold the same [world][values] table, row r is rank r's
    t0, t1 = np.load(tmp_path / "table0.npy"), np.load(tmp_path / "table1.npy")
    np.testing.assert_array_equal(t0, t1)
    assert t0.shape == (2, 3) and t0[0, 0] == 0.5 and t0[1, 0] == 1.5
    assert t0[:, 1].sum() == b.n_loci and t0[:, 2].sum() == (status == 3).sum()


def test_abi_comm_id_broadcast_carries_an_explicit_status():
    """Rank 0's id (or its failure) travels as status byte + payload; a receiver never infers failure from the length:
    an error message of exactly 127 bytes used to pass as an id and went into sbgpu_comm_init."""
    import pytest
    from strawberry_amd import _lib, dist

    class NoCtx:        # never reached: the payload is refused before the communicator is made
        h = None

    with pytest.raises(_lib.SbgpuError, match="could not make an RCCL id: " + "x" * 127):
        dist.AbiComm(NoCtx(), rank=1, world=2, broadcast_id=lambda raw: b"\x00" + b"x" * 127)
    with pytest.raises(_lib.SbgpuError, match="arrived with 5 bytes"):
        dist.AbiComm(NoCtx(), rank=1, world=2, broadcast_id=lambda raw: b"\x01" + b"short")
    with pytest.raises(_lib.SbgpuError, match="could not make an RCCL id"):
        dist.AbiComm(NoCtx(), rank=1, world=2, broadcast_id=lambda raw: b"")
