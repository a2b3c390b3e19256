"""GPU tests of the N > 1 path through the HIP kernels: two ranks (two processes) sharing the one GPU of the
test box, each solving ITS shard of a C3 slice with ShardQuantifier.step -- EM, epilogue, the all-reduce, TPM --
against the single-rank HIP result; and the C-ABI collective (sbgpu_comm_*) at world size 1.  RCCL refuses two
ranks on one device, so the two-rank test reduces through gloo (dist._allreduce_'s host-copy path); the RCCL
calls themselves run in the driver's multi-GPU bench (`SB_COMM=abi`) and in examples/quantify_fragments.cpp."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch
    sys.path.insert(0, %(root)r)
    from strawberry_amd import dist, em, synth
    rank, world, _ = dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    b = synth.make_c3(n_loci=4000, total_frags=2e6, seed=78)
    parts = dist.shard_loci(b.nrow, b.niso, world)
    mine = b.select(parts[rank])
    ctx = em.Context(0)
    solver = em.EmBatchSolver(mine, ctx)
    tot = torch.tensor([mine.n_frags], dtype=torch.int64, device="cuda:0")
    dist.allreduce_sum_(tot)                 # pass-1 normaliser (alignments.cpp:1372)
    q = dist.ShardQuantifier(solver, int(tot.item()), min_isoform_frac=0.0)
    q.step()
    q.step()
    q.finish()
    r = solver.results()
    np.savez(os.path.join(%(out)r, "rank%%d.npz" %% rank), idx=parts[rank], iso_off=mine.iso_off, tpm=r["tpm"],
             fpkm=r["fpkm"], status=r["status"], iters=r["iters"], total_mapped=int(tot.item()))
    dist.barrier()
""")


def test_two_ranks_on_one_gpu_equal_single_rank(tmp_path):
    from strawberry_amd import em, synth
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "out": str(tmp_path)})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", SB_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29641", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    b = synth.make_c3(n_loci=4000, total_frags=2e6, seed=78)
    s = em.EmBatchSolver(b, em.default_context(0))
    s.run_em()
    s.run_abundance(total_mapped_reads=b.n_frags, min_isoform_frac=0.0)
    s.run_tpm()
    ref = s.results()
    got_tpm = np.full(len(ref["tpm"]), np.nan)
    got_it = np.full(b.n_loci, -1)
    for rank in range(2):
        z = np.load(tmp_path / ("rank%d.npz" % rank))
        assert int(z["total_mapped"]) == b.n_frags
        for pos, l in enumerate(z["idx"]):
            n = b.iso_off[l + 1] - b.iso_off[l]
            got_tpm[b.iso_off[l]:b.iso_off[l + 1]] = z["tpm"][z["iso_off"][pos]:z["iso_off"][pos] + n]
            got_it[l] = z["iters"][pos]
    assert not np.isnan(got_tpm).any()
    np.testing.assert_array_equal(got_it, ref["iters"])
    # the shard changes which loci share a wave, never a locus' arithmetic; the FPKM total is summed in another order
    np.testing.assert_allclose(got_tpm, ref["tpm"], rtol=1e-12, atol=0)


def test_abi_comm_world_of_one_and_bad_arguments():
    import ctypes as C
    import torch
    from strawberry_amd import _lib, dist, em
    ctx = em.default_context(0)
    comm = dist.AbiComm(ctx, rank=0, world=1)
    x = torch.tensor([1.5, 2.5], dtype=torch.float64, device="cuda:0")
    n = torch.tensor([7], dtype=torch.int64, device="cuda:0")
    comm.allreduce_sum_(x)
    comm.allreduce_sum_(n)
    torch.cuda.synchronize()
    assert x.tolist() == [1.5, 2.5] and n.item() == 7      # the sum over one rank
    L = _lib.load()
    rank, world = C.c_int(-1), C.c_int(-1)
    assert L.sbgpu_comm_info(comm.h, C.byref(rank), C.byref(world)) == 0 and (rank.value, world.value) == (0, 1)
    h = C.c_void_p()
    assert L.sbgpu_comm_init(ctx.h, 2, 2, None, C.byref(h)) == _lib.SBGPU_EINVAL      # rank out of range
    assert L.sbgpu_comm_init(ctx.h, 0, 2, None, C.byref(h)) == _lib.SBGPU_EINVAL      # world > 1 without an id
    buf = (C.c_uint8 * 128)()
    assert L.sbgpu_comm_unique_id(buf) == 0 and any(buf)
    comm.close()


def test_rccl_runs_at_world_one_when_forced(tmp_path):
    """SBGPU_COMM_FORCE_RCCL=1: a world of ONE goes through the same RCCL calls as a world of eight -- dlopen(librccl.so),
    ncclGetUniqueId, ncclCommInitRank(1), ncclAllReduce of an f64 and an i64 buffer on a NON-default stream, the host-buffer
    forms, ncclCommCount -- so the C-ABI collective is executed code before an 8-GPU node ever runs it.  In a child
    process: the variable is read when the communicator is made, and RCCL stays loaded for the life of a process."""
    import subprocess
    import sys
    import textwrap
    code = textwrap.dedent("""
        import ctypes as C, os, sys
        sys.path.insert(0, %r)
        import torch
        from strawberry_amd import _lib, dist, em
        ctx = em.default_context(0)
        comm = dist.AbiComm(ctx, rank=0, world=1)
        assert comm.rccl_ranks() == 1, comm.rccl_ranks()            # a real RCCL communicator of one rank
        L = _lib.load()
        rank, world = C.c_int(-1), C.c_int(-1)
        assert L.sbgpu_comm_info(comm.h, C.byref(rank), C.byref(world)) == 0 and (rank.value, world.value) == (0, 1)
        side = torch.cuda.Stream(device="cuda:0")
        with torch.cuda.stream(side):
            x = torch.arange(1, 1025, dtype=torch.float64, device="cuda:0") * 0.5
            n = torch.tensor([7, -3, 1 << 40], dtype=torch.int64, device="cuda:0")
            for _ in range(3):
                comm.allreduce_sum_(x)
                comm.allreduce_sum_(n)
        side.synchronize()
        assert x.tolist() == [0.5 * k for k in range(1, 1025)] and n.tolist() == [7, -3, 1 << 40]
        hf = (C.c_double * 2)(1.25, -2.5)
        hi = (C.c_int64 * 2)(5, 1 << 50)
        _lib.check(L.sbgpu_allreduce_sum_f64_host(comm.h, hf, 2), "f64_host")
        _lib.check(L.sbgpu_allreduce_sum_i64_host(comm.h, hi, 2), "i64_host")
        assert list(hf) == [1.25, -2.5] and list(hi) == [5, 1 << 50]
        # round 6: the collectives INSIDE sbgpu_quantify_resident through the same communicator -- ncclAllReduce(max) of the
        # histogram's length (host form), ncclAllReduce(sum) of the int64 histogram on the context's stream, ncclAllReduce(sum)
        # of the FPKM total between the two epilogue kernels: the results of the call without a communicator, bit for bit
        hm = (C.c_int64 * 2)(9, -4)
        _lib.check(L.sbgpu_allreduce_max_i64_host(comm.h, hm, 2), "max_i64_host")
        assert list(hm) == [9, -4]
        import numpy as np
        from strawberry_amd import exonbin as eb, synth
        from strawberry_amd.quantify import quantify_resident
        loci = synth.make_gene_models(50, seed=81)
        hl, pairs = synth.make_fragments(loci, 100, seed=82, noise=0.2)
        rows = [(l, eb.hit_features(lb, rb)) for l, (lb, rb) in zip(hl, pairs)]
        rows = [(l, f) for l, f in rows if f is not None]
        annot, hits = eb.Annotation(loci), eb.Hits([l for l, _ in rows], [f for _, f in rows])
        a = quantify_resident(annot, hits, None, 75, hits.n_hits, ctx=ctx, comm=comm)
        b = quantify_resident(annot, hits, None, 75, hits.n_hits, ctx=ctx, comm=None)
        for k in ("theta", "fpkm", "frac", "tpm", "keep", "status", "iters"):
            assert np.array_equal(a[k], b[k]), k
        assert a["insert"]["mean"] == b["insert"]["mean"] and np.array_equal(a["insert"]["emp_hist"], b["insert"]["emp_hist"])
        assert a["total_fpkm"] == b["total_fpkm"] and a["total_mapped_reads"] == b["total_mapped_reads"] == hits.n_hits
        comm.close()
        print("rccl world-1 ok")
    """ % ROOT)
    env = dict(os.environ, SBGPU_COMM_FORCE_RCCL="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl world-1 ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    # and without the variable a world of one opens no RCCL at all
    from strawberry_amd import dist, em
    comm = dist.AbiComm(em.default_context(0), rank=0, world=1)
    assert comm.rccl_ranks() == 0
    comm.close()


@pytest.mark.gpu
def test_pipelined_steps_equal_the_steps_one_after_the_other():
    """ShardQuantifier's pipelined steps (round 6: the epilogue on a stream of its own beside the next step's EM kernels,
    the kernels' completion joined into THAT stream -- sbgpu_em_run_device_split -- and theta / status / iterations double-
    buffered) leave, after 1, 2, 3 and 8 steps, the bytes of the steps run strictly one after the other; on a plan with
    wide loci too (their runs wait for each other: per-run barrier words), and the split entry refuses a null join stream."""
    import ctypes as C
    import torch
    from strawberry_amd import _lib, dist, em, synth
    ctx = em.default_context(0)
    for make in (lambda: synth.make_c3(n_loci=6000, total_frags=2e7, seed=91),
                 lambda: synth.make_c3t(n_loci=3000, total_frags=1e7, seed=92, n_tail=12)):
        b = make()
        ref = em.EmBatchSolver(b, ctx)
        q0 = dist.ShardQuantifier(ref, 10 ** 7, min_isoform_frac=0.01, pipelined=False, comm=dist.AbiComm(ctx, 0, 1))
        q0.step()
        q0.finish()
        want = {k: getattr(ref, "d_" + k).cpu().numpy().copy() for k in ("theta", "status", "iters", "fpkm", "tpm")}
        for n_steps in (1, 2, 3, 8):
            s = em.EmBatchSolver(b, ctx)
            q = dist.ShardQuantifier(s, 10 ** 7, min_isoform_frac=0.01, pipelined=True, comm=dist.AbiComm(ctx, 0, 1))
            assert q.pipelined
            for _ in range(n_steps):
                q.step()
            q.finish()
            for k, w in want.items():
                assert np.array_equal(getattr(s, "d_" + k).cpu().numpy(), w), (k, n_steps)
    L = ctx.L
    assert L.sbgpu_em_run_device_split(ctx.h, ref.plan.h, ref.d_count.data_ptr(), ref.d_F.data_ptr(), ref.d_theta.data_ptr(),
                                       ref.d_status.data_ptr(), ref.d_iters.data_ptr(), None, None) == _lib.SBGPU_EINVAL
