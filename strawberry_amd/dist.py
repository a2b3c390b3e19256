"""One process per GPU: locus sharding and the single collective of the path.

Loci are independent (the reference runs them on detached threads,
/root/reference/src/alignments.cpp:1782-1804), so ranks never exchange locus
data.  The only cross-locus quantities are the two global normalisers of
/root/reference/src/alignments.cpp:1372 (total mapped reads, before the EM) and
:1821-1829 (sum of FPKM, after it): one all-reduce(sum) of a tiny buffer each --
RCCL over xGMI on GPUs (`backend="nccl"` is RCCL on ROCm), gloo in the CPU tests.
"""
import os

import numpy as np


def env_world():
    """(rank, world_size, local_rank) from the torchrun environment (1 process: 0,1,0)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_process_group(backend=None):
    """Initialise torch.distributed from the environment when WORLD_SIZE > 1."""
    import datetime
    import torch
    import torch.distributed as dist
    rank, world, local_rank = env_world()
    # a rank that leaves the others in a collective must fail the job in minutes, not in the half hour of the default
    tmo = datetime.timedelta(seconds=float(os.environ.get("SB_DIST_TIMEOUT_S", "600")))
    if world > 1 and not dist.is_initialized():
        # (the rendezvous may print banners -- gloo's "Rank 1 is connected to 1 peer ranks" -- on STDOUT, which belongs to the one
        # JSON line of bench.py: file descriptor 1 points at stderr while the group is made)
        import sys
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            local_rank = _init_group(dist, torch, backend, rank, world, local_rank, tmo)
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    return rank, world, local_rank


def _init_group(dist, torch, backend, rank, world, local_rank, tmo):
    if backend is None:
        backend = os.environ.get("SB_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend == "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
    return local_rank


def predicted_iterations(nrow, niso):
    """The plan's shape-only estimate of a locus' EM iteration count (csrc/plan.cpp::predicted_iterations,
    tools/probe_iteration_shape.py): slow when a locus has about as many bins as isoforms, slower with more isoforms."""
    nrow, niso = np.asarray(nrow, np.float64), np.asarray(niso, np.float64)
    r = nrow / np.maximum(niso, 1)
    w = np.select([r < 0.5, r < 0.75, r < 1.25, r < 1.5, r < 2, r < 3, r < 5, r < 10],
                  [0.10, 0.60, 1.00, 0.80, 0.55, 0.35, 0.27, 0.20], 0.17)
    return np.minimum(80.0 * w * (np.minimum(niso, 24) + 18), 1000.0).astype(np.int64)


def shard_loci(nrow, niso, world_size):
    """Greedy LPT partition of loci over ranks by cost = elements x predicted iterations.

    Loci with more than 64 isoforms run on the multi-workgroup kernel, whose iterations cost about
    50x more per element (an exchange between workgroups every iteration): their cost is weighted
    accordingly, so that a human annotation's few hundred such loci spread evenly over the ranks.
    Returns a list of int64 index arrays, one per rank, each sorted ascending so a
    rank's outputs stay in locus order.  Deterministic."""
    nrow, niso = np.asarray(nrow, np.int64), np.asarray(niso, np.int64)
    cost = nrow * niso * np.where(niso > 64, 50, 1) * np.maximum(predicted_iterations(nrow, niso), 1) + 1
    order = np.argsort(-cost, kind="stable")
    load = np.zeros(world_size, np.int64)
    owner = np.empty(len(cost), np.int64)
    # LPT with a tie rule that is independent of the heap implementation
    for l in order:
        r = int(np.argmin(load))
        owner[l] = r
        load[r] += cost[l]
    return [np.nonzero(owner == r)[0].astype(np.int64) for r in range(world_size)]


def _allreduce_(tensor, op):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return tensor
    if tensor.is_cuda and dist.get_backend() == "gloo":
        # test rigs only (several ranks sharing one GPU): gloo reduces a host copy
        host = tensor.cpu()
        dist.all_reduce(host, op=op)
        tensor.copy_(host)
    else:
        dist.all_reduce(tensor, op=op)
    return tensor


def allreduce_sum_(tensor):
    """In-place all-reduce(sum) over the default group; identity for one process."""
    import torch.distributed as dist
    return _allreduce_(tensor, dist.ReduceOp.SUM)


def allreduce_max_(tensor):
    import torch.distributed as dist
    return _allreduce_(tensor, dist.ReduceOp.MAX)


def gather_values(values, rank, world, device=None):
    """Every rank hands in a list of numbers; every rank gets the [world][len(values)] table back (a diagnostic that
    rides on ONE all-reduce(sum) of a zero-padded table: no second collective type, works over RCCL and gloo alike)."""
    import torch
    t = torch.zeros((world, len(values)), dtype=torch.float64, device=device)
    t[rank] = torch.tensor([float(v) for v in values], dtype=torch.float64)
    allreduce_sum_(t)
    return t.cpu().numpy()


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


class AbiComm:
    """The collective behind the C ABI (include/sbgpu.h: sbgpu_comm_*, sbgpu_allreduce_sum_*): RCCL without
    torch.distributed in the data path -- what a C++14 Strawberry driver uses.  Rank 0's id reaches the other ranks
    through `broadcast_id` (a callable bytes -> bytes; default: torch.distributed's object broadcast when a
    process group exists)."""

    def __init__(self, ctx, rank=None, world=None, broadcast_id=None):
        import ctypes as C
        from . import _lib
        self.ctx, self.L = ctx, _lib.load()
        r, w, _ = env_world()
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        ident = None
        if self.world > 1:
            buf = (C.c_uint8 * 128)()
            # what travels: ONE status byte (1 = an id follows, 0 = an error message follows) + the payload.  A failure on
            # rank 0 must still reach the broadcast below, or the other ranks wait for it forever -- and it must not be
            # inferred from the payload's length (a 127-byte message is not an id).
            raw = b""
            if self.rank == 0:
                rc = self.L.sbgpu_comm_unique_id(buf)
                raw = (b"\x01" + bytes(buf)) if rc == 0 else b"\x00" + (self.L.sbgpu_last_error() or b"sbgpu_comm_unique_id failed")
            if broadcast_id is None:
                import torch.distributed as dist
                box = [raw]
                dist.broadcast_object_list(box, src=0)
                raw = box[0]
            else:
                raw = broadcast_id(raw)
            if raw[:1] != b"\x01":
                raise _lib.SbgpuError("rank 0 could not make an RCCL id: " + raw[1:].decode(errors="replace"))
            if len(raw) != 129:
                raise _lib.SbgpuError("the RCCL id arrived with %d bytes instead of 128" % (len(raw) - 1))
            ident = (C.c_uint8 * 128).from_buffer_copy(raw[1:])
        h = C.c_void_p()
        _lib.check(self.L.sbgpu_comm_init(ctx.h, self.rank, self.world, ident, C.byref(h)), "sbgpu_comm_init")
        self.h = h

    def rccl_ranks(self):
        """Ranks in the RCCL communicator behind this one (ncclCommCount); 0: a world of one that never opened RCCL."""
        import ctypes as C
        from . import _lib
        n = C.c_int(-1)
        _lib.check(self.L.sbgpu_comm_rccl_ranks(self.h, C.byref(n)), "sbgpu_comm_rccl_ranks")
        return int(n.value)

    def allreduce_sum_(self, tensor):
        """In place, on the tensor's device buffer, asynchronous on torch's current stream."""
        import ctypes as C
        import torch
        from . import _lib
        stream = C.c_void_p(torch.cuda.current_stream(tensor.device).cuda_stream)
        if tensor.dtype == torch.float64:
            fn, name = self.L.sbgpu_allreduce_sum_f64, "sbgpu_allreduce_sum_f64"
        elif tensor.dtype == torch.int64:
            fn, name = self.L.sbgpu_allreduce_sum_i64, "sbgpu_allreduce_sum_i64"
        else:
            raise TypeError("AbiComm reduces float64 or int64 buffers")
        _lib.check(fn(self.h, tensor.data_ptr(), tensor.numel(), stream), name)
        return tensor

    def close(self):
        if self.h:
            self.L.sbgpu_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostComm:
    """A communicator whose exchange is the caller's (sbgpu_comm_init_host): every all-reduce of the library is staged
    through a host buffer and handed to torch.distributed's default group -- gloo in the CPU tests and wherever several
    ranks share one GPU (RCCL does not serve two ranks on one device); a world of one never calls back."""

    def __init__(self, ctx, rank=None, world=None):
        import ctypes as C
        from . import _lib
        self.L = _lib.load()
        r, w, _ = env_world()
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        self.calls = 0

        def fn(user, buf, n, is_f64, op):
            try:
                import torch
                import torch.distributed as dist
                a = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_double if is_f64 else C.c_int64)), shape=(int(n),))
                t = torch.from_numpy(a)
                red = dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM
                if dist.get_backend() == "nccl":      # (RCCL reduces device buffers only)
                    d = t.cuda()
                    dist.all_reduce(d, op=red)
                    t.copy_(d.cpu())
                else:
                    dist.all_reduce(t, op=red)
                self.calls += 1
                return 0
            except Exception:
                return 1

        self._fn = _lib.HOST_ALLREDUCE_FN(fn)     # (kept alive with the object)
        h = C.c_void_p()
        _lib.check(self.L.sbgpu_comm_init_host(ctx.h, self.rank, self.world, self._fn, None, C.byref(h)), "sbgpu_comm_init_host")
        self.h = h

    def close(self):
        if self.h:
            self.L.sbgpu_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardQuantifier:
    """A rank's share of the quantification: EM -> FPKM/Frac -> all-reduce -> TPM.

    `solver` is an em.EmBatchSolver over this rank's loci.  `total_mapped_reads`
    is the GLOBAL pass-1 count (alignments.cpp:1372), already all-reduced by the
    caller (it precedes the EM: src/estimate.cpp:328).

    Consecutive steps are PIPELINED (round 6): a step's epilogue -- abundance_kernel, the all-reduce, tpm_kernel: a chain of
    three short launches and, at N > 1, a collective's latency -- runs on a stream of its own beside the NEXT step's EM
    kernels, theta / status / iterations double-buffered so that the next EM does not overwrite what the epilogue still
    reads; and the EM kernels' completion is joined into THAT stream (sbgpu_em_run_device_split), so that the next step's
    kernels queue behind this step's on the library's own streams, kind by kind, without a hand-off across streams between
    two steps.  Every step still produces all of its outputs; what overlaps is the tail of one batch with the head of the next,
    as in a run over many batches.  pipelined=False: one stream, one buffer set (the steps strictly one after the other)."""

    def __init__(self, solver, total_mapped_reads, min_isoform_frac=0.01, effective_len_norm=False,
                 insert_mean=0.0, filter_by_expression=True, comm=None, f32=False, pipelined=True):
        """comm: an AbiComm (the C-ABI collective); None: torch.distributed's default group.
        f32: run the EM's fp32 variant (BASELINE config 5; not a parity path), the epilogue stays fp64."""
        self.s = solver
        self.comm = comm
        self.f32 = f32
        self.kw = dict(total_mapped_reads=int(total_mapped_reads), min_isoform_frac=min_isoform_frac,
                       effective_len_norm=effective_len_norm, insert_mean=insert_mean,
                       filter_by_expression=filter_by_expression)
        self.pipelined = bool(pipelined) and not f32
        if self.pipelined:
            torch = solver.torch
            self._epi = torch.cuda.Stream(device=solver.dev)
            self._sets = [(solver.d_theta, solver.d_status, solver.d_iters),
                          (torch.zeros_like(solver.d_theta), torch.full_like(solver.d_status, -1), torch.zeros_like(solver.d_iters))]
            self._read_done = [None, None]     # per buffer set: the event behind the epilogue that last read it
            self._k = 0

    def _epilogue(self):
        s = self.s
        s.run_abundance(**self.kw)          # leaves this rank's sum of kept FPKM in d_sum_fpkm
        # the one collective: 8 bytes over xGMI
        if self.comm is not None:
            self.comm.allreduce_sum_(s.d_sum_fpkm)
        else:
            allreduce_sum_(s.d_sum_fpkm)
        s.run_tpm(s.d_sum_fpkm)

    def step(self):
        s = self.s
        if not self.pipelined:
            if self.f32:
                s.run_em_f32()
                s.theta32_as_f64()
            else:
                s.run_em()
            self._epilogue()
            return
        torch = s.torch
        main = torch.cuda.current_stream(s.dev)
        k = self._k
        self._k ^= 1
        s.d_theta, s.d_status, s.d_iters = self._sets[k]
        if self._read_done[k] is not None:
            main.wait_event(self._read_done[k])     # the epilogue of two steps ago has read this set
        if getattr(s, "d_row_bias", None) is None:
            s.run_em(join_stream=self._epi)      # (the kernels' completion is joined into the epilogue's stream, not into `main`)
        else:
            s.run_em()
            solved = torch.cuda.Event()
            solved.record(main)
            self._epi.wait_event(solved)
        with torch.cuda.stream(self._epi):
            self._epilogue()
            done = torch.cuda.Event()
            done.record(self._epi)
        self._read_done[k] = done

    def finish(self):
        """Wait for the step(s) issued so far; raises if a run failed on the device (SbgpuError)."""
        if self.pipelined:
            self._epi.synchronize()
        self.s.synchronize()
