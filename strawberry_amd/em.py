"""Host-side mirror of the reference's EM call surface over libsbgpu.so.

Reference seam (/root/reference):
    EmSolver em; em.init(niso, n, alpha); em.run(); em._theta     src/estimate.cpp:305-313
    LocusContext::estimate_abundances() -> FPKM / Frac            src/estimate.cpp:279-364
    Sample::procSample tail -> TPM                                src/alignments.cpp:1821-1829

`EmBatchSolver` is the batched form the GPU needs (collect -> one solve ->
epilogue); `EmSolver` keeps the reference's per-locus init/run/_theta shape for
tests and for callers that have a single locus.  torch is used only to own
device memory and streams; all arithmetic happens in the HIP kernels behind the
C ABI.  No CPU fallback: without libsbgpu.so or without a GPU these raise.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import EM_DENOM_ZERO, EM_INIT_EMPTY, EM_MAXITER, EM_OK, SbgpuError  # noqa: F401


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise SbgpuError("no HIP device visible: the EM hot path has no CPU fallback")
    return torch


class Context:
    """One per process / GPU (sbgpu_init)."""

    def __init__(self, device=0):
        self.L = _lib.load()
        h = C.c_void_p()
        _lib.check(self.L.sbgpu_init(int(device), C.byref(h)), "sbgpu_init")
        self.h = h
        self.device = int(device)

    def device_info(self):
        out = (C.c_int64 * 8)()
        _lib.check(self.L.sbgpu_device_info(self.h, out), "sbgpu_device_info")
        return {"cus": out[0], "wave": out[1], "lds_per_cu": out[2], "clock_khz": out[3], "hbm_mib": out[4]}

    def close(self):
        if self.h:
            self.L.sbgpu_finalize(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


class Plan:
    """Shapes of one batch sorted into size classes (sbgpu_plan_create)."""

    def __init__(self, ctx, row_off, iso_off, f_off):
        self.ctx = ctx
        self.row_off = np.ascontiguousarray(row_off, np.int64)
        self.iso_off = np.ascontiguousarray(iso_off, np.int64)
        self.f_off = np.ascontiguousarray(f_off, np.int64)
        n = len(self.row_off) - 1
        if len(self.iso_off) != n + 1 or len(self.f_off) != n + 1:
            raise ValueError("offset arrays must all have n_loci+1 entries")
        h = C.c_void_p()
        _lib.check(ctx.L.sbgpu_plan_create(ctx.h, n, self.row_off.ctypes.data, self.iso_off.ctypes.data,
                                           self.f_off.ctypes.data, C.byref(h)), "sbgpu_plan_create")
        self.h = h
        self.n_loci = n

    def info(self):
        out = (C.c_int64 * 8)()
        _lib.check(self.ctx.L.sbgpu_plan_info(self.h, out), "sbgpu_plan_info")
        return {"n_loci": out[0], "n_rows": out[1], "n_iso": out[2], "n_elem": out[3], "n_classes": out[4],
                "n_stream_loci": out[5], "algorithmic_bytes": out[6]}

    def classes(self):
        cap = 1024
        out = (C.c_int64 * (6 * cap))()
        n = self.ctx.L.sbgpu_plan_classes(self.h, out, cap)
        keys = ("kind", "C", "R", "G", "n_loci", "n_waves")
        return [dict(zip(keys, out[i * 6:(i + 1) * 6])) for i in range(min(n, cap))]

    def locus_kinds(self):
        """int8[n_loci]: 0/1/2 wave kinds, 3 block, 4 tall block, 5 streaming kind."""
        out = np.zeros(max(self.n_loci, 1), np.int8)
        _lib.check(self.ctx.L.sbgpu_plan_locus_kinds(self.h, out.ctypes.data), "sbgpu_plan_locus_kinds")
        return out[:self.n_loci]

    def close(self):
        if self.h:
            self.ctx.L.sbgpu_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EmBatchSolver:
    """Device-resident batch: upload once, run the EM (and epilogue) many times."""

    def __init__(self, batch, ctx=None, device=0, d_F=None):
        """d_F: F already on the device (float64[f_off[-1]], e.g. written by the bin-weight
        kernel); batch.F is not read then."""
        torch = _torch()
        self.torch = torch
        self.ctx = ctx or default_context(device)
        self.dev = torch.device("cuda", self.ctx.device)
        self.batch = batch
        self.plan = Plan(self.ctx, batch.row_off, batch.iso_off, batch.f_off)
        n, n_iso = batch.n_loci, int(batch.iso_off[-1])
        self.d_count = torch.from_numpy(np.ascontiguousarray(batch.count, np.int32)).to(self.dev)
        self.d_F = d_F if d_F is not None else torch.from_numpy(np.ascontiguousarray(batch.F, np.float64)).to(self.dev)
        self.d_length = torch.from_numpy(np.ascontiguousarray(batch.length, np.int32)).to(self.dev)
        self.d_theta = torch.zeros(max(n_iso, 1), dtype=torch.float64, device=self.dev)
        self.d_status = torch.full((max(n, 1),), -1, dtype=torch.int32, device=self.dev)
        self.d_iters = torch.zeros(max(n, 1), dtype=torch.int32, device=self.dev)
        self.d_fpkm = torch.zeros(max(n_iso, 1), dtype=torch.float64, device=self.dev)
        self.d_frac = torch.zeros(max(n_iso, 1), dtype=torch.float64, device=self.dev)
        self.d_tpm = torch.zeros(max(n_iso, 1), dtype=torch.float64, device=self.dev)
        self.d_keep = torch.zeros(max(n_iso, 1), dtype=torch.int32, device=self.dev)
        self.d_sum_fpkm = torch.zeros(1, dtype=torch.float64, device=self.dev)
        self.n_iso = n_iso

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def set_bias(self, d_row_bias, d_iso_bias):
        """BASELINE config 5: factors 2^(row_bias[i] * iso_bias[j]) on the weights, applied by the kernels as they load
        their tiles (sbgpu_em_run_device_bias).  float64 device tensors [total rows] / [total isoforms]; None: off."""
        self.d_row_bias, self.d_iso_bias = d_row_bias, d_iso_bias
        self.d_row_bias32 = None if d_row_bias is None else d_row_bias.to(self.torch.float32)
        self.d_iso_bias32 = None if d_iso_bias is None else d_iso_bias.to(self.torch.float32)

    def run_em(self, join_stream=None):
        """EmSolver::init + run for every locus; asynchronous on torch's current stream.  join_stream (a torch stream): the
        kernels' completion is joined into IT instead of the current stream (sbgpu_em_run_device_split: batch after batch)."""
        L = self.ctx.L
        if join_stream is not None and getattr(self, "d_row_bias", None) is None:
            _lib.check(L.sbgpu_em_run_device_split(self.ctx.h, self.plan.h, self.d_count.data_ptr(), self.d_F.data_ptr(),
                                                   self.d_theta.data_ptr(), self.d_status.data_ptr(), self.d_iters.data_ptr(),
                                                   self._stream(), C.c_void_p(join_stream.cuda_stream)), "sbgpu_em_run_device_split")
            return
        if getattr(self, "d_row_bias", None) is not None:
            _lib.check(L.sbgpu_em_run_device_bias(self.ctx.h, self.plan.h, self.d_count.data_ptr(), self.d_F.data_ptr(),
                                                  self.d_row_bias.data_ptr(), self.d_iso_bias.data_ptr(), self.d_theta.data_ptr(),
                                                  self.d_status.data_ptr(), self.d_iters.data_ptr(), self._stream()),
                       "sbgpu_em_run_device_bias")
            return
        _lib.check(L.sbgpu_em_run_device(self.ctx.h, self.plan.h, self.d_count.data_ptr(), self.d_F.data_ptr(),
                                         self.d_theta.data_ptr(), self.d_status.data_ptr(),
                                         self.d_iters.data_ptr(), self._stream()), "sbgpu_em_run_device")

    def run_em_f32(self):
        """The fp32 variant (BASELINE config 5's tolerance sweep; not a parity path): F and theta in fp32.
        Results land in d_theta32 / d_status / d_iters; theta32_as_f64() feeds the fp64 epilogue."""
        torch = self.torch
        if getattr(self, "d_F32", None) is None:
            self.d_F32 = self.d_F.to(torch.float32)
            self.d_theta32 = torch.zeros(max(self.n_iso, 1), dtype=torch.float32, device=self.dev)
        if getattr(self, "d_row_bias", None) is not None:
            _lib.check(self.ctx.L.sbgpu_em_run_device_bias_f32(
                self.ctx.h, self.plan.h, self.d_count.data_ptr(), self.d_F32.data_ptr(), self.d_row_bias32.data_ptr(),
                self.d_iso_bias32.data_ptr(), self.d_theta32.data_ptr(), self.d_status.data_ptr(), self.d_iters.data_ptr(),
                self._stream()), "sbgpu_em_run_device_bias_f32")
            return
        _lib.check(self.ctx.L.sbgpu_em_run_device_f32(self.ctx.h, self.plan.h, self.d_count.data_ptr(), self.d_F32.data_ptr(),
                                                     self.d_theta32.data_ptr(), self.d_status.data_ptr(),
                                                     self.d_iters.data_ptr(), self._stream()), "sbgpu_em_run_device_f32")

    def theta32_as_f64(self):
        """Copy the fp32 run's theta into d_theta (fp64) so that the epilogue kernels can follow."""
        self.d_theta.copy_(self.d_theta32)

    def set_timing(self, on=True):
        """Timing events around the EM kernels (off by default; bench.py turns them on for its probe steps)."""
        _lib.check(self.ctx.L.sbgpu_set_timing(self.ctx.h, int(bool(on))), "sbgpu_set_timing")

    def last_phase_ms(self):
        """Device time of each phase of the wave kind in the last run_em (timing on); [] for one phase."""
        ms = (C.c_float * 8)()
        n = self.ctx.L.sbgpu_em_last_phase_ms(self.ctx.h, ms, 8)
        if n < 0:
            _lib.check(n, "sbgpu_em_last_phase_ms")
        return [float(ms[i]) for i in range(n)]

    def synchronize(self):
        """Wait for this solver's stream and report a failed run (a wide-locus barrier that timed out leaves
        loci unsolved: that must raise, not hand stale theta to FPKM/TPM and the all-reduce)."""
        _lib.check(self.ctx.L.sbgpu_synchronize(self.ctx.h, self._stream()), "sbgpu_synchronize")

    def last_kernel_ms(self):
        """Device time of the last run_em per kernel kind:
        [wave half tile, wave base tile, wave double tile, block, tall block, stream]."""
        ms = (C.c_float * 6)()
        _lib.check(self.ctx.L.sbgpu_em_last_kernel_ms(self.ctx.h, ms), "sbgpu_em_last_kernel_ms")
        return [float(x) for x in ms]

    def run_abundance(self, total_mapped_reads, effective_len_norm=False, insert_mean=0.0,
                      filter_by_expression=True, min_isoform_frac=0.01):
        """theta -> FPKM, Frac, keep and this rank's sum of kept FPKM (d_sum_fpkm)."""
        p = _lib.sbgpu_abundance_params_t(int(total_mapped_reads), int(effective_len_norm),
                                          int(filter_by_expression), 0, float(insert_mean),
                                          float(min_isoform_frac))
        _lib.check(self.ctx.L.sbgpu_abundance_device(
            self.ctx.h, self.plan.h, self.d_theta.data_ptr(), self.d_status.data_ptr(), self.d_length.data_ptr(),
            C.byref(p), self.d_fpkm.data_ptr(), self.d_frac.data_ptr(), self.d_keep.data_ptr(),
            self.d_sum_fpkm.data_ptr(), self._stream()), "sbgpu_abundance_device")

    def run_tpm(self, d_total_fpkm=None):
        """TPM from the (all-reduced) total FPKM; defaults to this rank's own sum."""
        tot = self.d_sum_fpkm if d_total_fpkm is None else d_total_fpkm
        _lib.check(self.ctx.L.sbgpu_tpm_device(self.ctx.h, self.n_iso, self.d_fpkm.data_ptr(),
                                               self.d_keep.data_ptr(), tot.data_ptr(), self.d_tpm.data_ptr(),
                                               self._stream()), "sbgpu_tpm_device")

    def results(self):
        """-> dict of host numpy arrays (synchronises; raises if the last run failed)."""
        self.synchronize()
        self.torch.cuda.synchronize(self.dev)
        n, k = self.batch.n_loci, self.n_iso
        return {
            "theta": self.d_theta[:k].cpu().numpy(),
            "status": self.d_status[:n].cpu().numpy(),
            "iters": self.d_iters[:n].cpu().numpy(),
            "fpkm": self.d_fpkm[:k].cpu().numpy(),
            "frac": self.d_frac[:k].cpu().numpy(),
            "keep": self.d_keep[:k].cpu().numpy(),
            "tpm": self.d_tpm[:k].cpu().numpy(),
            "sum_fpkm": float(self.d_sum_fpkm.cpu().numpy()[0]),
        }


def em_batch_host(batch, ctx=None, device=0):
    """sbgpu_em_batch: host buffers in, host buffers out (plan + H2D + solve + D2H)."""
    ctx = ctx or default_context(device)
    n, n_iso = batch.n_loci, int(batch.iso_off[-1])
    row_off = np.ascontiguousarray(batch.row_off, np.int64)
    iso_off = np.ascontiguousarray(batch.iso_off, np.int64)
    f_off = np.ascontiguousarray(batch.f_off, np.int64)
    count = np.ascontiguousarray(batch.count, np.int32)
    F = np.ascontiguousarray(batch.F, np.float64)
    b = _lib.sbgpu_batch_t(n, row_off.ctypes.data, iso_off.ctypes.data, f_off.ctypes.data,
                           count.ctypes.data if count.size else None, F.ctypes.data if F.size else None)
    theta = np.zeros(max(n_iso, 1), np.float64)
    status = np.full(max(n, 1), -1, np.int32)
    iters = np.zeros(max(n, 1), np.int32)
    _lib.check(ctx.L.sbgpu_em_batch(ctx.h, C.byref(b), theta.ctypes.data, status.ctypes.data, iters.ctypes.data),
               "sbgpu_em_batch")
    return theta[:n_iso], status[:n], iters[:n]


class EmSolver:
    """Per-locus adapter with the reference's shape (include/estimate.hpp:230-257):

        em = EmSolver(); ok = em.init(niso, n, alpha); ran = em.run(); em._theta

    init() returns False exactly when the reference's does (no row has a weight
    > 1e-5); run() returns False on a zero row denominator, leaving _theta at
    theta0.  Both are decided on the GPU; init() triggers the solve and caches it.
    """

    def __init__(self, ctx=None, device=0):
        self._ctx = ctx
        self._device = device
        self._theta = []
        self._status = None
        self.iters = 0

    def init(self, num_iso, count, model):
        from .synth import from_loci
        count = np.asarray(count, np.int32).reshape(-1)
        F = np.asarray(model, np.float64).reshape(len(count), num_iso)
        theta, status, iters = em_batch_host(from_loci([(count, F)]), self._ctx, self._device)
        self._theta = list(theta)
        self._status = int(status[0])
        self.iters = int(iters[0])
        return self._status != EM_INIT_EMPTY

    def run(self):
        if self._status is None or self._status == EM_INIT_EMPTY:
            return False
        return self._status != EM_DENOM_ZERO
