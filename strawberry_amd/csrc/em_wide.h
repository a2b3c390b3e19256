// strawberry_amd/csrc/em_wide.h -- EmSolver for loci too wide for a register tile of one workgroup
// (more than 64 isoforms, or more rows than a 256-lane tile holds): SEVERAL workgroups per locus.
//
// The streaming kernel (em_device.h) gives such a locus one workgroup and re-reads F from L2 in every
// iteration: ~90 us per iteration for a 1160 x 194 locus, i.e. 90 ms for the 1000-iteration cap, and a
// human annotation has tens of such loci.  Here the rows of the locus are dealt to G workgroups (CUs);
// each keeps its rows of F in registers for all iterations -- F is read from HBM once -- and an iteration
// costs one exchange: every workgroup writes its partial column sums (niso doubles) to a small buffer in
// global memory, a counter barrier among the G workgroups of the locus, every workgroup adds the G
// partials in the same order.  All G workgroups thus hold bitwise identical theta and take identical
// decisions (convergence, zero denominator), so no second barrier is needed; two buffers alternate.
//
// Launched cooperatively (hipLaunchCooperativeKernel): the barrier needs all workgroups of a locus
// resident.  A barrier that does not complete within kWideSpinLimit polls gives up and raises an error
// flag instead of hanging the GPU.
//
// Same arithmetic as the other kernels: F' = F * scale through phi, fast_div, fp64 flush mode.
#pragma once

#include "em_device.h"

namespace sb {

constexpr int kWideThreads = 512;              // 8 waves, 2 per SIMD: up to 256 VGPRs each
constexpr int kWideWaves = kWideThreads / 64;
// rows of F a wave keeps in registers (x NSLOT columns per lane): bounded by 256 VGPRs with n_i, flags, temporaries
constexpr int wide_rows(int nslot) { return nslot <= 2 ? 16 : (nslot <= 4 ? 12 : 8); }
constexpr unsigned kWideSpinLimit = 1u << 24;  // polls of a barrier before giving up (~ seconds)

struct WideDesc {
   int32_t locus;
   int32_t first_block; // in this launch
   int32_t n_blocks;    // G
   int32_t rows_per_block;
   int64_t buf_off;     // doubles: start of this locus' 2 x G x (npad + 2) exchange buffers
   int32_t barrier;     // index of its counter
   int32_t npad;
};

struct WideArgs {
   EmArgs a;
   const WideDesc *table;
   int32_t n_desc;
   double *bufs;
   unsigned *barriers; // zeroed before the launch
   int32_t *error;     // set to 1 when a barrier timed out
};

// All G workgroups of the locus have written exchange number `round` (1-based).
// Hand-off form (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility", first
// row of the sc1 table): every byte of the partials is stored with an agent-scope relaxed atomic store
// (global_store sc1: write-through, no line kept in this XCD's non-coherent L2) and loaded with an agent-scope
// relaxed atomic load (sc1: never served from the vector L1); every storing wave waits for its stores
// (s_waitcnt vmcnt(0)), a workgroup barrier, ONE lane adds to the locus' counter (agent-scope atomic) and polls
// it with sc1 loads, a workgroup barrier, then the loads.  No L2 write-back / invalidate fences: those cost
// more than the whole iteration (measured 10 of 13.5 us with release / acquire fences).
__device__ __forceinline__ bool wide_barrier(unsigned *counter, unsigned G, unsigned round, int32_t *error)
{
   asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's partial stores have left
   __syncthreads();
   __shared__ int ok;
   if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      const unsigned target = G * round;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
         __builtin_amdgcn_s_sleep(1);
         if (++spins > kWideSpinLimit || __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_store(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
         }
      }
      ok = __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
   }
   __syncthreads();
   return ok != 0;
}

// NSLOT columns per lane (64 lanes per row): niso <= 64 * NSLOT; R = wide_rows(NSLOT) rows per wave
template <int NSLOT>
__global__ __launch_bounds__(kWideThreads) void em_wide_kernel(WideArgs g)
{
   constexpr int R = wide_rows(NSLOT);
   extern __shared__ double s_dyn[]; // phi[npad] | theta[npad] | scale[npad] | accw[kWideWaves][npad + 2]
   const EmArgs &a = g.a;
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   set_fp64_flush_denormals();
   // my locus
   int di = 0;
   for (int k = 1; k < g.n_desc; ++k)
      if (g.table[k].first_block <= (int)blockIdx.x) di = k;
   const WideDesc d = g.table[di];
   const int w = (int)blockIdx.x - d.first_block, G = d.n_blocks;
   const int locus = d.locus;
   const int64_t r0 = a.row_off[locus];
   const int nrow = (int)(a.row_off[locus + 1] - r0);
   const int64_t iso_base = a.iso_off[locus];
   const int niso = (int)(a.iso_off[locus + 1] - iso_base);
   const double *Fg = a.F + a.f_off[locus];
   const int npad = d.npad, nv = npad + 2; // exchanged vector: npad column values + 2 scalars
   double *phi = s_dyn, *theta = s_dyn + npad, *scale = s_dyn + 2 * npad, *accw = s_dyn + 3 * npad;
   double *bufs = g.bufs + d.buf_off;
   unsigned *counter = g.barriers + d.barrier;
   unsigned round = 0;

   // ---- my rows: block w owns rows [w * rows_per_block, ...), wave v of it the rows v, v + 16, ...
   const int row_lo = w * d.rows_per_block;
   const int row_hi = min(nrow, row_lo + d.rows_per_block);
   double F[R][NSLOT], nn[R];
   bool act[R];
   double tot = 0.0;
   int kept = 0;
#pragma unroll
   for (int r = 0; r < R; ++r) {
      const int i = row_lo + wave + r * kWideWaves;
      const bool valid = i < row_hi;
      const int ic = valid ? i : (nrow > 0 ? min(i, nrow - 1) : 0);
      double cnt = (valid && nrow > 0) ? (double)a.count[r0 + ic] : 0.0;
      double mx = 0.0;
#pragma unroll
      for (int k = 0; k < NSLOT; ++k) {
         const int j = lane + 64 * k;
         double x = 0.0;
         if (valid && j < niso) x = Fg[(int64_t)ic * niso + j];
         F[r][k] = x;
         mx = fmax(mx, x);
      }
      mx = fmax(mx, xor_get<1>(mx));
      mx = fmax(mx, xor_get<2>(mx));
      mx = fmax(mx, xor_get<4>(mx));
      mx = fmax(mx, xor_get<8>(mx));
      mx = fmax(mx, xor_get<16>(mx));
      mx = fmax(mx, xor_get<32>(mx));
      const bool keep = valid && mx > kRowEps; // estimate.cpp:380
      act[r] = keep;
      nn[r] = keep ? cnt : 0.0;
      if (!keep) {
#pragma unroll
         for (int k = 0; k < NSLOT; ++k) F[r][k] = 0.0;
      }
      tot += cnt; // theta_0 counts ALL rows (:374-375)
      kept |= keep ? 1 : 0;
   }
   // exchange: every lane contributes NSLOT column values + 2 scalars through accw, then the blocks
   // through `bufs`; returns the sums in accw[0 .. npad + 2) of wave slot 0 (identical in all blocks)
   auto exchange = [&](const double (&col)[NSLOT], double s0, double s1) -> bool {
      double *mine = accw + wave * nv;
#pragma unroll
      for (int k = 0; k < NSLOT; ++k)
         if (lane + 64 * k < npad) mine[lane + 64 * k] = col[k];
      if (lane == 0) {
         mine[npad] = s0;
         mine[npad + 1] = s1;
      }
      __syncthreads();
      ++round;
      double *out = bufs + ((size_t)(round & 1) * G + w) * nv;
      for (int j = tid; j < nv; j += kWideThreads) {
         double s = 0.0;
         for (int v = 0; v < kWideWaves; ++v) s += accw[v * nv + j];
         if (G > 1) __hip_atomic_store(out + j, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // sc1 store
         else accw[j] = s; // (wave 0's slot; safe: every thread owns its j)
      }
      if (G == 1) {
         __syncthreads();
         return true;
      }
      if (!wide_barrier(counter, (unsigned)G, round, g.error)) return false;
      const double *in = bufs + (size_t)(round & 1) * G * nv;
      for (int j = tid; j < nv; j += kWideThreads) {
         double s = 0.0;
         for (int v = 0; v < G; ++v) s += __hip_atomic_load(in + (size_t)v * nv + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // sc1 load
         accw[j] = s;
      }
      __syncthreads();
      return true;
   };

   // ---- EmSolver::init: total count and "any row kept" over the whole locus
   {
      double col[NSLOT];
#pragma unroll
      for (int k = 0; k < NSLOT; ++k) col[k] = 0.0;
      // per-wave scalars: lane 0 carries the wave's totals
      double t = tot; // every lane of the wave holds the same tot (rows are per wave)
      if (!exchange(col, t, (double)kept)) return;
   }
   const double theta0 = accw[npad] / (double)niso; // :375
   const bool any_kept = accw[npad + 1] != 0.0;
   __syncthreads();
   for (int j = tid; j < npad; j += kWideThreads) {
      const double t = (j < niso) ? theta0 : 0.0;
      theta[j] = t;
      scale[j] = 1.0;
      phi[j] = t;
   }
   __syncthreads();
   if (!any_kept) { // init() == false (:391)
      if (w == 0) {
         if (tid == 0) {
            a.status[locus] = kStInitEmpty;
            a.iters[locus] = 0;
         }
         for (int j = tid; j < niso; j += kWideThreads) a.theta[iso_base + j] = theta0;
      }
      return;
   }
   int32_t st = kStMaxIter;
   int it = 0;
   bool theta0_out = false;
   while (it < kMaxIter) {
      double ph[NSLOT], acc[NSLOT];
#pragma unroll
      for (int k = 0; k < NSLOT; ++k) {
         ph[k] = (lane + 64 * k < npad) ? phi[lane + 64 * k] : 0.0;
         acc[k] = 0.0;
      }
      int zf = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
         double part = 0.0;
#pragma unroll
         for (int k = 0; k < NSLOT; ++k) part = __builtin_fma(F[r][k], ph[k], part); // :450
         const double dd = wave_group_sum<64>(part);
         zf |= (act[r] && dd == 0.0) ? 1 : 0; // :451
         double wgt = fast_div(nn[r], dd);
         wgt = act[r] ? wgt : 0.0;
#pragma unroll
         for (int k = 0; k < NSLOT; ++k) acc[k] = __builtin_fma(wgt, F[r][k], acc[k]);
      }
      if (!exchange(acc, (double)zf, 0.0)) return;
      const bool dz = accw[npad] != 0.0;
      // next_theta, identical in every block; column j is owned by thread j (and j + 1024 ...)
      double d2 = 0.0;
      for (int j = tid; j < niso; j += kWideThreads) {
         const double nt = phi[j] * accw[j]; // :454-464
         const double df = nt - theta[j];
         d2 = __builtin_fma(df, df, d2);
         accw[nv + j] = nt; // wave 1's slot is free now: next_theta
      }
      d2 = wave_group_sum<64>(d2);
      __shared__ double s_part[kWideWaves];
      if (lane == 0) s_part[wave] = d2;
      __syncthreads();
      d2 = 0.0;
      for (int v = 0; v < kWideWaves; ++v) d2 += s_part[v];
      ++it;
      if (dz) { // run() == false, _theta untouched (:451-453)
         st = kStDenomZero;
         theta0_out = true;
         break;
      }
      if (it == 1) {
         // scale_j = 1 / (column sum over kept rows), a zero column stays zero (:466-478)
         double cs[NSLOT];
#pragma unroll
         for (int k = 0; k < NSLOT; ++k) {
            cs[k] = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) cs[k] += F[r][k];
         }
         // next_theta lives in wave 1's slot of accw, which the exchange overwrites: park it in theta's
         // shadow (scale[] is free until we set it below)
         for (int j = tid; j < niso; j += kWideThreads) scale[j] = accw[nv + j];
         __syncthreads();
         if (!exchange(cs, 0.0, 0.0)) return;
         for (int j = tid; j < niso; j += kWideThreads) {
            const double nt = scale[j];
            const double s = accw[j];
            scale[j] = (s == 0.0) ? 0.0 : 1.0 / s;
            accw[nv + j] = nt;
         }
         __syncthreads();
      }
      if (sqrt(d2) < kThetaLimit) { // :479-480, theta NOT updated
         st = kStOk;
         break;
      }
      for (int j = tid; j < niso; j += kWideThreads) theta[j] = accw[nv + j]; // :481
      __syncthreads();
      for (int j = tid; j < npad; j += kWideThreads) phi[j] = (j < niso) ? theta[j] * scale[j] : 0.0;
      __syncthreads();
   }
   if (w == 0) {
      if (tid == 0) {
         a.status[locus] = st;
         a.iters[locus] = it;
      }
      for (int j = tid; j < niso; j += kWideThreads) a.theta[iso_base + j] = theta0_out ? theta0 : theta[j];
   }
}

hipError_t launch_wide(int nslot, const WideArgs &g, int n_blocks, size_t lds_bytes, hipStream_t s);

} // namespace sb
