// strawberry_amd/csrc/em_wide.h -- EmSolver for loci too wide for a register tile of one workgroup
// (more than 64 isoforms, or more rows than a 256-lane tile holds): SEVERAL workgroups per locus.
//
// The streaming kernel (em_device.h) gives such a locus one workgroup and re-reads F from L2 in every
// iteration: ~90 us per iteration for a 1160 x 194 locus, i.e. 90 ms for the 1000-iteration cap, and a
// human annotation has tens to hundreds of such loci.  Here the rows of the locus are dealt to G workgroups
// (CUs); each keeps its rows of F in registers for all iterations -- F is read from HBM once -- and an iteration
// costs one exchange: every workgroup publishes its partial column sums (one double per column) as tagged 16-byte
// granules in global memory, and the thread that owns column j of a workgroup reads the G partials of column j
// (all in flight together, re-reading what has not arrived yet) and adds them in workgroup order.  All G
// workgroups thus hold bitwise identical theta and take identical decisions (convergence, zero denominator);
// two buffers alternate, there is no barrier object at all.
//
// Round 3 layout.  A wave is a 2-D grid of lanes like the tile kernels' groups (em_device.h, "matrix lane map"):
// CL = 16, 32 or 64 COLUMN lanes on the lane bits 4 5 0 1 (3 (2)) -- a row's denominator is summed over them by a
// PAIR of v_mfma_f64_4x4x4 (+ one or two rotations of the 16-lane row) -- and 4, 2 or 1 ROW lanes on the bits
// left (3 2): the column sums of a wave take two, one or no rotation step per column.  A lane keeps R rows x CPL
// columns (96 doubles: 192 of its 256 registers; the old layout kept 40-64 and spent 12 reduction steps per row on
// all 64 lanes), the counts of its rows sit in LDS, "row kept" is a lane mask per row slot in scalar registers.
// Two workgroup barriers per iteration (partials visible / phi visible): the zero-denominator flag rides in the
// granules' tag words and the shares of ||next - theta||^2 are read after the second barrier, with theta's update
// already written -- the thread that owns a column keeps the previous theta in a register, so the pre-update value
// the reference returns on convergence (estimate.cpp:479-481) is still there.
//
// Launched cooperatively (hipLaunchCooperativeKernel): the exchange needs all workgroups of a locus resident.
// A partial that does not arrive within kWideSpinLimit sweeps raises an error flag instead of hanging the GPU.
//
// Same arithmetic as the other kernels: F' = F * scale through phi, reciprocals of four rows from one v_rcp_f64,
// fp64 flush mode.
#pragma once

#include "em_device.h"

namespace sb {

constexpr int kWideWaves = 8;                  // 2 per SIMD: up to 256 VGPRs each (the sums over the waves are written as trees of 8)
constexpr int kWideThreads = 64 * kWideWaves;
constexpr unsigned kWideSpinLimit = 1u << 20;  // sweeps of an exchange's partials before giving up (~ seconds)
constexpr int kWideSweep = 8;                  // granules a thread has in flight per sweep

// (log2 column lanes, columns per lane, rows per lane, rows that share one v_rcp_f64): 72-80 doubles of F per lane --
// what fits beside the ~90 registers the iteration itself needs (phi and the partial sums of the lane's columns, a
// block's denominators and reciprocals, the owner thread's theta / phi / scale), without a spill
struct WideLayout {
   int lb_cl, cpl, r, rblk;
};
constexpr int kWideLayouts = 7;
constexpr WideLayout wide_layout(int id)
{
   return id == 0 ? WideLayout{4, 4, 16, 4} : id == 1 ? WideLayout{4, 6, 12, 4} : id == 2 ? WideLayout{4, 8, 10, 2}
        : id == 3 ? WideLayout{5, 6, 12, 4} : id == 4 ? WideLayout{5, 8, 10, 2} : id == 5 ? WideLayout{6, 6, 12, 4}
                                                                                           : WideLayout{6, 8, 10, 2};
}
constexpr int wide_cols(int id) { return (1 << wide_layout(id).lb_cl) * wide_layout(id).cpl; } // 64 ... 512
constexpr int wide_rows_per_block(int id) { return kWideWaves * (64 >> wide_layout(id).lb_cl) * wide_layout(id).r; }
constexpr int kWideMaxCols = wide_cols(kWideLayouts - 1);
inline int wide_layout_for(int64_t niso)
{
   for (int id = 0; id < kWideLayouts; ++id)
      if (niso <= wide_cols(id)) return id;
   return -1;
}
// dynamic LDS of a workgroup: phi[npad] | accw[waves][npad] | part[waves] | misc[8] | zf[16 ints] | counts[R][threads] doubles
// | stage[2 npad] granules (the two-level exchange)
constexpr size_t wide_lds_bytes(int id)
{
   return (size_t)(wide_cols(id) * (1 + kWideWaves) + kWideWaves + 8) * sizeof(double) + 16 * sizeof(int) +
          (size_t)wide_layout(id).r * kWideThreads * sizeof(double) + (size_t)2 * wide_cols(id) * 16;
}
// columns per slice of the two-level exchange: the smallest power of two m with m * G >= npad (m * G < 2 npad)
inline int wide_lb_slice(int npad, int G)
{
   int lb = 0;
   while (((int64_t)G << lb) < npad) ++lb;
   return lb;
}

struct WideDesc {
   int32_t locus;
   int32_t first_block; // in this launch
   int32_t n_blocks;    // G
   int32_t rows_per_block;
   int64_t buf_off;     // doubles: start of this locus' 2 x G x npad exchange granules (16 bytes each)
   int32_t layout;      // wide_layout id
   int32_t npad;        // wide_cols(layout)
   int32_t lb_slice;    // two-level exchange (G > kWideSweep): a workgroup sums the 2^lb_slice columns from its index << lb_slice
   int32_t pad_;
};

struct WideArgs {
   EmArgs a;
   const WideDesc *table;
   int32_t n_desc;
   double *bufs;       // zeroed when the plan is made
   unsigned epoch;     // distinguishes this run's granules from those an earlier run left in the buffers
   int32_t *error;     // set to 1 when a partial never arrived
};

// The exchange of partial column sums among the G workgroups of a locus, once per iteration.
// Every partial travels as a 16-byte granule {value, tag, ~tag ^ flag} written by ONE global_store_dwordx4 sc1
// (write-through: nothing stays in this XCD's non-coherent L2) and read by global_load_dwordx4 sc1 (never served
// from the vector L1): a reader that finds the current exchange's tag beside a value has the value -- no counter,
// no flag, no second round trip (MI355X_MICROARCH.md: "handoff-1to1", data-tagged granules, 0.8-1.0 us against
// 1.7-2.5x that for a separate flag; 16-byte sc1 granules are observed untorn on gfx950 -- and a torn one would
// fail the tag / ~tag test and be read again).  tag = run epoch * 2048 + exchange number, so nothing an earlier
// run left in the buffers can pass.  The lowest bit of the fourth word carries the producer's "a kept row's
// denominator was zero" flag (estimate.cpp:451).  Two buffers alternate by the exchange's parity: a workgroup can
// run at most one exchange ahead of the slowest one, because it needs everyone's partials of an exchange to leave it.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void granule_store(void *p, double v, unsigned round, unsigned flag)
{
   u32x4 g;
   g.x = (unsigned)__double2loint(v);
   g.y = (unsigned)__double2hiint(v);
   g.z = round;
   g.w = ~round ^ flag;
   asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(g) : "memory");
}
__device__ __forceinline__ bool granule_ready(const u32x4 &g, unsigned round) { return g.z == round && (g.w ^ ~round) <= 1u; }
__device__ __forceinline__ unsigned granule_flag(const u32x4 &g, unsigned round) { return (g.w ^ ~round) & 1u; }
__device__ __forceinline__ double granule_value(const u32x4 &g) { return __hiloint2double((int)g.y, (int)g.x); }

// all-reduce of N values over the column lanes of a wave (lane bits 4 5 0 1, then 3, then 2)
template <int LB_CL, int N, int NA>
__device__ __forceinline__ void wide_col_lanes_sum(double (&x)[NA])
{
#pragma unroll
   for (int v = 0; v < N; ++v) x[v] = __builtin_amdgcn_mfma_f64_4x4x4f64(x[v], 1.0, 0.0, 0, 0, 0);
#pragma unroll
   for (int v = 0; v < N; ++v) x[v] = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, x[v], 0.0, 0, 0, 0);
   if (LB_CL >= 5) {
#pragma unroll
      for (int v = 0; v < N; ++v) x[v] += row_ror8(x[v]);
   }
   if (LB_CL >= 6) {
#pragma unroll
      for (int v = 0; v < N; ++v) x[v] += row_ror4(x[v]); // symmetric under the rotation by 8 by now
   }
}
template <int LB_CL>
__device__ __forceinline__ double wide_col_lanes_max(double x)
{
   x = fmax(x, xor_get<1>(x));
   x = fmax(x, xor_get<2>(x));
   x = fmax(x, xor_get<16>(x));
   x = fmax(x, xor_get<32>(x));
   if (LB_CL >= 5) x = fmax(x, xor_get<8>(x));
   if (LB_CL >= 6) x = fmax(x, xor_get<4>(x));
   return x;
}
// all-reduce of N values over the row lanes of a wave (the lane bits 3 2 the column lanes leave)
template <int LB_CL, int N>
__device__ __forceinline__ void wide_row_lanes_sum(double (&x)[N])
{
   if (LB_CL == 4) {
#pragma unroll
      for (int v = 0; v < N; ++v) x[v] += row_ror8(x[v]);
#pragma unroll
      for (int v = 0; v < N; ++v) x[v] += row_ror4(x[v]);
   }
   if (LB_CL == 5) {
#pragma unroll
      for (int v = 0; v < N; ++v) x[v] = xor_sum<4, false>(x[v]);
   }
}

template <int LB_CL, int CPL, int R, int RBLK>
__device__ __forceinline__ void em_wide_body(const WideArgs &g, const int di)
{
   constexpr int CL = 1 << LB_CL, GR = 64 / CL, NPAD = CL * CPL;
   constexpr int ROWSTEP = kWideWaves * GR; // rows of the block between a lane's consecutive row slots
   static_assert(R % RBLK == 0 && (RBLK == 4 || RBLK == 2 || RBLK == 1) && CPL % 2 == 0 && NPAD <= kWideThreads, "layout");
   extern __shared__ double s_dyn[];
   double *phi = s_dyn;                       // [NPAD] theta_j * scale_j of the iteration about to run
   double *accw = phi + NPAD;                 // [kWideWaves][NPAD] the waves' partial column sums
   double *s_part = accw + kWideWaves * NPAD; // [kWideWaves] the owner waves' shares of ||next - theta||^2
   double *s_misc = s_part + kWideWaves;      // 0: total count, 1: rows kept, 3: abort, 4 + parity: zero-denominator flag of an iteration
   int *s_zf = (int *)(s_misc + 8);           // [kWideWaves] "a kept row of this wave had a zero denominator"
   double *s_cnt = (double *)(s_zf + 16);     // [R][kWideThreads] counts of the lanes' rows as doubles (0 for rows not kept)
   const EmArgs &a = g.a;
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   set_fp64_flush_denormals();
   const WideDesc d = g.table[di];
   const int w = (int)blockIdx.x - d.first_block, G = d.n_blocks;
   const int locus = d.locus;
   const int64_t r0 = a.row_off[locus];
   const int nrow = (int)(a.row_off[locus + 1] - r0);
   const int64_t iso_base = a.iso_off[locus];
   const int niso = (int)(a.iso_off[locus + 1] - iso_base);
   const double *Fg = a.F + a.f_off[locus];
   char *bufs = (char *)(g.bufs + d.buf_off);
   int32_t *g_err = g.error;
   unsigned round = g.epoch << 11; // tags = epoch * 2048 + exchange number (at most 1002 exchanges per run)

   // ---- the lane's place in the wave's grid: column lane c (bits 4 5 0 1 [3 [2]]), row lane b (the bits left)
   const int c16 = ((lane >> 4) & 3) | ((lane & 3) << 2);
   const int c = LB_CL == 4 ? c16 : (LB_CL == 5 ? (c16 | (((lane >> 3) & 1) << 4)) : (c16 | (((lane >> 3) & 1) << 4) | (((lane >> 2) & 1) << 5)));
   const int b = LB_CL == 4 ? ((lane >> 2) & 3) : (LB_CL == 5 ? ((lane >> 2) & 1) : 0);

   // ---- my rows: block w owns rows [w * rows_per_block, ...); the lane's slot r is row r * ROWSTEP + wave * GR + b of it
   const int row_lo = w * d.rows_per_block;
   const int row_hi = min(nrow, row_lo + d.rows_per_block);
   double F[R][CPL];
   bool keep[R]; // lane masks in scalar registers
   double csum[CPL];
#pragma unroll
   for (int k = 0; k < CPL; ++k) csum[k] = 0.0;
   double tot = 0.0;
   int kept = 0;
#pragma unroll
   for (int r = 0; r < R; ++r) {
      const int i = row_lo + r * ROWSTEP + wave * GR + b;
      const bool valid = i < row_hi;
      const int ic = valid ? i : row_lo; // (row_lo < nrow: every block owns at least one row)
      int cnt = a.count[r0 + ic];
      cnt = valid ? cnt : 0;
      double mx = 0.0;
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
         const int j = k * CL + c; // the lane's k-th column: consecutive column lanes, consecutive columns (LDS banks)
         const bool ok = valid && j < niso;
         double x = Fg[(int64_t)ic * niso + (j < niso ? j : 0)]; // unconditional load, then select: no branch per element
         x = ok ? x : 0.0;
         F[r][k] = x;
         mx = fmax(mx, x);
      }
      mx = wide_col_lanes_max<LB_CL>(mx);
      keep[r] = valid && mx > kRowEps; // estimate.cpp:380
      if (!keep[r]) {
#pragma unroll
         for (int k = 0; k < CPL; ++k) F[r][k] = 0.0;
      }
      s_cnt[r * kWideThreads + tid] = keep[r] ? (double)cnt : 0.0;
      if (c == 0) tot += (double)cnt; // theta_0 counts ALL rows (:374-375); one lane per row
      kept |= keep[r] ? 1 : 0;
#pragma unroll
      for (int k = 0; k < CPL; ++k) csum[k] += F[r][k];
   }

   // per block of row slots, for this wave: 0 = no kept row in any lane, 1 = a kept row in every lane, 2 = mixed
   int blk_kind[R / RBLK];
#pragma unroll
   for (int rb = 0; rb < R; rb += RBLK) {
      bool any = false, all = true;
#pragma unroll
      for (int q = 0; q < RBLK; ++q) {
         any |= keep[rb + q];
         all &= keep[rb + q];
      }
      blk_kind[rb / RBLK] = !wave_any(any) ? 0 : (wave_any(!all) ? 2 : 1);
   }

   // The exchange: the thread that owns item `tid` (tid < n_items) hands in its workgroup's partial `s` and the
   // workgroup's flag bit and gets the sum of the G workgroups' partials, added in workgroup order (identical in
   // every workgroup), and the OR of their flags.  Called by ALL threads.  G == 1: nothing travels.
   //   up to kWideSweep workgroups: the owner reads the G partials of its item itself, all in flight together;
   //   more: two levels -- G x G x npad granules per exchange would be tens of MB per iteration chip-wide -- workgroup
   //   w sums the columns [w << lb_slice, ...) (one granule per thread, staged in LDS, a workgroup barrier, added in
   //   workgroup order) and publishes the totals, then every owner reads the ONE total of its item.
   bool aborted = false;
   u32x4 *stage = (u32x4 *)(s_cnt + R * kWideThreads);
   // one granule, polled until it carries this exchange's tag
   auto fetch = [&](const char *p, u32x4 &gr) {
      for (unsigned spins = 0;; ++spins) {
         asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(gr) : "v"(p) : "memory");
         if (granule_ready(gr, round)) break;
         if (spins > kWideSpinLimit || ((spins & 63) == 63 && __hip_atomic_load(g_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            aborted = true;
            break;
         }
         __builtin_amdgcn_s_sleep(1);
      }
   };
   auto exchange = [&](double s, unsigned flag, const int n_items, unsigned &flag_out) -> double {
      flag_out = flag;
      if (G == 1) return s;
      ++round;
      char *buf1 = bufs + (size_t)(round & 1) * G * NPAD * 16; // [G][NPAD] partials
      if (tid < n_items) granule_store(buf1 + ((size_t)w * NPAD + tid) * 16, s, round, flag);
      // the others' granules take ~0.5 us to become visible: a sweep issued right behind the own stores mostly
      // comes back empty and costs a memory round trip
      __builtin_amdgcn_s_sleep(8);
      if (G <= kWideSweep) {
         if (tid < n_items) {
            const char *in = buf1 + (size_t)tid * 16;
            unsigned pending = 0;
#pragma unroll
            for (int k = 0; k < kWideSweep; ++k)
               if (k < G) pending |= 1u << k;
            u32x4 gr[kWideSweep];
            for (unsigned spins = 0;; ++spins) {
#pragma unroll
               for (int k = 0; k < kWideSweep; ++k)
                  if (pending & (1u << k))
                     asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(gr[k]) : "v"(in + (size_t)k * NPAD * 16) : "memory");
               asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
               for (int k = 0; k < kWideSweep; ++k)
                  if ((pending & (1u << k)) && granule_ready(gr[k], round)) pending &= ~(1u << k);
               if (!pending) break;
               // give up: after the spin limit, or when another workgroup has (the error word lives in host memory)
               if (spins > kWideSpinLimit || ((spins & 63) == 63 && __hip_atomic_load(g_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                  aborted = true;
                  break;
               }
               __builtin_amdgcn_s_sleep(1);
            }
            double sum = 0.0;
            unsigned fl = 0;
            if (!aborted) {
#pragma unroll
               for (int k = 0; k < kWideSweep; ++k)
                  if (k < G) {
                     sum += granule_value(gr[k]);
                     fl |= granule_flag(gr[k], round);
                  }
            }
            s = sum;
            flag_out = fl;
         }
      } else {
         char *buf2 = bufs + ((size_t)2 * G * NPAD + (size_t)(round & 1) * NPAD) * 16; // [NPAD] totals
         const int lbm = d.lb_slice, m = 1 << lbm;
         const int col0 = w << lbm;
         const int mw = max(0, min(m, n_items - col0)); // the columns of my slice that exist
         // level 1: thread t = g * m + jj fetches workgroup g's partial of column col0 + jj
         for (int t = tid; t < (G << lbm); t += kWideThreads) {
            const int gg = t >> lbm, jj = t & (m - 1);
            if (jj < mw) {
               u32x4 gr;
               fetch(buf1 + ((size_t)gg * NPAD + col0 + jj) * 16, gr);
               stage[t] = gr;
            }
         }
         if (aborted) s_misc[3] = 1.0;
         __syncthreads();
         if (tid < mw) {
            double sum = 0.0;
            unsigned fl = 0;
            for (int gg = 0; gg < G; ++gg) {
               const u32x4 e = stage[(gg << lbm) + tid];
               sum += granule_value(e);
               fl |= granule_flag(e, round);
            }
            granule_store(buf2 + (size_t)(col0 + tid) * 16, sum, round, fl);
         }
         // level 2: the total of my item
         if (tid < n_items) {
            u32x4 gr;
            fetch(buf2 + (size_t)tid * 16, gr);
            s = granule_value(gr);
            flag_out = granule_flag(gr, round);
         }
      }
      return s;
   };

   // ---- EmSolver::init: total count and "any row kept" over the whole locus (exchange 1, two items)
   {
      const double tw = wave_group_sum<64>(tot);
      const int kw = wave_any(kept != 0) ? 1 : 0;
      if (lane == 0) {
         accw[wave * NPAD + 0] = tw;
         accw[wave * NPAD + 1] = (double)kw;
      }
      if (tid < kWideWaves) s_part[tid] = 0.0;
      if (tid == 0) s_misc[3] = 0.0;
      __syncthreads();
      double s = 0.0;
      if (tid < 2) {
         double p[kWideWaves];
#pragma unroll
         for (int v = 0; v < kWideWaves; ++v) p[v] = accw[v * NPAD + tid];
         s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
      }
      unsigned fo;
      s = exchange(s, 0u, 2, fo);
      if (tid < 2) s_misc[tid] = s;
      if (aborted) s_misc[3] = 1.0;
      __syncthreads();
   }
   if (s_misc[3] != 0.0) {
      if (tid == 0) __hip_atomic_store(g_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
   }
   const double theta0 = s_misc[0] / (double)niso; // :375
   const bool any_kept = s_misc[1] != 0.0;
   if (!any_kept) { // init() == false (:391)
      if (w == 0) {
         if (tid == 0) {
            a.status[locus] = kStInitEmpty;
            a.iters[locus] = 0;
         }
         if (tid < niso) a.theta[iso_base + tid] = theta0;
      }
      return;
   }
   // ---- column sums over the kept rows (exchange 2): scale_j = 1 / sum, a zero column stays zero (:466-478).  The
   // reference normalises F when its first iteration is over; the sums do not depend on theta, so they are made here
   wide_row_lanes_sum<LB_CL>(csum);
   if (b == 0) {
#pragma unroll
      for (int k = 0; k < CPL; ++k) accw[wave * NPAD + k * CL + c] = csum[k];
   }
   __syncthreads();
   // the thread that owns column tid: theta_j (the value BEFORE the iteration's update), phi_j, scale_j
   double th = 0.0, ph_own = 0.0, sc = 0.0;
   {
      double s = 0.0;
      if (tid < NPAD) {
         double p[kWideWaves];
#pragma unroll
         for (int v = 0; v < kWideWaves; ++v) p[v] = accw[v * NPAD + tid];
         s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
      }
      unsigned fo;
      s = exchange(s, 0u, NPAD, fo);
      if (tid < NPAD) {
         sc = (s == 0.0) ? 0.0 : 1.0 / s;
         th = (tid < niso) ? theta0 : 0.0;
         ph_own = th; // the first iteration runs on the raw F (:449-465)
         phi[tid] = ph_own;
      }
      if (aborted) s_misc[3] = 1.0;
   }
   __syncthreads();
   if (s_misc[3] != 0.0) {
      if (tid == 0) __hip_atomic_store(g_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
   }

   // The loop is pipelined by one decision: pass `it` runs iteration `it`'s E- and M-step and update, and -- behind
   // its first barrier -- decides on iteration it - 1, whose shares of ||next - theta||^2 the owner waves reduce at
   // the top of the pass, off the critical path (the old order reduced, synchronised and summed between the update
   // and the next E-step: half of an iteration's cycles were waves parked at barriers and LDS waits).  theta's
   // update is therefore written before anybody knows whether the iteration converged; the owner thread keeps
   // the value of before (th_prev), which is what the reference returns then (estimate.cpp:479-481).  Pass
   // kMaxIter + 1 only decides.
   int32_t st = kStMaxIter;
   int it_done = 0;
   bool theta0_out = false, prev_out = false;
   double th_prev = 0.0, q_prev = 0.0;
   for (int it = 1;; ++it) {
      if (it > 1) {
         const double x = wave_group_sum<64>(q_prev);
         if (lane == 0 && wave < (NPAD + 63) / 64) s_part[wave] = x;
      }
      if (it > kMaxIter) { // the pass that only decides on iteration kMaxIter
         __syncthreads();
         if (s_misc[3] != 0.0) {
            if (tid == 0) __hip_atomic_store(g_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
         }
         double p[kWideWaves];
#pragma unroll
         for (int v = 0; v < kWideWaves; ++v) p[v] = s_part[v];
         const double dsum = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
         it_done = kMaxIter;
         if (s_misc[4 + ((it - 1) & 1)] != 0.0) {
            st = kStDenomZero;
            theta0_out = true;
         } else if (dsum <= kThetaLimitSq) {
            st = kStOk;
            prev_out = true;
         }
         break;
      }
      // ---- E-step and M-step over my tile
      double ph[CPL], acc[CPL];
#pragma unroll
      for (int k = 0; k < CPL; ++k) ph[k] = phi[k * CL + c];
      bool zf = false;
#ifdef SB_WIDE_DIAG
      if (!(SB_WIDE_DIAG & 1))
#endif
#pragma unroll
      for (int rb = 0; rb < R; rb += RBLK) {
         constexpr int NB = RBLK; // rows of this block (R is a multiple of RBLK)
         const int kind = blk_kind[rb / RBLK];
         if (rb > 0 && kind == 0) continue; // no lane of this wave has a kept row in these slots: nothing to add
         double cnt[NB];
#pragma unroll
         for (int q = 0; q < NB; ++q) cnt[q] = s_cnt[(rb + q) * kWideThreads + tid];
         double dd[4];
#pragma unroll
         for (int q = 0; q < NB; ++q) {
            double part = 0.0;
#pragma unroll
            for (int k = 0; k < CPL; ++k) part = __builtin_fma(F[rb + q][k], ph[k], part); // :450
            dd[q] = part;
         }
         wide_col_lanes_sum<LB_CL, NB>(dd);
         double de[4], inv[4];
         double prod = 1.0;
         if (kind == 1) { // every lane's row is a kept one
#pragma unroll
            for (int q = 0; q < NB; ++q) de[q] = dd[q];
         } else {
#pragma unroll
            for (int q = 0; q < NB; ++q) de[q] = keep[rb + q] ? dd[q] : 1.0; // a row outside the problem: n = 0, weight 0 / 1
         }
         if (NB == 4) prod = (de[0] * de[1]) * (de[2] * de[3]);
         else if (NB == 2) prod = de[0] * de[1];
         else prod = de[0];
         // the block's reciprocals from one v_rcp_f64 (em_device.h: batch_reciprocals) unless their product leaves the
         // exponent range in some lane of the wave (tiny denominators of a dying isoform's bins -- or a zero: a
         // product in range has no zero factor, so the test for zero denominators, estimate.cpp:451, lives on the
         // slow side only): then every row gets its own reciprocal
         if (!wave_any(!(prod >= 0x1p-960))) {
            batch_reciprocals<NB>(de, inv);
         } else {
#pragma unroll
            for (int q = 0; q < NB; ++q) {
               zf |= keep[rb + q] && dd[q] == 0.0; // :451
               inv[q] = newton_rcp(de[q]);
            }
         }
#pragma unroll
         for (int q = 0; q < NB; ++q) {
            const double wgt = cnt[q] * inv[q];
            if (rb == 0 && q == 0) { // the first row starts the column partials
#pragma unroll
               for (int k = 0; k < CPL; ++k) acc[k] = wgt * F[0][k];
            } else {
#pragma unroll
               for (int k = 0; k < CPL; ++k) acc[k] = __builtin_fma(wgt, F[rb + q][k], acc[k]);
            }
         }
         // one block's temporaries at a time (the tile leaves few registers): the empty asm pins the column partials
         // here, so the block's updates cannot be sunk behind the later blocks' reciprocals
#pragma unroll
         for (int k = 0; k < CPL; ++k) asm volatile("" : "+v"(acc[k]));
      }
      wide_row_lanes_sum<LB_CL>(acc);
      if (b == 0) {
#pragma unroll
         for (int k = 0; k < CPL; ++k) accw[wave * NPAD + k * CL + c] = acc[k];
      }
      {
         const int zw = wave_any(zf) ? 1 : 0;
         if (lane == 0) s_zf[wave] = zw;
      }
#ifdef SB_WIDE_DIAG
      if (!(SB_WIDE_DIAG & 2))
#endif
      __syncthreads(); // (A) the waves' partial column sums and flags (and the previous iteration's shares) are visible
      if (s_misc[3] != 0.0) {
         if (tid == 0) __hip_atomic_store(g_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         return;
      }
      if (it > 1) { // ---- the decision on iteration it - 1
         double p[kWideWaves];
#pragma unroll
         for (int v = 0; v < kWideWaves; ++v) p[v] = s_part[v];
         const double dsum = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
#ifdef SB_WIDE_DIAG
         if (false)
#endif
         if (s_misc[4 + ((it - 1) & 1)] != 0.0) { // run() == false, _theta untouched (:451-453)
            st = kStDenomZero;
            theta0_out = true;
            it_done = it - 1;
            break;
         }
#ifdef SB_WIDE_DIAG
         if (false)
#endif
         if (dsum <= kThetaLimitSq) { // sqrt(d2) < 1e-2 (:479-480), theta NOT updated: the value of before
            st = kStOk;
            prev_out = true;
            it_done = it - 1;
            break;
         }
      }
      // ---- the owner of column tid: the locus-wide sum, next_theta (:454-464), its share of ||next - theta||^2
#ifdef SB_WIDE_DIAG
      if (!(SB_WIDE_DIAG & 8))
#endif
      {
         double s = 0.0;
         unsigned flag = 0;
         if (tid < NPAD) {
            double p[kWideWaves];
#pragma unroll
            for (int v = 0; v < kWideWaves; ++v) p[v] = accw[v * NPAD + tid];
            s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
#pragma unroll
            for (int v = 0; v < kWideWaves; ++v) flag |= (unsigned)s_zf[v];
         }
         unsigned fo;
         s = exchange(s, flag, NPAD, fo);
         if (tid < NPAD) {
            double nt = 0.0;
            q_prev = 0.0;
            if (tid < niso) {
               nt = ph_own * s;
               const double df = nt - th;
               q_prev = df * df;
            }
            th_prev = th;
            th = nt;          // :481 (undone by prev_out when the iteration turns out to have converged)
            ph_own = nt * sc; // (phi of the padding columns stays 0)
            phi[tid] = ph_own;
            if (tid == 0) s_misc[4 + (it & 1)] = fo ? 1.0 : 0.0;
            if (aborted) s_misc[3] = 1.0;
         }
      }
#ifdef SB_WIDE_DIAG
      if (!(SB_WIDE_DIAG & 4))
#endif
      __syncthreads(); // (B) phi is complete
   }
   if (w == 0) {
      if (tid == 0) {
         a.status[locus] = st;
         a.iters[locus] = it_done;
      }
      if (tid < niso) a.theta[iso_base + tid] = theta0_out ? theta0 : (prev_out ? th_prev : th);
   }
}

// One launch serves loci of all widths (a round's workgroups must all be resident together, and cooperative
// launches do not overlap, so rounds should be few and full): the workgroup looks up its locus and jumps to the
// instantiation its width needs.
#ifdef SB_COMPILE_WIDE_KERNEL
__global__ __launch_bounds__(kWideThreads) void em_wide_kernel(WideArgs g)
{
   int di = 0;
   for (int k = 1; k < g.n_desc; ++k)
      if (g.table[k].first_block <= (int)blockIdx.x) di = k;
   di = __builtin_amdgcn_readfirstlane(di);
   switch (g.table[di].layout) {
#ifdef SB_WIDE_ONLY // diagnostic: one instantiation per build (register report)
#define SB_WIDE_CASE(ID) \
   case ID: if (ID == SB_WIDE_ONLY) em_wide_body<wide_layout(ID).lb_cl, wide_layout(ID).cpl, wide_layout(ID).r, wide_layout(ID).rblk>(g, di); break;
#else
#define SB_WIDE_CASE(ID) \
   case ID: em_wide_body<wide_layout(ID).lb_cl, wide_layout(ID).cpl, wide_layout(ID).r, wide_layout(ID).rblk>(g, di); break;
#endif
      SB_WIDE_CASE(0)
      SB_WIDE_CASE(1)
      SB_WIDE_CASE(2)
      SB_WIDE_CASE(3)
      SB_WIDE_CASE(4)
      SB_WIDE_CASE(5)
      SB_WIDE_CASE(6)
#undef SB_WIDE_CASE
   }
}
#endif

hipError_t launch_wide(const WideArgs &g, int n_blocks, size_t lds_bytes, hipStream_t s);

} // namespace sb
