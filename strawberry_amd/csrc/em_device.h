// strawberry_amd/csrc/em_device.h
//
// Device code of the per-locus Latent-Class-Model EM for gfx950 (MI355X, wave64).
// Replaces EmSolver::init / EmSolver::run of the reference
// (/root/reference/src/estimate.cpp:366-409 and :411-488); see DESIGN.md for the
// mapping onto the hardware.
//
// One EM iteration for a locus with bin counts n_i, weights F_ij, abundances th_j:
//     d_i   = sum_j F_ij th_j                     (E-step denominator, estimate.cpp:450)
//     w_i   = n_i / d_i
//     th'_j = th_j * sum_i w_i F_ij               (= sum_i U_ij, estimate.cpp:454-464)
//     stop when ||th' - th||_2 < 1e-2, returning th  (estimate.cpp:479-486)
// The first iteration runs on the raw F; afterwards F is the column-normalised
// F (estimate.cpp:466-478).  The reference renormalises every iteration, which is
// idempotent up to rounding, so it is done once here.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sb {

constexpr int kMaxIter = 1000;          // include/estimate.hpp:237
constexpr double kThetaLimit = 1e-2;    // include/estimate.hpp:241
constexpr double kRowEps = 1e-5;        // src/estimate.cpp:380

constexpr int32_t kStOk = 0, kStInitEmpty = 1, kStDenomZero = 2, kStMaxIter = 3;

// Device view of a batch (CSR-of-loci, include/sbgpu.h) and its outputs.
struct EmArgs {
   const int64_t *row_off;
   const int64_t *iso_off;
   const int64_t *f_off;
   const int32_t *count;
   const double *F;
   double *theta;
   int32_t *status;
   int32_t *iters;
};

// One size class: the loci it holds (ordered by decreasing work) and the
// dynamic-pull cursor.
struct ClassArgs {
   const int32_t *loci;
   int32_t n;
   int32_t *cursor;
};

// ------------------------------------------------------------------ cross-lane
// 64-bit values move as two 32-bit DPP movs (v_add_f64 is VOP3: no DPP operand).
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x)
{
   int lo = __double2loint(x), hi = __double2hiint(x);
   lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
   hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
   return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_mov_i(int x)
{
   return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true);
}

constexpr int kDppXor1 = 0xB1;        // quad_perm:[1,0,3,2]
constexpr int kDppXor2 = 0x4E;        // quad_perm:[2,3,0,1]
constexpr int kDppHalfMirror = 0x141; // lane i <- 7-i   (other quad of the 8-lane group)
constexpr int kDppMirror = 0x140;     // lane i <- 15-i  (other half of the 16-lane row)

__device__ __forceinline__ double swizzle_xor16(double x)
{
   int lo = __double2loint(x), hi = __double2hiint(x);
   lo = __builtin_amdgcn_ds_swizzle(lo, 0x401F); // bitmask mode: xor 0x10, and 0x1F
   hi = __builtin_amdgcn_ds_swizzle(hi, 0x401F);
   return __hiloint2double(hi, lo);
}

// x(lane) + x(lane^32) with v_permlane32_swap (gfx950): after the swap one
// register holds the low half twice and the other the high half twice.
__device__ __forceinline__ double sum_xor32(double x)
{
   int lo = __double2loint(x), hi = __double2hiint(x);
   auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
   auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
   return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}

// All-reduce (sum) over the GW consecutive lanes of a wave that form one group
// (GW = 1,2,4,...,64).  Every step adds a value to its butterfly partner's, so all
// lanes of a group end with bitwise identical sums.
template <int GW>
__device__ __forceinline__ double wave_group_sum(double x)
{
   if (GW >= 2) x += dpp_mov<kDppXor1>(x);
   if (GW >= 4) x += dpp_mov<kDppXor2>(x);
   if (GW >= 8) x += dpp_mov<kDppHalfMirror>(x);
   if (GW >= 16) x += dpp_mov<kDppMirror>(x);
   if (GW >= 32) x += swizzle_xor16(x);
   if (GW >= 64) x = sum_xor32(x);
   return x;
}

template <int GW>
__device__ __forceinline__ int wave_group_or(int x)
{
   if (GW >= 2) x |= dpp_mov_i<kDppXor1>(x);
   if (GW >= 4) x |= dpp_mov_i<kDppXor2>(x);
   if (GW >= 8) x |= dpp_mov_i<kDppHalfMirror>(x);
   if (GW >= 16) x |= dpp_mov_i<kDppMirror>(x);
   if (GW >= 32) x |= __builtin_amdgcn_ds_swizzle(x, 0x401F);
   if (GW >= 64) {
      auto a = __builtin_amdgcn_permlane32_swap(x, x, false, false);
      x = a[0] | a[1];
   }
   return x;
}

// ------------------------------------------------------------------ arithmetic
// n / d to <= 1 ulp without the IEEE div_scale/div_fixup tail: v_rcp_f64 seed,
// two Newton steps, one residual correction.  Operands here are O(1e-6 .. 1e9);
// d == 0 gives NaN, which callers mask or turn into DENOM_ZERO before use.
__device__ __forceinline__ double fast_div(double n, double d)
{
   double r = __builtin_amdgcn_rcp(d);
   double e = __builtin_fma(-d, r, 1.0);
   r = __builtin_fma(r, e, r);
   e = __builtin_fma(-d, r, 1.0);
   r = __builtin_fma(r, e, r);
   double q = n * r;
   double rem = __builtin_fma(-d, q, n);
   return __builtin_fma(rem, r, q);
}

// ================================================================== tile kernel
// A locus is owned by a GROUP of G lanes; lane g of the group keeps rows
// g, g+G, g+2G, ... (R of them) of the locus' F in registers as an R x C tile,
// plus a private copy of theta[C].  G <= 64: 64/G groups share a wave and run
// independent loci; G > 64: the group is the whole workgroup (G threads) and the
// cross-wave part of every reduction goes through LDS.
//
// Groups pull loci from the class list through an atomic cursor and keep pulling
// until it runs dry, so a wave stays busy while one of its groups is still
// iterating (iteration counts range from 1 to the 1000 cap).
template <int G>
struct GroupComm {
   static constexpr int GW = (G < 64) ? G : 64;   // lanes of the group inside one wave
   static constexpr int NW = (G < 64) ? 1 : G / 64; // waves per group
   double *lds;                                   // NW > 1: [2][NW] doubles per value slot

   // all-reduce of V values at once (cross-wave: one LDS round for all of them)
   template <int V>
   __device__ __forceinline__ void sum(double (&x)[V], int wave_id, int &phase)
   {
#pragma unroll
      for (int v = 0; v < V; ++v) x[v] = wave_group_sum<GW>(x[v]);
      if (NW > 1) {
         // double-buffered by phase so one barrier per round is enough
         double *buf = lds + (size_t)phase * (V_MAX * NW);
         if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int v = 0; v < V; ++v) buf[v * NW + wave_id] = x[v];
         }
         __syncthreads();
#pragma unroll
         for (int v = 0; v < V; ++v) {
            double s = 0.0;
            for (int w = 0; w < NW; ++w) s += buf[v * NW + w];
            x[v] = s;
         }
         phase ^= 1;
      }
   }
   static constexpr int V_MAX = 40; // >= C + 3 for the largest C (32)
};

template <int C, int R, int G>
__global__ __launch_bounds__((G < 64) ? 64 : G) void em_tile_kernel(EmArgs a, ClassArgs cls)
{
   constexpr int GW = GroupComm<G>::GW;
   constexpr int NW = GroupComm<G>::NW;
   __shared__ double s_red[(NW > 1) ? 2 * GroupComm<G>::V_MAX * NW : 1];
   __shared__ int s_idx;
   GroupComm<G> comm;
   comm.lds = s_red;
   int phase = 0;

   const int lane = threadIdx.x & 63;
   const int wave_id = threadIdx.x >> 6;
   const int g = (G < 64) ? (lane & (G - 1)) : (int)threadIdx.x; // index inside the group

   double F[R][C];
   double nn[R];      // n_i as double (obs_d, estimate.cpp:418-419)
   bool act[R];       // row kept by init() (estimate.cpp:377-390) and inside the locus
   double theta[C];
   double theta0 = 0.0;
   int it = 0;
   int niso = 0;
   int locus = -1;
   int64_t iso_base = 0;
   bool have = false;       // the group currently owns a locus
   bool exhausted = false;  // the class list ran dry for this group

#pragma unroll
   for (int r = 0; r < R; ++r) {
      nn[r] = 0.0;
      act[r] = false;
#pragma unroll
      for (int j = 0; j < C; ++j) F[r][j] = 0.0;
   }
#pragma unroll
   for (int j = 0; j < C; ++j) theta[j] = 0.0;

   for (;;) {
      // ---------------------------------------------------------------- refill
      if (!have && !exhausted) {
         int idx;
         if (NW > 1) {
            if (threadIdx.x == 0) s_idx = atomicAdd(cls.cursor, 1);
            __syncthreads();
            idx = s_idx;
            __syncthreads();
         } else {
            idx = 0;
            if (g == 0) idx = atomicAdd(cls.cursor, 1);
            idx = __shfl(idx, lane & ~(GW - 1));
         }
         if (idx >= cls.n) {
            exhausted = true;
         } else {
            locus = cls.loci[idx];
            const int64_t r0 = a.row_off[locus];
            const int nrow = (int)(a.row_off[locus + 1] - r0);
            iso_base = a.iso_off[locus];
            niso = (int)(a.iso_off[locus + 1] - iso_base);
            const double *Fg = a.F + a.f_off[locus];
            // EmSolver::init, estimate.cpp:366-391
            double red[2];
            red[0] = 0.0; // sum of ALL counts (theta0 precedes the row drop, :374-375)
            red[1] = 0.0; // number of kept rows
#pragma unroll
            for (int r = 0; r < R; ++r) {
               const int i = r * G + g;
               const bool valid = i < nrow;
               nn[r] = valid ? (double)a.count[r0 + i] : 0.0;
               red[0] += nn[r];
               bool keep = false;
#pragma unroll
               for (int j = 0; j < C; ++j) {
                  double v = (valid && j < niso) ? Fg[(int64_t)i * niso + j] : 0.0;
                  keep = keep || (v > kRowEps); // :380
                  F[r][j] = v;
               }
               act[r] = keep;
               if (!keep) {
                  nn[r] = 0.0;
#pragma unroll
                  for (int j = 0; j < C; ++j) F[r][j] = 0.0;
               } else {
                  red[1] += 1.0;
               }
            }
            comm.template sum<2>(red, wave_id, phase);
            theta0 = red[0] / (double)niso; // :375, IEEE division
#pragma unroll
            for (int j = 0; j < C; ++j) theta[j] = (j < niso) ? theta0 : 0.0;
            it = 0;
            if (red[1] == 0.0) {
               // init() == false (:391): theta = theta0, the caller drops the locus
               if (g == 0) {
                  a.status[locus] = kStInitEmpty;
                  a.iters[locus] = 0;
               }
#pragma unroll
               for (int j = 0; j < C; ++j)
                  if (j < niso && (j % G) == g) a.theta[iso_base + j] = theta0;
            } else {
               have = true;
            }
         }
      }
      if (NW > 1) {
         // have / exhausted are workgroup-uniform
         if (!have) {
            if (exhausted) break;
            continue;
         }
      } else {
         if (!__any(have)) {
            if (__all(exhausted)) break;
            continue;
         }
      }

      // ------------------------------------------------- one EM iteration (uniform)
      // E-step denominators and the weighted column sums of the M-step.
      double red[C + 1];
#pragma unroll
      for (int j = 0; j < C + 1; ++j) red[j] = 0.0;
      int zero_flag = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
         double d = 0.0;
#pragma unroll
         for (int j = 0; j < C; ++j) d = __builtin_fma(F[r][j], theta[j], d); // :450
         zero_flag |= (act[r] && d == 0.0) ? 1 : 0;                          // :451
         double w = fast_div(nn[r], d);
         w = act[r] ? w : 0.0;
#pragma unroll
         for (int j = 0; j < C; ++j) red[j] = __builtin_fma(w, F[r][j], red[j]);
      }
      red[C] = (double)zero_flag;
      comm.template sum<C + 1>(red, wave_id, phase);
      const bool dz = red[C] != 0.0;

      double next_theta[C];
      double d2 = 0.0;
#pragma unroll
      for (int j = 0; j < C; ++j) {
         next_theta[j] = theta[j] * red[j]; // = sum_i U_ij, :454-464
         double df = next_theta[j] - theta[j];
         d2 = __builtin_fma(df, df, d2);    // :479
      }

      // ------------------------------------------------------- per-group epilogue
      if (have) {
         bool finished = false;
         int32_t st = kStOk;
         double out_scalar = 0.0;
         bool out_is_theta0 = false;
         if (dz) {
            // run() returns false before touching _theta (:451-453): theta0 survives
            finished = true;
            st = kStDenomZero;
            out_is_theta0 = true;
            out_scalar = theta0;
         } else {
            if (it == 0) {
               // F <- column-normalised F (:466-478); a zero column stays zero
               double cs[C];
#pragma unroll
               for (int j = 0; j < C; ++j) {
                  double s = 0.0;
#pragma unroll
                  for (int r = 0; r < R; ++r) s += F[r][j];
                  cs[j] = s;
               }
               comm.template sum<C>(cs, wave_id, phase);
#pragma unroll
               for (int j = 0; j < C; ++j) {
                  const double inv = (cs[j] == 0.0) ? 0.0 : 1.0 / cs[j];
#pragma unroll
                  for (int r = 0; r < R; ++r) F[r][j] *= inv;
               }
            }
            if (sqrt(d2) < kThetaLimit) {
               finished = true; // break before theta = next_theta (:480)
               st = kStOk;
            } else {
#pragma unroll
               for (int j = 0; j < C; ++j) theta[j] = next_theta[j]; // :481
               if (it + 1 == kMaxIter) {
                  finished = true;
                  st = kStMaxIter;
               }
            }
         }
         ++it;
         if (finished) {
            if (g == 0) {
               a.status[locus] = st;
               a.iters[locus] = it;
            }
#pragma unroll
            for (int j = 0; j < C; ++j) {
               if (j < niso && (j % G) == g) a.theta[iso_base + j] = out_is_theta0 ? out_scalar : theta[j];
            }
            have = false;
         }
      }
   }
}

// ============================================================= streaming kernel
// Any shape (niso <= 64*kStreamSlots): one 256-thread workgroup per locus, F is
// re-read from memory (L2 / Infinity Cache) every iteration and never modified;
// the column normalisation is carried as a per-column scale s_j folded into
// phi_j = s_j * theta_j.  LW lanes cooperate on one row (coalesced row reads),
// 64/LW rows per wave-step.
constexpr int kStreamThreads = 256;
constexpr int kStreamSlots = 8; // column slots per lane: niso <= LW * 8 <= 512

__device__ __forceinline__ double wave_group_sum_rt(double x, int gw)
{
   // runtime group width (wave-uniform branches)
   if (gw >= 2) x += dpp_mov<kDppXor1>(x);
   if (gw >= 4) x += dpp_mov<kDppXor2>(x);
   if (gw >= 8) x += dpp_mov<kDppHalfMirror>(x);
   if (gw >= 16) x += dpp_mov<kDppMirror>(x);
   if (gw >= 32) x += swizzle_xor16(x);
   if (gw >= 64) x = sum_xor32(x);
   return x;
}
// sum over lanes that share (lane % lw): the butterfly steps ABOVE lw
__device__ __forceinline__ double wave_stride_sum_rt(double x, int lw)
{
   if (lw <= 1) x += __shfl_xor(x, 1);
   if (lw <= 2) x += __shfl_xor(x, 2);
   if (lw <= 4) x += __shfl_xor(x, 4);
   if (lw <= 8) x += __shfl_xor(x, 8);
   if (lw <= 16) x += __shfl_xor(x, 16);
   if (lw <= 32) x += __shfl_xor(x, 32);
   return x;
}

__global__ __launch_bounds__(kStreamThreads) void em_stream_kernel(EmArgs a, ClassArgs cls,
                                                                   uint8_t *row_keep /*[total rows]*/)
{
   // phi[npad] | theta[npad] | scale[npad] | accw[NWAVE][npad]
   extern __shared__ double s_dyn[];
   __shared__ double s_part[kStreamThreads / 64];
   __shared__ int s_flag;
   __shared__ int s_idx;
   const int tid = threadIdx.x;
   const int lane = tid & 63;
   const int wave = tid >> 6;
   constexpr int NWAVE = kStreamThreads / 64;

   for (;;) {
      if (tid == 0) {
         s_idx = atomicAdd(cls.cursor, 1);
         s_flag = 0;
      }
      __syncthreads();
      const int idx = s_idx;
      if (idx >= cls.n) break;
      const int locus = cls.loci[idx];
      const int64_t r0 = a.row_off[locus];
      const int nrow = (int)(a.row_off[locus + 1] - r0);
      const int64_t iso_base = a.iso_off[locus];
      const int niso = (int)(a.iso_off[locus + 1] - iso_base);
      const double *Fg = a.F + a.f_off[locus];
      int lw = 1;                           // lanes cooperating on one row
      while (lw < niso && lw < 64) lw <<= 1;
      const int rps = 64 / lw;              // rows per wave-step
      const int nslot = (niso + lw - 1) / lw;
      const int npad = lw * nslot;
      double *phi = s_dyn;
      double *theta = s_dyn + npad;
      double *scale = s_dyn + 2 * npad;
      double *accw = s_dyn + 3 * npad;
      const int c0 = lane & (lw - 1);
      const int rsub = lane / lw;

      // ---- EmSolver::init (estimate.cpp:366-391): total count, row keep flags
      double tot = 0.0;
      for (int i = tid; i < nrow; i += kStreamThreads) tot += (double)a.count[r0 + i];
      tot = wave_group_sum<64>(tot);
      if (lane == 0) s_part[wave] = tot;
      int kept = 0;
      for (int base = wave * rps; base < nrow; base += NWAVE * rps) {
         const int i = base + rsub;
         double mx = 0.0;
         if (i < nrow) {
            for (int k = 0; k < nslot; ++k) {
               const int j = c0 + k * lw;
               if (j < niso) mx = fmax(mx, Fg[(int64_t)i * niso + j]);
            }
         }
         for (int m = 1; m < lw; m <<= 1) mx = fmax(mx, __shfl_xor(mx, m));
         const bool keep = (i < nrow) && (mx > kRowEps); // :380
         if (i < nrow && c0 == 0) row_keep[r0 + i] = keep ? 1 : 0;
         kept |= keep ? 1 : 0;
      }
      if (kept) atomicOr(&s_flag, 1);
      __syncthreads();
      tot = 0.0;
      for (int w = 0; w < NWAVE; ++w) tot += s_part[w];
      const double theta0 = tot / (double)niso; // :375
      const bool any_kept = s_flag != 0;
      for (int j = tid; j < npad; j += kStreamThreads) {
         const double t = (j < niso) ? theta0 : 0.0;
         theta[j] = t;
         scale[j] = 1.0;
         phi[j] = t;
      }
      __syncthreads();
      if (tid == 0) s_flag = 0;
      if (!any_kept) {
         // init() == false (:391)
         if (tid == 0) {
            a.status[locus] = kStInitEmpty;
            a.iters[locus] = 0;
         }
         for (int j = tid; j < niso; j += kStreamThreads) a.theta[iso_base + j] = theta0;
         __syncthreads();
         continue;
      }
      __syncthreads();

      int32_t st = kStMaxIter;
      int it = 0;
      bool theta0_out = false;
      while (it < kMaxIter) {
         double acc[kStreamSlots];
#pragma unroll
         for (int k = 0; k < kStreamSlots; ++k) acc[k] = 0.0;
         int zf = 0;
         for (int base = wave * rps; base < nrow; base += NWAVE * rps) {
            const int i = base + rsub;
            const bool valid = i < nrow;
            double fv[kStreamSlots];
            double part = 0.0;
#pragma unroll
            for (int k = 0; k < kStreamSlots; ++k) {
               fv[k] = 0.0;
               if (k < nslot) {
                  const int j = c0 + k * lw;
                  if (valid && j < niso) fv[k] = Fg[(int64_t)i * niso + j];
                  part = __builtin_fma(fv[k], phi[j], part); // :450 with F' = F*scale
               }
            }
            const double d = wave_group_sum_rt(part, lw);
            const bool keep = valid && row_keep[r0 + i] != 0;
            zf |= (keep && d == 0.0) ? 1 : 0;                 // :451
            double w = fast_div(valid ? (double)a.count[r0 + i] : 0.0, d);
            w = keep ? w : 0.0;
#pragma unroll
            for (int k = 0; k < kStreamSlots; ++k) acc[k] = __builtin_fma(w, fv[k], acc[k]);
         }
         // column partials: over the row sub-groups of the wave, then one slot per wave
#pragma unroll
         for (int k = 0; k < kStreamSlots; ++k) {
            if (k < nslot) {
               const double s = wave_stride_sum_rt(acc[k], lw);
               if (rsub == 0) accw[wave * npad + c0 + k * lw] = s;
            }
         }
         if (zf) atomicOr(&s_flag, 1);
         __syncthreads(); // (A) accw and s_flag complete
         const bool dz = s_flag != 0;
         double d2 = 0.0;
         for (int j = tid; j < niso; j += kStreamThreads) {
            double s = 0.0;
            for (int w = 0; w < NWAVE; ++w) s += accw[w * npad + j];
            const double nt = phi[j] * s; // theta_j * scale_j * sum_i w_i F_ij  (:454-464)
            const double df = nt - theta[j];
            d2 = __builtin_fma(df, df, d2);
            accw[j] = nt; // column j is owned by this thread: wave-0 slot now holds next_theta
         }
         d2 = wave_group_sum<64>(d2);
         if (lane == 0) s_part[wave] = d2;
         __syncthreads(); // (B)
         if (tid == 0) s_flag = 0; // every thread has read dz; next atomicOr is behind more barriers
         d2 = 0.0;
         for (int w = 0; w < NWAVE; ++w) d2 += s_part[w];
         ++it;
         if (dz) {
            // run() == false, _theta untouched (:451-453)
            st = kStDenomZero;
            theta0_out = true;
            break;
         }
         if (it == 1) {
            // scale_j = 1 / (column sum over kept rows), zero column stays zero (:466-478)
            double cs[kStreamSlots];
#pragma unroll
            for (int k = 0; k < kStreamSlots; ++k) cs[k] = 0.0;
            for (int base = wave * rps; base < nrow; base += NWAVE * rps) {
               const int i = base + rsub;
               const bool keep = (i < nrow) && row_keep[r0 + i] != 0;
#pragma unroll
               for (int k = 0; k < kStreamSlots; ++k) {
                  const int j = c0 + k * lw;
                  if (keep && k < nslot && j < niso) cs[k] += Fg[(int64_t)i * niso + j];
               }
            }
#pragma unroll
            for (int k = 0; k < kStreamSlots; ++k) {
               if (k < nslot) {
                  const double s = wave_stride_sum_rt(cs[k], lw);
                  // wave 0's accw slot holds next_theta: its partial goes to scale[]
                  if (rsub == 0) {
                     if (wave == 0) scale[c0 + k * lw] = s;
                     else accw[wave * npad + c0 + k * lw] = s;
                  }
               }
            }
            __syncthreads();
            for (int j = tid; j < niso; j += kStreamThreads) {
               double s = scale[j];
               for (int w = 1; w < NWAVE; ++w) s += accw[w * npad + j];
               scale[j] = (s == 0.0) ? 0.0 : 1.0 / s;
            }
            // the barrier below orders scale[] before phi is rebuilt
         }
         if (sqrt(d2) < kThetaLimit) { // :479-480, theta NOT updated
            st = kStOk;
            break;
         }
         for (int j = tid; j < niso; j += kStreamThreads) theta[j] = accw[j]; // :481
         __syncthreads();
         for (int j = tid; j < npad; j += kStreamThreads) phi[j] = (j < niso) ? theta[j] * scale[j] : 0.0;
         __syncthreads();
      }
      if (tid == 0) {
         a.status[locus] = st;
         a.iters[locus] = it;
      }
      for (int j = tid; j < niso; j += kStreamThreads) a.theta[iso_base + j] = theta0_out ? theta0 : theta[j];
      __syncthreads();
   }
}

} // namespace sb
