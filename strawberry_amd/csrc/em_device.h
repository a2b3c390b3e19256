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

#include <type_traits>

#include "device_common.h"

namespace sb {

constexpr int kMaxIter = 1000;          // include/estimate.hpp:237
constexpr double kThetaLimit = 1e-2;    // include/estimate.hpp:241
constexpr double kThetaLimitSq = 0x1.a36e2eb1c432bp-14; // 9.999999999999998e-05: sqrt(x) < 1e-2  <=>  x <= this
constexpr double kRowEps = 1e-5;        // src/estimate.cpp:380

constexpr int kMaxCPLv = 8;            // columns per lane of the widest register tile
constexpr int32_t kStOk = 0, kStInitEmpty = 1, kStDenomZero = 2, kStMaxIter = 3;
constexpr int32_t kStRunning = 4; // transient: suspended at a phase limit, resumed by the next phase

// Device view of a batch (CSR-of-loci, include/sbgpu.h) and its outputs.
// T = double: the reference's arithmetic (every parity claim is about this instantiation).  T = float: the fp32
// variant of BASELINE config 5 (F, theta and all arithmetic in fp32), for the tolerance sweep -- not a parity target.
template <class T>
struct EmArgsT {
   const int64_t *row_off;
   const int64_t *iso_off;
   const int64_t *f_off;
   const int32_t *count;
   const T *F;
   T *theta;
   int32_t *status;
   int32_t *iters;
   // BASELINE config 5 ("bias kernel fused into the E-step"): null, or a factor per bin and per isoform -- the tile
   // kernels then take F_ij * 2^(row_bias[i] * iso_bias[j]) as the weight, applied to the operand as the tile is
   // loaded (once per solve; the iterations run on the biased tile).  Both in [-1, 1]: b_ij in [0.5, 2].
   const T *row_bias = nullptr; // [total rows]
   const T *iso_bias = nullptr; // [total isoforms]
};
typedef EmArgsT<double> EmArgs;

// Descriptor of a size class inside a launch's table (sorted by first block).
struct ClassDesc {
   int32_t block_begin; // first blockIdx.x of this class (later phases: written on the device, phase_prepare_kernel)
   int32_t n;           // loci in the class (later phases: its capacity -- every locus that could reach it)
   int32_t loci_off;    // offset of its list in the concatenated class lists
   int32_t shape;       // layout | rmult << 8 | lbG << 16
};

// One size class as a workgroup sees it: the loci it holds (likeliest stragglers first) and the batch of them
// this workgroup serves.
struct ClassArgs {
   const int32_t *loci;
   int32_t n;
   // phased execution: a locus still running after `it_limit` iterations is suspended (theta and the
   // iteration count are its whole state) and appended to the list of ITS class of the next phase
   // (`route[locus]`, an index into `next_table`): the next phase gives the survivors layouts with more
   // lanes per locus -- fewer instructions per iteration -- now that fewer loci are alive
   int32_t *out;                 // the next phase's class lists (one array, next_table[c].loci_off)
   int32_t *out_count;           // [class of the next phase]
   const int32_t *route;         // [locus] -> class of the next phase
   const ClassDesc *next_table;
   int32_t it_limit;
   int32_t resume; // this phase's loci carry state from the previous one
   int32_t batch;  // the workgroup serves exactly this batch of the class: loci [batch * k, batch * k + k), k = loci per workgroup
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
__device__ __forceinline__ float dpp_mov(float x)
{
   return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
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

// x(lane) + x(lane^16) with v_permlane16_swap (gfx950): the swap leaves the even rows (16 lanes each) twice in
// one register and the odd rows twice in the other.  13 cycles per swap against 56 for a ds_swizzle round trip
// (tools/microbench.hip).
__device__ __forceinline__ double sum_xor16(double x)
{
   int lo = __double2loint(x), hi = __double2hiint(x);
   auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
   auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
   return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}

__device__ __forceinline__ float sum_xor16(float x)
{
   auto a = __builtin_amdgcn_permlane16_swap(__float_as_int(x), __float_as_int(x), false, false);
   return __int_as_float(a[0]) + __int_as_float(a[1]);
}
__device__ __forceinline__ float sum_xor32(float x)
{
   auto a = __builtin_amdgcn_permlane32_swap(__float_as_int(x), __float_as_int(x), false, false);
   return __int_as_float(a[0]) + __int_as_float(a[1]);
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
__device__ __forceinline__ double sum_bits45(double x); // below: lane bits 4 and 5 through the matrix pipe
template <int GW>
__device__ __forceinline__ double wave_group_sum(double x)
{
   if (GW >= 2) x += dpp_mov<kDppXor1>(x);
   if (GW >= 4) x += dpp_mov<kDppXor2>(x);
   if (GW >= 8) x += dpp_mov<kDppHalfMirror>(x);
   if (GW >= 16) x += dpp_mov<kDppMirror>(x);
   if (GW >= 64) return sum_bits45(x);
   if (GW >= 32) x = sum_xor16(x);
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
// n / d without the IEEE div_scale / div_fixup tail: v_rcp_f64 seed (~2^-24 relative)
// and two Newton steps (-> ~2^-52), then one multiply: <= 2 ulp.  Operands here are
// O(1e-6 .. 1e9); d == 0 gives NaN, which callers mask or turn into DENOM_ZERO.
__device__ __forceinline__ double fast_div(double n, double d)
{
   double r = __builtin_amdgcn_rcp(d);
   double e = __builtin_fma(-d, r, 1.0);
   r = __builtin_fma(r, e, r);
   e = __builtin_fma(-d, r, 1.0);
   r = __builtin_fma(r, e, r);
   return n * r;
}

// Reciprocals of N = 1..4 denominators from ONE v_rcp_f64 (the instruction occupies the pipe four times as long as an
// FMA, and each of its results wants two Newton steps): invert the product and multiply the other factors back in
// (a * b inverted gives 1/a = r * b, 1/b = r * a; four go as a tree of two pairs).  <= 5 roundings per result instead
// of 3.  A zero among the denominators makes ALL results NaN (0 * inf) -- callers only ever ask whether any
// denominator was zero.  The product of N numbers can leave the exponent range where no single one does: an underflow
// (flushed to zero) also gives NaN -- callers confirm a NaN with exact_reciprocal-based arithmetic before they
// believe it; denominators here are <= ~1e10, so no overflow.
__device__ __forceinline__ double newton_rcp(double p)
{
   double r = __builtin_amdgcn_rcp(p);
   double e = __builtin_fma(-p, r, 1.0);
   r = __builtin_fma(r, e, r);
   e = __builtin_fma(-p, r, 1.0);
   return __builtin_fma(r, e, r);
}
template <int N>
__device__ __forceinline__ void batch_reciprocals(const double (&d)[4], double (&inv)[4])
{
   if (N == 1) {
      inv[0] = newton_rcp(d[0]);
   } else if (N == 2) {
      const double r = newton_rcp(d[0] * d[1]);
      inv[0] = r * d[1];
      inv[1] = r * d[0];
   } else if (N == 3) {
      const double a = d[0] * d[1];
      const double r = newton_rcp(a * d[2]);
      const double ra = r * d[2]; // 1 / (d0 d1)
      inv[0] = ra * d[1];
      inv[1] = ra * d[0];
      inv[2] = r * a;
   } else {
      const double a = d[0] * d[1], b = d[2] * d[3];
      const double r = newton_rcp(a * b);
      const double ra = r * b, rb = r * a;
      inv[0] = ra * d[1];
      inv[1] = ra * d[0];
      inv[2] = rb * d[3];
      inv[3] = rb * d[2];
   }
}
// fp32: v_rcp_f32 is good to 1 ulp; one Newton step on the quotient keeps n / d within ~1 ulp
__device__ __forceinline__ float fast_div(float n, float d)
{
   const float r = __builtin_amdgcn_rcpf(d);
   const float q = n * r;
   return __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
}
__device__ __forceinline__ double bias_factor(double x) { return exp2(x); }
__device__ __forceinline__ float bias_factor(float x) { return exp2f(x); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// squared convergence threshold: largest value whose correctly rounded square root is below 1e-2
template <class T> struct ThetaLimitSq;
template <> struct ThetaLimitSq<double> { static constexpr double value = 0x1.a36e2eb1c432bp-14; };
template <> struct ThetaLimitSq<float> { static constexpr float value = 0x1.a36e2ap-14f; };

// ================================================================== tile kernel
// A locus is owned by a GROUP of G = CL x GR lanes laid out as a 2-D grid:
//   gc  "column lane": owns columns [gc*CPL, gc*CPL + CPL)
//   gr  "row lane":    owns rows gr, gr+GR, gr+2GR, ... (R of them)
// so each lane keeps an R x CPL tile of F in registers for the whole solve, plus
// theta for its own CPL columns.  Per iteration the group needs
//   - the row denominators: all-reduce over the CL column lanes,
//   - the weighted column sums: all-reduce over the GR row lanes (in the workgroup
//     form it continues through LDS across the waves);
// which lane bits carry gc and gr -- and hence what these reductions cost -- is the "matrix lane map" below
// (row_rank).  Every lane of a group ends with bitwise identical sums, so the convergence decision is
// group-uniform by construction.
//
// Wave form  (BLOCK = false): G = 2^lbG <= 64 lanes, 64/G groups share a wave and
//                             run independent loci; G is a run-time (wave-uniform)
//                             value so that one kernel serves every size class.
// Block form (BLOCK = true):  the group is the whole workgroup (256 lanes).
// A workgroup serves exactly one batch of its class (64/G loci; one locus in the block form).

template <int MASK>
__device__ __forceinline__ double xor_get(double x)
{
   return __hiloint2double(xor_get_i<MASK>(__double2hiint(x)), xor_get_i<MASK>(__double2loint(x)));
}
template <int MASK>
__device__ __forceinline__ float xor_get(float x)
{
   return __int_as_float(xor_get_i<MASK>(__float_as_int(x)));
}
// Sum over lane bits 4 and 5 -- x(l) + x(l^16) + x(l^32) + x(l^48), in every lane -- with ONE matrix
// instruction: v_mfma_f64_4x4x4_4b with A = 1 and B = x gives D(lane) = sum over k of x(16 k + lane % 16)
// (layout probed by tools/microbench.hip).  19-22 cycles on the matrix pipe (which the vector instructions of the
// other waves do not wait for) against ~50 for the two lane-swap steps; every lane of a column adds the same four
// values in the same order, so the sums are bitwise identical across the lanes, as the butterflies' are.
__device__ __forceinline__ double sum_bits45(double x)
{
   return __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, x, 0.0, 0, 0, 0);
}
// fp32: the two lane-swap steps (the fp32-input matrix instructions have no 4-deep form to sum with)
__device__ __forceinline__ float sum_bits45(float x) { return sum_xor32(sum_xor16(x)); }
// value of lane (l - 4) mod 16 / (l - 8) mod 16 of the own 16-lane row
template <class T>
__device__ __forceinline__ T row_ror4(T x) { return dpp_mov<0x124>(x); }
template <class T>
__device__ __forceinline__ T row_ror8(T x) { return dpp_mov<0x128>(x); }

// x(lane) + x(lane ^ MASK).  LOW_UNIFORM: every lane below bit log2(MASK) of the
// group already holds the same value, so the cheaper mirror forms are valid.
template <int MASK, bool LOW_UNIFORM, class T>
__device__ __forceinline__ T xor_sum(T x)
{
   if (MASK == 32) return sum_xor32(x);
   if (MASK == 16) return sum_xor16(x);
   if (LOW_UNIFORM && MASK == 4) return x + dpp_mov<kDppHalfMirror>(x);
   if (LOW_UNIFORM && MASK == 8) return x + dpp_mov<kDppMirror>(x);
   return x + xor_get<MASK>(x);
}

constexpr int ilog2(int x) { return x <= 1 ? 0 : 1 + ilog2(x / 2); }

// rows per row lane of a register tile: rh halves of the base height, capped so that the
// tile stays within the 256 VGPRs a VALU instruction can address (128 fp64 elements)
constexpr int kMaxTileElems = 128;
constexpr int tile_rows(int cpl, int rhalf, int rh)
{
   return (rhalf * rh * cpl > kMaxTileElems) ? kMaxTileElems / cpl : rhalf * rh;
}

// all-reduce over lane bits [0, HI) (compile-time)
template <int HI, class T>
__device__ __forceinline__ T low_bits_sum(T x)
{
   if (HI > 0) x = xor_sum<1, true>(x);
   if (HI > 1) x = xor_sum<2, true>(x);
   if (HI > 2) x = xor_sum<4, true>(x);
   if (HI > 3) x = xor_sum<8, true>(x);
   if (HI > 4) x = xor_sum<16, true>(x);
   if (HI > 5) x = xor_sum<32, true>(x);
   return x;
}

// all-reduce of NVAL values over lane bits [LO, hi): LO compile-time, hi wave-uniform.
// Bits 4 and 5 together go through the matrix pipe (sum_bits45).  Bits 2 and 3 together, when the lanes below are
// not uniform, go as a rotate butterfly inside the 16-lane row -- y = x + ror8(x), z = y + ror4(y): the second
// step needs one DPP move per half where a true xor 4 needs two, and every lane still adds the same pairs.
template <int LO, int NVAL, class T>
__device__ __forceinline__ void high_bits_sum(T (&x)[NVAL], int hi)
{
#define SB_STEP(BIT)                                                                \
   {                                                                                \
      _Pragma("unroll") for (int v = 0; v < NVAL; ++v)                              \
         x[v] = xor_sum<(1 << BIT), (LO == 0)>(x[v]);                               \
   }
   if (LO <= 0 && hi > 0) SB_STEP(0)
   if (LO <= 1 && hi > 1) SB_STEP(1)
   if (LO > 0 && LO <= 2 && hi > 3) {
      SB_STEP(3)
#pragma unroll
      for (int v = 0; v < NVAL; ++v) x[v] += row_ror4(x[v]);
   } else {
      if (LO <= 2 && hi > 2) SB_STEP(2)
      if (LO <= 3 && hi > 3) SB_STEP(3)
   }
   if (LO <= 4 && hi > 5) {
#pragma unroll
      for (int v = 0; v < NVAL; ++v) x[v] = sum_bits45(x[v]);
   } else {
      if (LO <= 4 && hi > 4) SB_STEP(4)
      if (LO <= 5 && hi > 5) SB_STEP(5)
   }
#undef SB_STEP
}

// all-reduce of NVAL values over the TOP `lb` lane bits [6 - lb, 6) (lb wave-uniform): bits 4 and 5 through the
// matrix pipe, the ones below as steps inside the 16-lane row (bit 3: rotate by 8 = xor 8; bit 2: rotate by 4
// of the result, which is symmetric under the rotation by 8 by then; bits 1 and 0: quad permutes).  The lanes
// below bit 6 - lb hold different data throughout.
template <int NVAL, class T>
__device__ __forceinline__ void top_bits_sum(T (&x)[NVAL], int lb)
{
   if (lb == 1) {
#pragma unroll
      for (int v = 0; v < NVAL; ++v) x[v] = sum_xor32(x[v]);
      return;
   }
   if (lb >= 3) {
#pragma unroll
      for (int v = 0; v < NVAL; ++v) x[v] += row_ror8(x[v]);
   }
   if (lb >= 4) {
#pragma unroll
      for (int v = 0; v < NVAL; ++v) x[v] += row_ror4(x[v]);
   }
   if (lb >= 5) {
#pragma unroll
      for (int v = 0; v < NVAL; ++v) x[v] = xor_sum<2, false>(x[v]);
   }
   if (lb >= 6) {
#pragma unroll
      for (int v = 0; v < NVAL; ++v) x[v] = xor_sum<1, false>(x[v]);
   }
   if (lb >= 2) {
#pragma unroll
      for (int v = 0; v < NVAL; ++v) x[v] = sum_bits45(x[v]);
   }
}

// ------------------------------------------------------------------ the matrix lane map
// Where the lanes of a group sit inside the wave decides what its all-reduces cost.  One v_mfma_f64_4x4x4_4b sums
// over lane bits 4 and 5 (A = 1, B = x), and a PAIR of them -- x as A against B = 1, which also moves lane bits 0,1
// of the source to bits 4,5 of the result, then that result as B against A = 1 -- sums over lane bits 0, 1, 4 and 5:
// sixteen lanes for the issue time of six vector instructions, where four butterfly steps take twelve to twenty
// (layouts: tools/probe_mfma_layout.hip).  So the ROW lanes of a group -- the wide reduction: CPL values over up to 64
// lanes, every iteration -- take the lane bits in the order 4, 5, 0, 1 and only then the rest, and the COLUMN lanes
// (at most 8, one value per row) sit on bits 3, 2, 1, where xor 8 and the pair (8, 4) are single rotations of the
// 16-lane row:
//      CL   column-lane bits    row-lane bits, in order     (the bits a group does not use number its wave's groups)
//       1   -                   4 5 0 1 3 2
//       2   3                   4 5 0 1 2
//       4   3 2                 4 5 0 1
//       8   3 2 1               4 5 0
// except that TWO row lanes are bit 0 (one quad permute).  (One matrix instruction could sum over bit 4 alone, with
// zeros in A where bit 5 differs -- but 0 times a NaN of the other half's group is a NaN.)
// Every lane of a group still ends with bitwise identical sums: a matrix instruction adds its four terms in one
// order for all of its results, and a butterfly step adds the same two numbers in both partners.
template <int LB_CL>
__device__ __forceinline__ int row_rank(int lane, int lbGR) // the lane's row-order bits, compressed: bit t = lane bit order[t]
{
   int r = (lane >> 4) & 3;
   if (LB_CL <= 2) r |= (lane & 3) << 2;
   else r |= (lane & 1) << 2;
   if (LB_CL == 0) r |= (((lane >> 3) & 1) << 4) | (((lane >> 2) & 1) << 5);
   if (LB_CL == 1) r |= ((lane >> 2) & 1) << 4;
   if (lbGR == 1) r = ((r >> 2) & 1) | ((r & 3) << 1) | (r & ~7); // two row lanes: order 0 4 5 ...
   return r;
}
template <int LB_CL>
__device__ __forceinline__ int column_lane(int lane) { return (lane >> (4 - LB_CL)) & ((1 << LB_CL) - 1); }

// all-reduce over the column lanes
template <int LB_CL, class T>
__device__ __forceinline__ T col_lanes_sum(T x)
{
   if (LB_CL >= 1) x += row_ror8(x);
   if (LB_CL >= 2) x += row_ror4(x); // symmetric under the rotation by 8 by now: this adds the xor-4 partner's pair
   if (LB_CL >= 3) x = xor_sum<2, false>(x);
   return x;
}
template <int LB_CL, class T>
__device__ __forceinline__ T col_lanes_max(T x)
{
   if (LB_CL >= 1) x = fmax(x, row_ror8(x));
   if (LB_CL >= 2) x = fmax(x, row_ror4(x));
   if (LB_CL >= 3) x = fmax(x, xor_get<2>(x));
   return x;
}

// all-reduce of NVAL values over the first lbGR row-lane bits (lbGR wave-uniform)
template <int LB_CL, int NVAL>
__device__ __forceinline__ void row_lanes_sum(double (&x)[NVAL], int lbGR)
{
   static_assert(LB_CL <= 3, "the matrix lane map has at most 8 column lanes");
   if (lbGR >= 4) {
#pragma unroll
      for (int v = 0; v < NVAL; ++v) x[v] = __builtin_amdgcn_mfma_f64_4x4x4f64(x[v], 1.0, 0.0, 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NVAL; ++v) x[v] = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, x[v], 0.0, 0, 0, 0);
   } else if (lbGR >= 2) {
#pragma unroll
      for (int v = 0; v < NVAL; ++v) x[v] = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, x[v], 0.0, 0, 0, 0);
   }
   if (lbGR == 3 || lbGR == 1) {
#pragma unroll
      for (int v = 0; v < NVAL; ++v) x[v] = xor_sum<1, false>(x[v]);
   }
   if (LB_CL == 0 && lbGR >= 5) {
#pragma unroll
      for (int v = 0; v < NVAL; ++v) x[v] += row_ror8(x[v]);
   }
   if (LB_CL == 0 && lbGR >= 6) {
#pragma unroll
      for (int v = 0; v < NVAL; ++v) x[v] += row_ror4(x[v]);
   }
   if (LB_CL == 1 && lbGR >= 5) {
#pragma unroll
      for (int v = 0; v < NVAL; ++v) x[v] = xor_sum<4, false>(x[v]);
   }
}
// fp32 has no matrix instruction that sums: the same bits as butterfly steps
template <int LB_CL, int NVAL>
__device__ __forceinline__ void row_lanes_sum(float (&x)[NVAL], int lbGR)
{
#define SB_STEP(MASKV)                                                                          \
   {                                                                                            \
      _Pragma("unroll") for (int v = 0; v < NVAL; ++v) x[v] = xor_sum<MASKV, false>(x[v]);      \
   }
   if (lbGR >= 2) SB_STEP(16)
   if (lbGR >= 2) SB_STEP(32)
   if (lbGR >= 3 || lbGR == 1) SB_STEP(1)
   if (lbGR >= 4) SB_STEP(2)
   if (LB_CL == 0 && lbGR >= 5) SB_STEP(8)
   if (LB_CL == 0 && lbGR >= 6) SB_STEP(4)
   if (LB_CL == 1 && lbGR >= 5) SB_STEP(4)
#undef SB_STEP
}

#ifdef SB_STAMPS
// Diagnostic build only (make stamps): per-wave cycle stamps, never compiled into the
// product library.  [wave][8] = {start, after class lookup, first refill done, end,
// batches, iterations, refill cycles total, events cycles total}
constexpr int kStampWaves = 1 << 16;
__device__ unsigned long long sb_debug_stamps[kStampWaves * 8];
__device__ __forceinline__ unsigned long long sb_now() { return __builtin_amdgcn_s_memrealtime(); } // 100 MHz, device-wide
// every translation unit has its own copy of the stamp buffer: each kernel file exports a reader of its copy and
// sbgpu_debug_read_stamps merges them (a wave slot is written by one kernel only)
#define SB_DEFINE_STAMP_READER(NAME)                                                        \
   hipError_t read_stamps_##NAME(void *out, size_t bytes)                                   \
   {                                                                                        \
      return hipMemcpyFromSymbol(out, HIP_SYMBOL(sb_debug_stamps), bytes);                  \
   }
hipError_t read_stamps_wave_h(void *out, size_t bytes);
hipError_t read_stamps_wave_1(void *out, size_t bytes);
hipError_t read_stamps_wave_2(void *out, size_t bytes);
hipError_t read_stamps_block(void *out, size_t bytes);
hipError_t read_stamps_block_tall(void *out, size_t bytes);
#endif

// any lane of the wave: the lane mask itself compared with zero (one scalar instruction; __any goes through a
// v_cndmask / v_cmp pair)
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

constexpr int kBlockWaves = 4; // block form: 256 lanes, one wave per SIMD, up to 512 VGPRs each

// NWAVES = 0: wave form; NWAVES = 4: block form.  R is the register-tile capacity
// in rows per row lane; the block form skips the row blocks a locus does not need.
// HIMAP (wave form only): the row lanes of a group are the TOP lane bits and the groups of a wave are
// interleaved between them and the column lanes -- lane = gc | group << log2(CL) | gr << (6 - log2(GR)) -- so that
// the all-reduce over the row lanes runs over lane bits 4 and 5 first, which one matrix instruction covers
// (sum_bits45), then 3, 2, ... as rotate steps inside the 16-lane rows.
template <class T, int CPL, int CL, int R, int NWAVES, bool HIMAP = false>
__device__ __forceinline__ void em_tile_body(const EmArgsT<T> &a, const ClassArgs &cls, const int lbG, T *s_red)
{
   static_assert(!(HIMAP && NWAVES > 0), "the high-bit lane map is a wave form");
   constexpr bool BLOCK = NWAVES > 0;
   constexpr int NW = BLOCK ? NWAVES : 1;                  // waves per group
   constexpr int LB_CL = ilog2(CL);
   constexpr int NV = CPL; // values in the per-iteration column reduce
   // (the lane id goes through an empty asm: one kernel holds twenty instantiations of this body behind a switch, and
   // the compiler otherwise hoists every instantiation's lane-derived addresses in front of the switch -- dozens of
   // live registers that it then spills)
   int lane_opaque = threadIdx.x & 63;
   asm volatile("" : "+v"(lane_opaque));
   const int lane = lane_opaque;
   const int wave_id = threadIdx.x >> 6;
   const int GW = BLOCK ? 64 : (1 << lbG);                 // lanes of the group inside one wave
   const int lbGW = BLOCK ? 6 : lbG;
   const int G = BLOCK ? 64 * NW : GW;
   const int GR = G >> LB_CL;                              // row lanes per group
   const int lbGR = lbGW - LB_CL;                          // (wave form) log2 of the row lanes
   // HIMAP: the later phases' map; else the matrix lane map (see row_rank)
   constexpr int LB_MM = HIMAP ? 0 : LB_CL; // (the matrix map's helpers are not instantiated for HIMAP's 16 column lanes)
   const int rrank = row_rank<LB_MM>(lane, lbGR);
   const int grp = HIMAP ? ((lane >> LB_CL) & ((64 >> lbGW) - 1)) : (rrank >> lbGR); // (wave form) group inside the wave
   const int gc = HIMAP ? (lane & (CL - 1)) : column_lane<LB_MM>(lane);
   const int gr = HIMAP ? (lane >> (6 - lbGR)) : (BLOCK ? (rrank | (wave_id << lbGR)) : (rrank & ((1 << lbGR) - 1)));
   const int g = gc | (gr << LB_CL);                       // index inside the group; 0 = its leader
   // all-reduce over the column lanes of the group
   auto col_sum = [&](T x) -> T {
      if constexpr (HIMAP) return low_bits_sum<LB_CL>(x);
      else return col_lanes_sum<LB_MM>(x);
   };
   bool batch_taken = false; // the one refill has happened
   int phase = 0;
   int r_used = R; // block form: rows per row lane the current locus needs (workgroup-uniform)
#ifdef SB_STAMPS
   unsigned long long st_refill = 0, st_events = 0, st_first = 0, st_batches = 0, st_iters = 0, st_t = 0;
#endif

   // all-reduce over the row lanes of the group
   // (lb_tag: the number of row-lane bits as a compile-time constant where the caller has one -- the steady-state
   // loop is instantiated for the common values, so that no branch on it is left inside -- else -1: use lbGR)
   auto row_lane_sum = [&](auto &x, auto nval_tag, auto lb_tag) {
      constexpr int NVAL = decltype(nval_tag)::value;
      constexpr int LBC = decltype(lb_tag)::value;
      const int lb = LBC >= 0 ? LBC : lbGR;
      if constexpr (HIMAP) top_bits_sum<NVAL>(x, lb);
      else row_lanes_sum<LB_MM, NVAL>(x, lb);
      if (BLOCK) {
         // cross-wave: [2 phases][NV values][CL column lanes][NW]; every lane then adds
         // the NW partials in the same order
         T *buf = s_red + (size_t)phase * (NV * CL * NW);
         if (rrank == 0) { // one lane per column lane
#pragma unroll
            for (int v = 0; v < NVAL; ++v) buf[(v * CL + gc) * NW + wave_id] = x[v];
         }
         __syncthreads();
         // the partials of a few values at a time: all NVAL x NW of them in flight together cost the block
         // kernels their last free registers (and then some: scratch spills inside the iteration)
         constexpr int kChunk = NVAL < 2 ? NVAL : 2;
#pragma unroll
         for (int v0 = 0; v0 < NVAL; v0 += kChunk) {
            T part[kChunk][NW];
#pragma unroll
            for (int u = 0; u < kChunk; ++u) {
               if (v0 + u < NVAL) {
                  const T *p = buf + ((v0 + u) * CL + gc) * NW;
#pragma unroll
                  for (int w = 0; w < NW; ++w) part[u][w] = p[w];
               }
            }
#pragma unroll
            for (int u = 0; u < kChunk; ++u) {
               if (v0 + u < NVAL) {
                  T sum = part[u][0];
#pragma unroll
                  for (int w = 1; w < NW; ++w) sum += part[u][w];
                  x[v0 + u] = sum;
               }
            }
         }
         phase ^= 1; // T-buffered: one barrier per round is enough
      }
   };

   T F[R][CPL];
   T scale[CPL]; // 1 / column sum over the kept rows (:466-478), a zero column stays zero; used once, by normalise_tile
   auto column_scale = [&]() {
      T cs[CPL];
#pragma unroll
      for (int jj = 0; jj < CPL; ++jj) {
         T sum = T(0);
#pragma unroll
         for (int r = 0; r < R; ++r)
            if (!BLOCK || (r & ~3) < r_used) sum += F[r][jj];
         cs[jj] = sum;
      }
      row_lane_sum(cs, std::integral_constant<int, CPL>(), std::integral_constant<int, -1>());
#pragma unroll
      for (int jj = 0; jj < CPL; ++jj) scale[jj] = (cs[jj] == T(0)) ? T(0) : T(1) / cs[jj];
   };
   // F <- F * scale, once, when the first iteration is over (all groups of a wave reach that point together): what the
   // reference does (:466-478), and it keeps a multiply per column off the per-iteration path
   auto normalise_tile = [&](bool mine) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
         if (BLOCK && (r & ~3) >= r_used) continue;
#pragma unroll
         for (int jj = 0; jj < CPL; ++jj) F[r][jj] = mine ? F[r][jj] * scale[jj] : F[r][jj];
      }
   };
   T nn[R];      // n_i as T (obs_d, estimate.cpp:418-419); 0 for dropped rows
   // A row that init() dropped (estimate.cpp:377-390), or that lies beyond the locus, has F = 0 and n = 0 here, hence
   // the denominator +0.0 -- which must not look like the reference's `denom == 0` of a KEPT row.  Its dot product
   // starts from dfix = 1 instead of 0 (the first FMA's addend: no instruction), so its denominator is CL and its
   // weight 0 / CL = 0.
   // (batched reciprocals: not for the tall tile, which has no registers to spare for a block's)
   constexpr bool kBatchDiv = std::is_same<T, double>::value && R * CPL <= 64;
   // (the tall tile has no registers for dfix either: it selects the weight of such a row away)
   constexpr bool kDfix = R * CPL <= 64;
   T dfix[kDfix ? R : 1];
   bool act[R];
   T theta[CPL];
   T theta0 = T(0);
   int it = 0; // iterations done; wave-uniform: the groups of a wave (the one batch it serves) start together
   int niso = 0;
   int locus = 0;
   int64_t iso_base = 0;
   bool have = false;       // the group currently owns a locus

#pragma unroll
   for (int r = 0; r < R; ++r) {
      nn[r] = T(0);
      if (kDfix) dfix[r] = T(1);
      act[r] = false;
#pragma unroll
      for (int j = 0; j < CPL; ++j) F[r][j] = T(0);
   }
#pragma unroll
   for (int j = 0; j < CPL; ++j) {
      theta[j] = T(0);
      scale[j] = T(1);
   }

   for (;;) {
      // ---------------------------------------------------------------- refill
      // Wave-synchronous: when every group of the wave (block form: the workgroup)
      // is idle, each group pulls its next locus from the class list.  All lanes
      // load together, so F, theta, ... are simply overwritten.
      if (BLOCK ? !have : !wave_any(have)) {
#ifdef SB_STAMPS
         st_t = sb_now();
#endif
         // static assignment: this workgroup serves batch `cls.batch` of its class and nothing else (no cursor to
         // pull from, hence nothing to zero between runs)
         if (batch_taken) break;
         batch_taken = true;
         const int idx = BLOCK ? cls.batch : cls.batch * (64 >> lbGW) + grp;
         if (BLOCK ? (idx >= cls.n) : !wave_any(idx < cls.n)) break;
         const bool got = idx < cls.n;
         const int loc = got ? cls.loci[idx] : 0; // idle groups shadow locus 0, results discarded
         const int64_t r0 = a.row_off[loc];
         const int nrow = (int)(a.row_off[loc + 1] - r0);
         const int64_t ib = a.iso_off[loc];
         const int ni = (int)(a.iso_off[loc + 1] - ib);
         const T *Fg = a.F + a.f_off[loc];
         if (BLOCK) {
            // row blocks the locus does not reach are skipped (workgroup-uniform)
            r_used = (nrow + GR - 1) / GR;
            r_used = r_used < 1 ? 1 : (r_used > R ? R : r_used);
         }
         // EmSolver::init, estimate.cpp:366-391
         T red[2];
         red[0] = T(0); // sum of ALL counts (theta0 precedes the row drop, :374-375)
         red[1] = T(0); // number of kept rows
#pragma unroll
         for (int r = 0; r < R; ++r) {
            if (BLOCK && (r & ~3) >= r_used) continue; // row block not needed by this locus
            const int i = r * GR + gr;
            const bool valid = got && i < nrow;
            // clamped indices: loads stay inside the locus (or touch nothing when it
            // has no rows), no per-element branches
            const int ic = (i < nrow) ? i : (nrow > 0 ? nrow - 1 : 0);
            T cnt = T(0);
            if (nrow > 0) cnt = (T)a.count[r0 + ic];
            cnt = valid ? cnt : T(0);
            if (gc == 0) red[0] += cnt;
            T v[CPL];
            T mx = T(0);
#pragma unroll
            for (int jj = 0; jj < CPL; ++jj) {
               const int j = gc * CPL + jj;
               const int jc = (j < ni) ? j : ni - 1;
               T x = T(0);
               if (nrow > 0) x = Fg[(int64_t)ic * ni + jc];
               x = (valid && j < ni) ? x : T(0);
               v[jj] = x;
            }
            if (a.row_bias && nrow > 0) { // (launch-uniform) config 5: the biased weight is the operand from here on
               const T rb = a.row_bias[r0 + ic];
#pragma unroll
               for (int jj = 0; jj < CPL; ++jj) {
                  const int j = gc * CPL + jj;
                  const int jc = (j < ni) ? j : ni - 1;
                  v[jj] *= bias_factor(rb * a.iso_bias[ib + jc]);
               }
            }
#pragma unroll
            for (int jj = 0; jj < CPL; ++jj) mx = fmax(mx, v[jj]);
            // any weight of the row > 1e-5 (:380), over all column lanes
            if constexpr (HIMAP) {
               if (CL >= 2) mx = fmax(mx, xor_get<1>(mx));
               if (CL >= 4) mx = fmax(mx, xor_get<2>(mx));
               if (CL >= 8) mx = fmax(mx, xor_get<4>(mx));
               if (CL >= 16) mx = fmax(mx, xor_get<8>(mx));
            } else {
               mx = col_lanes_max<LB_MM>(mx);
            }
            const bool keep = mx > (T)kRowEps;
            if (gc == 0 && keep) red[1] += T(1);
            if (kDfix) dfix[r] = keep ? T(0) : T(1);
            act[r] = keep;
            nn[r] = keep ? cnt : T(0);
#pragma unroll
            for (int jj = 0; jj < CPL; ++jj) F[r][jj] = keep ? v[jj] : T(0);
         }
         // group totals: over the column lanes, then over the row lanes
         red[0] = col_sum(red[0]);
         red[1] = col_sum(red[1]);
         row_lane_sum(red, std::integral_constant<int, 2>(), std::integral_constant<int, -1>());
         theta0 = red[0] / (T)ni; // :375, IEEE division
#pragma unroll
         for (int jj = 0; jj < CPL; ++jj) {
            theta[jj] = (gc * CPL + jj < ni) ? theta0 : T(0);
            scale[jj] = T(1);
         }
         it = 0;
         locus = loc;
         niso = ni;
         iso_base = ib;
         if (cls.resume) {
            // state saved at the previous phase's limit: theta and the iteration count;
            // the column scale is recomputed exactly as it was computed the first time
            // (the same for every locus of the wave: they were all suspended at the previous phase's limit)
            it = __builtin_amdgcn_readfirstlane(got ? a.iters[loc] : 0);
#pragma unroll
            for (int jj = 0; jj < CPL; ++jj)
               theta[jj] = (gc * CPL + jj < ni) ? a.theta[ib + gc * CPL + jj] : T(0);
            column_scale();
            normalise_tile(true);
         }
         const bool empty = got && red[1] == T(0);
         if (empty) {
            // init() == false (:391): theta = theta0, the caller drops the locus
            if (g == 0) {
               a.status[loc] = kStInitEmpty;
               a.iters[loc] = 0;
            }
            if (gr == 0) {
#pragma unroll
               for (int jj = 0; jj < CPL; ++jj)
                  if (gc * CPL + jj < ni) a.theta[ib + gc * CPL + jj] = theta0;
            }
         }
         have = got && !empty;
#ifdef SB_STAMPS
         st_refill += sb_now() - st_t;
         st_batches += 1;
         if (!st_first) st_first = sb_now();
#endif
      }
      if (BLOCK ? !have : !wave_any(have)) continue; // e.g. only init()==false loci: pull again

      // ------------------------------------------------------ steady-state loop
      // Runs until some group of the wave needs attention (first iteration's
      // column normalisation, convergence, zero denominator, iteration limit).
      // F is read-only in here and nothing touches memory.  Two iterations per
      // trip, theta ping-ponging between `theta` and `nt`, so that no copy or
      // select sits on the per-iteration path.
      T nt[NV]; // next_theta of the own columns
      bool dz, conv, special;
      unsigned long long special_mask = 0;
      const unsigned long long have_mask = __builtin_amdgcn_ballot_w64(have); // (have changes only outside the loop)
      T d2_last; // ||next - theta||^2 of the iteration just done
      // one EM iteration: reads tin, writes tout (all lanes, no predication)
      auto iterate = [&](const T *tin, T *tout, auto exact_tag, auto lb_tag) {
         constexpr bool kExact = decltype(exact_tag)::value;
         T acc[NV];
#pragma unroll
         for (int v = 0; v < NV; ++v) acc[v] = T(0);
         // A zero denominator of a kept row (:451) is not tested row by row: 1 / 0 is infinite, the Newton steps turn
         // it into a NaN, the row's weight n * NaN is a NaN whatever n is, NaN * F poisons every column sum of the
         // lane -- and the same row denominator is seen by all column lanes -- so every next_theta of the group and
         // with them ||next - theta||^2 come out as NaN, which is tested once per iteration below.  Nothing else
         // makes a NaN here: denominators are sums of products of non-negative finite numbers.
         // rows in blocks of 4 to bound the live temporaries
#pragma unroll
         for (int rb = 0; rb < R; rb += 4) {
            if (BLOCK && rb >= r_used) continue;
            T d[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
               if (rb + q < R) {
                  T sum = kDfix ? dfix[rb + q] : T(0);
#pragma unroll
                  for (int jj = 0; jj < CPL; ++jj) sum = fma_t(F[rb + q][jj], tin[jj], sum); // :450
                  d[q] = col_sum(sum);
               }
            }
            if constexpr (kBatchDiv && !kExact) {
               // the block's reciprocals from one v_rcp_f64 (see batch_reciprocals)
               const int nb = R - rb < 4 ? R - rb : 4; // a constant once the loop is unrolled
               double dd[4], inv[4];
#pragma unroll
               for (int q = 0; q < 4; ++q) dd[q] = q < nb ? d[q] : 1.0;
               if (nb == 4) batch_reciprocals<4>(dd, inv);
               else if (nb == 3) batch_reciprocals<3>(dd, inv);
               else if (nb == 2) batch_reciprocals<2>(dd, inv);
               else batch_reciprocals<1>(dd, inv);
#pragma unroll
               for (int q = 0; q < 4; ++q) {
                  if (q >= nb) continue;
                  const int r = rb + q;
                  const T w = nn[r] * inv[q];
#pragma unroll
                  for (int jj = 0; jj < CPL; ++jj) acc[jj] = fma_t(w, F[r][jj], acc[jj]);
               }
            } else {
#pragma unroll
               for (int q = 0; q < 4; ++q) {
                  if (rb + q < R) {
                     const int r = rb + q;
                     T w = fast_div(nn[r], d[q]);
                     if (!kDfix) w = act[r] ? w : T(0);
#pragma unroll
                     for (int jj = 0; jj < CPL; ++jj) acc[jj] = fma_t(w, F[r][jj], acc[jj]);
                  }
               }
            }
         }
         row_lane_sum(acc, std::integral_constant<int, NV>(), lb_tag);
         T p2 = T(0);
#pragma unroll
         for (int jj = 0; jj < CPL; ++jj) {
            const T t = tin[jj] * acc[jj]; // next_theta_j = sum_i U_ij, :454-464
            const T df = t - tin[jj];
            p2 = fma_t(df, df, p2); // :479
            tout[jj] = t;
         }
         const T d2 = col_sum(p2);
         // ||next - theta||_2 < 1e-2 (:479-480) tested on the square: kThetaLimitSq is the largest T whose
         // (correctly rounded) square root is below 1e-2, so this is the same predicate as sqrt(d2) < 1e-2 of the
         // reference, the oracle and the streaming / wide kernels, for every d2
         // One compare on the per-iteration path: "not above the limit" is converged OR NaN -- some kept row had a
         // zero denominator (:451), see above; the event code tells the two apart.  The iteration count is a scalar.
         // The groups that need attention are kept as a lane MASK in scalar registers (the compare writes one; the
         // rest is scalar arithmetic, and "any?" is one scalar compare with zero).
         d2_last = d2;
         const bool edge = (unsigned)(it - 1) >= (unsigned)(cls.it_limit - 2); // it == 0 or it + 1 == limit, one compare
         const unsigned long long stop_mask = __builtin_amdgcn_ballot_w64(!(d2 > ThetaLimitSq<T>::value));
         special_mask = have_mask & (edge ? ~0ull : stop_mask);
      };
      auto lane_of = [&](unsigned long long m) -> bool { return (m >> lane) & 1ull; };
      auto classify = [&]() {
         conv = d2_last <= ThetaLimitSq<T>::value;
         dz = __builtin_isnan(d2_last);
      };
      auto run = [&](auto lb_tag) {
      for (;;) {
         iterate(theta, nt, std::false_type(), lb_tag); // old in theta, new in nt
#ifdef SB_STAMPS
         st_iters += 1;
#endif
         if (special_mask != 0ull) {
            // groups that simply finished an iteration move on (:481); the special ones keep
            // (old, new) = (theta, nt) for the event handling below
            special = lane_of(special_mask);
            const bool adv = have && !special;
#pragma unroll
            for (int jj = 0; jj < CPL; ++jj) theta[jj] = adv ? nt[jj] : theta[jj];
            break;
         }
         ++it;
         iterate(nt, theta, std::false_type(), lb_tag); // old in nt, new in theta
#ifdef SB_STAMPS
         st_iters += 1;
#endif
         if (special_mask != 0ull) {
            // special groups: bring (old, new) back to (theta, nt); the others already hold
            // their new theta in `theta`
            special = lane_of(special_mask);
#pragma unroll
            for (int jj = 0; jj < CPL; ++jj) {
               const T o = nt[jj], n2 = theta[jj];
               theta[jj] = special ? o : n2;
               nt[jj] = special ? n2 : o;
            }
            break;
         }
         ++it;
      }
      };
      // the loop once per common group height (wave form; the block form's is a constant anyway)
      if (BLOCK || HIMAP) run(std::integral_constant<int, -1>());
      else if (lbGR == 4 && LB_CL <= 2) run(std::integral_constant<int, (LB_CL <= 2 ? 4 : -1)>());
      else if (lbGR == 3) run(std::integral_constant<int, 3>());
      else if (lbGR == 5 && LB_CL <= 1) run(std::integral_constant<int, (LB_CL <= 1 ? 5 : -1)>());
      else if (lbGR == 2) run(std::integral_constant<int, 2>());
      else if (lbGR == 1) run(std::integral_constant<int, 1>());
      else if (lbGR == 6 && LB_CL == 0) run(std::integral_constant<int, (LB_CL == 0 ? 6 : -1)>());
      else if (lbGR == 0) run(std::integral_constant<int, 0>()); // (a group of CL lanes: no row lanes to sum over)
      else run(std::integral_constant<int, -1>());

      // ------------------------------------------------------- per-group events
      // (`it` still counts the iterations BEFORE the one whose results are looked at here)
      classify();
      if constexpr (kBatchDiv) {
         // a NaN of the batched reciprocals is a zero denominator -- or a product of several tiny ones that left the
         // exponent range: the groups that saw one repeat the iteration with a division per row and take ITS
         // verdict and thetas (the other groups keep what they have: their results never depend on their neighbours)
         const bool suspect = special && dz;
         if (BLOCK ? suspect : wave_any(suspect)) {
            T keep_nt[CPL];
            const bool keep_conv = conv, keep_special = special;
            const unsigned long long keep_mask = special_mask;
#pragma unroll
            for (int jj = 0; jj < CPL; ++jj) keep_nt[jj] = nt[jj];
            iterate(theta, nt, std::true_type(), std::integral_constant<int, -1>());
            classify();
#pragma unroll
            for (int jj = 0; jj < CPL; ++jj) nt[jj] = suspect ? nt[jj] : keep_nt[jj];
            conv = suspect ? conv : keep_conv;
            dz = suspect && dz;
            special = keep_special;
            special_mask = keep_mask;
         }
      }
      // first iteration done: switch to the column-normalised problem (:466-478)
      const bool norm = special && !dz && it == 0;
      if (BLOCK ? norm : wave_any(norm)) {
         T keep_scale[CPL];
#pragma unroll
         for (int jj = 0; jj < CPL; ++jj) keep_scale[jj] = scale[jj];
         column_scale();
#pragma unroll
         for (int jj = 0; jj < CPL; ++jj) scale[jj] = norm ? scale[jj] : keep_scale[jj];
         normalise_tile(norm);
      }
      if (special) {
         bool finished = false;
         int32_t st = kStOk;
         if (dz) {
            // run() returns false before touching _theta (:451-453): theta0 survives
            finished = true;
            st = kStDenomZero;
#pragma unroll
            for (int jj = 0; jj < CPL; ++jj) theta[jj] = theta0;
         } else if (conv) {
            finished = true; // break before theta = next_theta (:480)
         } else {
#pragma unroll
            for (int jj = 0; jj < CPL; ++jj) theta[jj] = nt[jj]; // :481
            if (it + 1 == cls.it_limit) {
               // the 1000-iteration cap (estimate.hpp:237), or only this phase's limit:
               // then the locus is suspended and handed to the next phase
               finished = true;
               st = (cls.it_limit >= kMaxIter) ? kStMaxIter : kStRunning;
            }
         }
         if (finished) {
            if (g == 0) {
               a.status[locus] = st;
               a.iters[locus] = it + 1;
               if (st == kStRunning) {
                  const int c2 = cls.route[locus];
                  cls.out[cls.next_table[c2].loci_off + atomicAdd(cls.out_count + c2, 1)] = locus;
               }
            }
            if (gr == 0) {
#pragma unroll
               for (int jj = 0; jj < CPL; ++jj)
                  if (gc * CPL + jj < niso) a.theta[iso_base + gc * CPL + jj] = theta[jj];
            }
            have = false;
         }
      }
      ++it; // the iteration that ended the loop is done, for every group
#ifdef SB_STAMPS
      st_events += sb_now() - st_t;
#endif
   }
#ifdef SB_STAMPS
   if (lane == 0 && !BLOCK) {
      const unsigned wg = (blockIdx.x * (blockDim.x >> 6) + wave_id) & (kStampWaves - 1);
      sb_debug_stamps[wg * 8 + 2] = st_first;
      sb_debug_stamps[wg * 8 + 3] = sb_now();
      sb_debug_stamps[wg * 8 + 4] = st_batches;
      sb_debug_stamps[wg * 8 + 5] = st_iters;
      sb_debug_stamps[wg * 8 + 7] = st_events;
   }
#endif
}

// One launch serves every size class of a batch: the workgroup looks its class
// up in the descriptor table (sorted by first block) and jumps to the matching
// instantiation.  `shape` packs (column-layout index, rows-per-lane multiplier,
// log2 lanes per group).
constexpr int kLayouts = 6; // (CPL, CL): (2,1) (4,1) (8,1) (8,2) (8,4) (8,8)

// class of workgroup b: the last descriptor whose first block is <= b (wave-uniform binary search; empty classes
// share their successor's first block and are skipped that way)
__device__ __forceinline__ int find_class(const ClassDesc *table, int n_classes, int b)
{
   int lo = 0, hi = n_classes - 1;
   while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (table[mid].block_begin <= b) lo = mid;
      else hi = mid - 1;
   }
   return __builtin_amdgcn_readfirstlane(lo);
}

// what a kernel of phase p needs besides the batch (one struct, passed by value)
struct PhaseArgs {
   const ClassDesc *table;       // this launch's classes
   int32_t n_classes;
   const int32_t *lists_in;      // class lists (table[c].loci_off)
   const int32_t *n_in;          // [class] loci in the list
   int32_t n_batches;            // phase 0: batches of this launch (the table's first-batch column is the host's)
   const int32_t *total_blocks;  // later phases: number of batches, computed on the device; nullptr: n_batches
   int32_t *lists_out;           // next phase
   int32_t *n_out;
   const int32_t *route;
   const ClassDesc *next_table;
   int32_t it_limit;
   int32_t resume;
};

// RH = rows per row lane in units of HALF the base tile (base: 8 rows for 2/4 columns per
// lane, 4 rows for 8): 1 = half tile (shortest iteration, most lanes per locus), 2 = base,
// 4 = double, 12 = the tall block tile.  Register budgets (waves per SIMD asked for): forcing
// more waves than the body's natural register use admits puts spills on the per-iteration
// path, which costs more than the occupancy buys.
#ifndef SB_WAVEH_OCC
#define SB_WAVEH_OCC 3
#endif
#ifndef SB_WAVE1_OCC
#define SB_WAVE1_OCC 2
#endif
#ifndef SB_WAVE2_OCC
#define SB_WAVE2_OCC 2
#endif
template <int NWAVES, int RH, class T = double>
__global__ __launch_bounds__(NWAVES > 0 ? 64 * NWAVES : 64,
                              NWAVES > 0 ? (RH <= 2 ? 2 : 1)
                                         : (RH == 1 ? SB_WAVEH_OCC : (RH == 2 ? SB_WAVE1_OCC : SB_WAVE2_OCC))) void em_fused_kernel(
   EmArgsT<T> a, PhaseArgs ph)
{
   __shared__ T s_red[NWAVES > 0 ? 2 * (kMaxCPLv + 1) * 8 * NWAVES : 1];
   set_flush_denormals<T>();
   // workgroup-per-locus waves have the longest iterations of the batch (LDS round + barrier):
   // they go first whenever they share a SIMD with wave-form waves
   if (NWAVES > 0) __builtin_amdgcn_s_setprio(3);
#ifdef SB_STAMPS
   const unsigned long long st0 = sb_now();
#endif
   // A workgroup serves the batches b = blockIdx.x, + gridDim.x, ... of the table (normally exactly one: the grid
   // is the number of batches).  Phase 0: the table's first-batch column comes from the host; a later phase that
   // re-packs its survivors into the same layouts: phase_prepare_kernel has written it from the survivor counts.
   const int n_batches = ph.total_blocks ? *ph.total_blocks : ph.n_batches;
   for (int b = (int)blockIdx.x; b < n_batches; b += (int)gridDim.x) {
   // block form serving a second batch (grids capped below the batch count): the next batch's first LDS round must not
   // overwrite partials a slower wave of this workgroup still reads from the last one (the layout may differ too)
   if (NWAVES > 0 && b != (int)blockIdx.x) __syncthreads();
   const int c = find_class(ph.table, ph.n_classes, b);
   const ClassDesc d = ph.table[c];
#ifdef SB_STAMPS
   if ((threadIdx.x & 63) == 0 && NWAVES == 0) { // wave kind only: the kinds share the stamp array
      const unsigned wg = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (kStampWaves - 1);
      sb_debug_stamps[wg * 8 + 0] = st0;
      sb_debug_stamps[wg * 8 + 1] = sb_now();
      // which SIMD the wave runs on: HW_ID simd 5:4, cu 11:8, sh 12, se 15:13; XCC_ID 3:0
      const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
      sb_debug_stamps[wg * 8 + 6] = ((xcc & 7u) << 10) | (((hw >> 8) & 0xFFu) << 2) | ((hw >> 4) & 3u);
   }
#endif
   ClassArgs cls;
   cls.loci = ph.lists_in + d.loci_off;
   cls.n = ph.n_in[c];
   cls.out = ph.lists_out;
   cls.out_count = ph.n_out;
   cls.route = ph.route;
   cls.next_table = ph.next_table;
   cls.it_limit = ph.it_limit;
   cls.resume = ph.resume;
   cls.batch = b - d.block_begin;
   const int layout = d.shape & 0xFF;
   const int lbG = (d.shape >> 16) & 0xFF;
   // layout = (CPL - 1) + 8 * log2(CL): exact columns per lane (no padding to a power of two),
   // CL = 1 for up to 8 isoforms, else 2 / 4 / 8 column lanes of 5..8 columns each.
   // Rows per row lane: RH halves of the base height (16, 8, 8, 8, 6, 6, 4, 4 for CPL = 1..8).
#define SB_BODY(CPLV, CLV, RHALF)                                                                  \
   case (CPLV - 1) + 8 * ilog2(CLV):                                                               \
      em_tile_body<T, CPLV, CLV, tile_rows(CPLV, RHALF, RH), NWAVES>(a, cls, lbG, s_red);          \
      break;
   switch (layout) {
      SB_BODY(1, 1, 8)
      SB_BODY(2, 1, 4)
      SB_BODY(3, 1, 4)
      SB_BODY(4, 1, 4)
      SB_BODY(5, 1, 3)
      SB_BODY(6, 1, 3)
      SB_BODY(7, 1, 2)
      SB_BODY(8, 1, 2)
      SB_BODY(5, 2, 3)
      SB_BODY(6, 2, 3)
      SB_BODY(7, 2, 2)
      SB_BODY(8, 2, 2)
      SB_BODY(5, 4, 3)
      SB_BODY(6, 4, 3)
      SB_BODY(7, 4, 2)
      SB_BODY(8, 4, 2)
      SB_BODY(5, 8, 3)
      SB_BODY(6, 8, 3)
      SB_BODY(7, 8, 2)
      SB_BODY(8, 8, 2)
   default: break;
   }
#undef SB_BODY
   } // batches
}

// ------------------------------------------------------------------ the later phases' kernel
// "Lane-rich" wave layouts for the loci that are still running when most of the batch has converged: few
// columns per lane (1..4) on 1..16 column lanes, few rows per lane (1..8), the row lanes in the top lane bits
// (HIMAP).  An iteration then is a few dozen instructions instead of ~200, at the price of more lanes per locus
// -- which are free by then.  shape = (CPL - 1) | log2(CL) << 2 | R << 8 | lbG << 16.
// A workgroup (one wave) serves the batches b = blockIdx.x, + gridDim.x, ... of the table; in the later phases
// the table's first-block column and the number of batches are computed on the device from the survivor counts.
#ifndef SB_LAT_MAXTILE
#define SB_LAT_MAXTILE 32
#endif
constexpr int kLatMaxTile = SB_LAT_MAXTILE; // R * CPL: elements of F a lane holds at most
constexpr int lat_shape(int cpl, int lb_cl, int r) { return (cpl - 1) | (lb_cl << 2) | (r << 8); }
#ifndef SB_LAT_OCC
#define SB_LAT_OCC 2
#endif
#ifdef SB_COMPILE_LAT_KERNEL
__global__ __launch_bounds__(64, SB_LAT_OCC) void em_lat_kernel(EmArgs a, PhaseArgs ph)
{
   set_fp64_flush_denormals();
   const int n_batches = ph.total_blocks ? *ph.total_blocks : ph.n_batches;
   for (int b = (int)blockIdx.x; b < n_batches; b += (int)gridDim.x) {
      const int c = find_class(ph.table, ph.n_classes, b);
      const ClassDesc d = ph.table[c];
      ClassArgs cls;
      cls.loci = ph.lists_in + d.loci_off;
      cls.n = ph.n_in[c];
      cls.out = ph.lists_out;
      cls.out_count = ph.n_out;
      cls.route = ph.route;
      cls.next_table = ph.next_table;
      cls.it_limit = ph.it_limit;
      cls.resume = ph.resume;
      cls.batch = b - d.block_begin;
      // every lane-rich class gives a locus the whole wave (lbG = 6, a literal below: the reductions' step
      // selection folds at compile time)
#define SB_LAT(CPLV, CLV, RV)                                                             \
   case lat_shape(CPLV, ilog2(CLV), RV):                                                  \
      if constexpr (CPLV * RV <= kLatMaxTile)                                             \
         em_tile_body<double, CPLV, CLV, RV, 0, true>(a, cls, 6, nullptr);                \
      break;
#define SB_LAT_ROWS(CPLV, CLV)                                                            \
   SB_LAT(CPLV, CLV, 1) SB_LAT(CPLV, CLV, 2) SB_LAT(CPLV, CLV, 3) SB_LAT(CPLV, CLV, 4)    \
   SB_LAT(CPLV, CLV, 6) SB_LAT(CPLV, CLV, 8)
#define SB_LAT_COLS(CLV) SB_LAT_ROWS(1, CLV) SB_LAT_ROWS(2, CLV) SB_LAT_ROWS(3, CLV) SB_LAT_ROWS(4, CLV)
      switch (d.shape & 0xFFFF) {
         SB_LAT_COLS(1)
         SB_LAT_COLS(2)
         SB_LAT_COLS(4)
         SB_LAT_COLS(8)
         SB_LAT_COLS(16)
      default: break;
      }
#undef SB_LAT_COLS
#undef SB_LAT_ROWS
#undef SB_LAT
   }
}

// First blocks of a later phase's classes from the survivor counts the previous phase left: one workgroup.
// A class needs ceil(survivors / loci per wave) batches, loci per wave = 64 >> lbG.
__global__ __launch_bounds__(256) void phase_prepare_kernel(ClassDesc *table, const int32_t *n_in, int n_classes,
                                                            int32_t *total_blocks)
{
   __shared__ int s_scan[256];
   int carry = 0;
   for (int base = 0; base < n_classes; base += 256) {
      const int c = base + (int)threadIdx.x;
      int nb = 0;
      if (c < n_classes) {
         const int lbG = (table[c].shape >> 16) & 0xFF;
         const int lpw = 64 >> lbG;
         nb = (n_in[c] + lpw - 1) / lpw;
      }
      s_scan[threadIdx.x] = nb;
      __syncthreads();
      for (int w = 1; w < 256; w <<= 1) {
         const int v = (int)threadIdx.x >= w ? s_scan[threadIdx.x - w] : 0;
         __syncthreads();
         s_scan[threadIdx.x] += v;
         __syncthreads();
      }
      if (c < n_classes) table[c].block_begin = carry + s_scan[threadIdx.x] - nb;
      carry += s_scan[255];
      __syncthreads();
   }
   if (threadIdx.x == 0) *total_blocks = carry;
}

#endif // SB_COMPILE_LAT_KERNEL

// host-callable launchers, one translation unit per instantiation (em_kernels_*.hip)
struct FusedLaunch {
   EmArgs a;
   PhaseArgs ph;
   int n_blocks;
};
struct FusedLaunchF32 {
   EmArgsT<float> a;
   PhaseArgs ph;
   int n_blocks;
};
hipError_t launch_fused_f32(int kind, const FusedLaunchF32 &l, hipStream_t s); // kind: plan.h ClassKind (wave base / double tile, block, tall block)
hipError_t launch_fused_wave_h(const FusedLaunch &l, hipStream_t s);
hipError_t launch_fused_wave_1(const FusedLaunch &l, hipStream_t s);
hipError_t launch_fused_wave_2(const FusedLaunch &l, hipStream_t s);
hipError_t launch_fused_block(const FusedLaunch &l, hipStream_t s);
hipError_t launch_fused_block_tall(const FusedLaunch &l, hipStream_t s);
hipError_t launch_lat(const FusedLaunch &l, hipStream_t s);
hipError_t launch_phase_prepare(ClassDesc *table, const int32_t *n_in, int n_classes, int32_t *total_blocks, hipStream_t s);
hipError_t launch_stream(const EmArgs &a, const ClassArgs &c, uint8_t *row_keep, int n_blocks, size_t lds_bytes,
                         hipStream_t s);

// ============================================================= streaming kernel
// Any shape (niso <= 64*kStreamSlots): one 256-thread workgroup per locus, F is
// re-read from memory (L2 / Infinity Cache) every iteration and never modified;
// the column normalisation is carried as a per-column scale s_j folded into
// phi_j = s_j * theta_j.  LW lanes cooperate on one row (coalesced row reads),
// 64/LW rows per wave-step.
constexpr int kStreamThreads = 1024;
constexpr int kStreamSlots = 8; // column slots per lane: niso <= LW * 8 <= 512

__device__ __forceinline__ double wave_group_sum_rt(double x, int gw)
{
   // runtime group width (wave-uniform branches)
   if (gw >= 2) x += dpp_mov<kDppXor1>(x);
   if (gw >= 4) x += dpp_mov<kDppXor2>(x);
   if (gw >= 8) x += dpp_mov<kDppHalfMirror>(x);
   if (gw >= 16) x += dpp_mov<kDppMirror>(x);
   if (gw >= 32) x = sum_xor16(x);
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

#ifdef SB_COMPILE_STREAM_KERNEL
__global__ __launch_bounds__(kStreamThreads) void em_stream_kernel(EmArgs a, ClassArgs cls,
                                                                   uint8_t *row_keep /*[total rows]*/)
{
   // phi[npad] | theta[npad] | scale[npad] | accw[NWAVE][npad]
   extern __shared__ double s_dyn[];
   __shared__ double s_part[kStreamThreads / 64];
   __shared__ int s_flag;
   const int tid = threadIdx.x;
   const int lane = tid & 63;
   const int wave = tid >> 6;
   constexpr int NWAVE = kStreamThreads / 64;
   set_fp64_flush_denormals();

   for (int idx = (int)blockIdx.x; idx < cls.n; idx += (int)gridDim.x) {
      if (tid == 0) s_flag = 0;
      __syncthreads();
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
         constexpr int U = 4; // row-steps in flight per wave (independent loads first, then math)
         for (int base = wave * rps; base < nrow; base += U * NWAVE * rps) {
            double fv[U][kStreamSlots];
            double cnt[U];
            bool keepu[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
               const int i = base + u * NWAVE * rps + rsub;
               const bool valid = i < nrow;
               const int ic = valid ? i : nrow - 1;
               cnt[u] = valid ? (double)a.count[r0 + ic] : 0.0;
               keepu[u] = valid && row_keep[r0 + ic] != 0;
#pragma unroll
               for (int k = 0; k < kStreamSlots; ++k) {
                  fv[u][k] = 0.0;
                  if (k < nslot) {
                     const int j = c0 + k * lw;
                     if (valid && j < niso) fv[u][k] = Fg[(int64_t)ic * niso + j];
                  }
               }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
               double part = 0.0;
#pragma unroll
               for (int k = 0; k < kStreamSlots; ++k)
                  if (k < nslot) part = __builtin_fma(fv[u][k], phi[c0 + k * lw], part); // :450 with F' = F*scale
               const double d = wave_group_sum_rt(part, lw);
               zf |= (keepu[u] && d == 0.0) ? 1 : 0; // :451
               double w = fast_div(cnt[u], d);
               w = keepu[u] ? w : 0.0;
#pragma unroll
               for (int k = 0; k < kStreamSlots; ++k) acc[k] = __builtin_fma(w, fv[u][k], acc[k]);
            }
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

#endif // SB_COMPILE_STREAM_KERNEL

} // namespace sb
