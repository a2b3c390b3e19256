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
// Layout.  A wave is a 2-D grid of lanes like the tile kernels' groups (em_device.h, "matrix lane map"):
// CL = 16, 32 or 64 COLUMN lanes on the lane bits 4 5 0 1 (3 (2)) -- a row's denominator is summed over them by a
// PAIR of v_mfma_f64_4x4x4 (+ one or two rotations of the 16-lane row) -- and 4, 2 or 1 ROW lanes on the bits
// left (3 2): the column sums of a wave take two, one or no rotation step per column.  A lane keeps R rows x CPL
// columns in REGISTERS (64-72 doubles: up to 144 of its 256 registers) and, round 5, RL more rows x CPL columns in
// LDS (its own slots, [slot][thread]: conflict-free 8-byte reads, nothing shared) -- 21-36 doubles more per lane,
// which is what the 160 KB of a CU's LDS hold beside the exchange's arrays.  A round of the tail is bound by the
// hand-off between a locus' workgroups, not by its arithmetic, so what counts is how many rows a CU holds: a third
// more rows per workgroup are a quarter fewer workgroups per locus and a quarter fewer rounds.  Columns per lane come
// in steps of one (4 ... 8) so that a locus' width is padded by 9 % on average instead of 22 %.
// A row's count sits in LDS once per row (a double, read as a broadcast; -1 marks a row that is not part of the
// problem: dropped by init() or beyond the workgroup's rows -- its weights are 0, its denominator counts as 1); a
// wave-uniform bit per block of rows says whether any lane has to look for the mark, so neither a lane mask per row
// slot nor a select is on the common path.
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

#include <utility>

#include "em_device.h"

namespace sb {

constexpr int kWideWaves = 8;                  // 2 per SIMD: up to 256 VGPRs each (the sums over the waves are written as trees of 8)
constexpr int kWideThreads = 64 * kWideWaves;
constexpr unsigned kWideSpinLimit = 1u << 20;  // sweeps of an exchange's partials before giving up (~ seconds)
constexpr int kWideSweep = 8;                  // granules a thread has in flight per sweep
constexpr size_t kWideLdsCap = 159 * 1024;     // of the 160 KB a workgroup may ask for on gfx950

// (log2 column lanes, columns per lane, rows per lane in registers, rows per lane in LDS, register rows that share one
// v_rcp_f64, LDS rows per block): 64-72 doubles of F per lane in registers -- what fits beside the ~100 registers the
// iteration itself needs (phi and the partial sums of the lane's columns, a block's denominators and reciprocals, an
// LDS block's weights in flight, the owner thread's theta / phi / scale) -- and what the LDS holds beside
struct WideLayout {
   int lb_cl, cpl, r, rl, rblk, lblk;
};
constexpr int kWideLayouts = 13;
constexpr WideLayout wide_layout(int id)
{
   constexpr WideLayout t[kWideLayouts] = {
      {4, 4, 16, 9, 4, 2}, {4, 5, 14, 7, 4, 2}, {4, 6, 12, 6, 4, 2}, {4, 7, 10, 5, 2, 1}, {4, 8, 9, 4, 2, 1},
      {5, 5, 14, 6, 4, 2}, {5, 6, 12, 5, 4, 2}, {5, 7, 10, 4, 2, 1}, {5, 8, 9, 4, 2, 1},
      {6, 5, 14, 6, 4, 2}, {6, 6, 12, 4, 4, 2}, {6, 7, 10, 4, 2, 1}, {6, 8, 9, 3, 2, 1}};
   return t[id];
}
constexpr int wide_cols(int id) { return (1 << wide_layout(id).lb_cl) * wide_layout(id).cpl; } // 64 ... 512
constexpr int wide_rows_per_block(int id) { return kWideWaves * (64 >> wide_layout(id).lb_cl) * (wide_layout(id).r + wide_layout(id).rl); }
constexpr int kWideMaxCols = 512;
// dynamic LDS of a workgroup: phi[npad] | accw[waves][npad] | part[waves] | misc[8] doubles | zf[16 ints] |
// row[rows per block] {count, flag} | stage[2 npad] granules (the two-level exchange) | F[rl][cpl][threads] doubles
constexpr size_t wide_lds_bytes(int id)
{
   return (size_t)(wide_cols(id) * (1 + kWideWaves) + kWideWaves + 8) * sizeof(double) + 16 * sizeof(int) +
          (size_t)wide_rows_per_block(id) * 8 + (size_t)2 * wide_cols(id) * 16 +
          (size_t)wide_layout(id).rl * wide_layout(id).cpl * kWideThreads * sizeof(double);
}
constexpr bool wide_layouts_fit()
{
   for (int id = 0; id < kWideLayouts; ++id)
      if (wide_lds_bytes(id) > kWideLdsCap || wide_cols(id) > kWideMaxCols) return false;
   return true;
}
static_assert(wide_layouts_fit(), "a wide layout asks for more LDS than a workgroup may have");
// the layout that serves a locus with the fewest workgroups (then the narrowest: less to exchange); -1: too wide
inline int wide_layout_for(int64_t niso, int64_t nrow)
{
   int best = -1;
   int64_t best_g = 0;
   for (int id = 0; id < kWideLayouts; ++id) {
      if (niso > wide_cols(id)) continue;
      const int64_t rpb = wide_rows_per_block(id);
      const int64_t g = nrow <= rpb ? 1 : (nrow + rpb - 1) / rpb;
      if (best < 0 || g < best_g || (g == best_g && wide_cols(id) < wide_cols(best))) best = id, best_g = g;
   }
   return best;
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
   // (s_nop: a store of more than 8 bytes followed by a write of its data registers wants wait states, and the
   // compiler's hazard recogniser does not look into the statement)
   asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(g) : "memory");
}
__device__ __forceinline__ bool granule_ready(const u32x4 &g, unsigned round) { return g.z == round && (g.w ^ ~round) <= 1u; }
__device__ __forceinline__ unsigned granule_flag(const u32x4 &g, unsigned round) { return (g.w ^ ~round) & 1u; }
__device__ __forceinline__ double granule_value(const u32x4 &g) { return __hiloint2double((int)g.y, (int)g.x); }

// One sweep of N granules (base + off[k], k < N), all in flight together, as ONE asm statement that ends with the wait:
// the registers the loads write are outputs of a statement that has completed when the compiler sees them.  (Round 4 had
// a statement per load and a separate wait; once the register allocator spills around the exchange it is free to store
// a granule's registers between the two -- before the data has arrived -- and reload the stale words behind the wait: a
// partial that "never arrives".  Seen as a sporadic time-out the day the tile grew.)
template <int N>
__device__ __forceinline__ void granule_sweep(const char *base_ptr, const unsigned (&off)[kWideSweep], u32x4 (&gr)[kWideSweep])
{
   static_assert(N >= 1 && N <= kWideSweep && kWideSweep == 8, "sweep");
   // the base in scalar registers (it is wave-uniform; the compiler cannot always tell).  The statements start with
   // s_nop 4: a v_readfirstlane's scalar result wants five wait states before a memory instruction reads it, and the
   // compiler's hazard recogniser does not look into an asm statement (seen: the loads went out with the register
   // pair's previous contents -- a memory access fault)
   const uint64_t bp = (uint64_t)base_ptr;
   const uint64_t base = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(bp >> 32)) << 32) |
                         (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bp);
   if (N == 1)
      asm volatile("s_nop 4\n\t"
                   "global_load_dwordx4 %0, %1, %2 sc1\n\t"
                   "s_waitcnt vmcnt(0)"
                   : "=&v"(gr[0])
                   : "v"(off[0]), "s"(base)
                   : "memory");
   if (N == 2)
      asm volatile("s_nop 4\n\t"
                   "global_load_dwordx4 %0, %2, %4 sc1\n\t"
                   "global_load_dwordx4 %1, %3, %4 sc1\n\t"
                   "s_waitcnt vmcnt(0)"
                   : "=&v"(gr[0]), "=&v"(gr[1])
                   : "v"(off[0]), "v"(off[1]), "s"(base)
                   : "memory");
   if (N == 3)
      asm volatile("s_nop 4\n\t"
                   "global_load_dwordx4 %0, %3, %6 sc1\n\t"
                   "global_load_dwordx4 %1, %4, %6 sc1\n\t"
                   "global_load_dwordx4 %2, %5, %6 sc1\n\t"
                   "s_waitcnt vmcnt(0)"
                   : "=&v"(gr[0]), "=&v"(gr[1]), "=&v"(gr[2])
                   : "v"(off[0]), "v"(off[1]), "v"(off[2]), "s"(base)
                   : "memory");
   if (N == 4)
      asm volatile("s_nop 4\n\t"
                   "global_load_dwordx4 %0, %4, %8 sc1\n\t"
                   "global_load_dwordx4 %1, %5, %8 sc1\n\t"
                   "global_load_dwordx4 %2, %6, %8 sc1\n\t"
                   "global_load_dwordx4 %3, %7, %8 sc1\n\t"
                   "s_waitcnt vmcnt(0)"
                   : "=&v"(gr[0]), "=&v"(gr[1]), "=&v"(gr[2]), "=&v"(gr[3])
                   : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(base)
                   : "memory");
   if (N == 5)
      asm volatile("s_nop 4\n\t"
                   "global_load_dwordx4 %0, %5, %10 sc1\n\t"
                   "global_load_dwordx4 %1, %6, %10 sc1\n\t"
                   "global_load_dwordx4 %2, %7, %10 sc1\n\t"
                   "global_load_dwordx4 %3, %8, %10 sc1\n\t"
                   "global_load_dwordx4 %4, %9, %10 sc1\n\t"
                   "s_waitcnt vmcnt(0)"
                   : "=&v"(gr[0]), "=&v"(gr[1]), "=&v"(gr[2]), "=&v"(gr[3]), "=&v"(gr[4])
                   : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "s"(base)
                   : "memory");
   if (N == 6)
      asm volatile("s_nop 4\n\t"
                   "global_load_dwordx4 %0, %6, %12 sc1\n\t"
                   "global_load_dwordx4 %1, %7, %12 sc1\n\t"
                   "global_load_dwordx4 %2, %8, %12 sc1\n\t"
                   "global_load_dwordx4 %3, %9, %12 sc1\n\t"
                   "global_load_dwordx4 %4, %10, %12 sc1\n\t"
                   "global_load_dwordx4 %5, %11, %12 sc1\n\t"
                   "s_waitcnt vmcnt(0)"
                   : "=&v"(gr[0]), "=&v"(gr[1]), "=&v"(gr[2]), "=&v"(gr[3]), "=&v"(gr[4]), "=&v"(gr[5])
                   : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "s"(base)
                   : "memory");
   if (N == 7)
      asm volatile("s_nop 4\n\t"
                   "global_load_dwordx4 %0, %7, %14 sc1\n\t"
                   "global_load_dwordx4 %1, %8, %14 sc1\n\t"
                   "global_load_dwordx4 %2, %9, %14 sc1\n\t"
                   "global_load_dwordx4 %3, %10, %14 sc1\n\t"
                   "global_load_dwordx4 %4, %11, %14 sc1\n\t"
                   "global_load_dwordx4 %5, %12, %14 sc1\n\t"
                   "global_load_dwordx4 %6, %13, %14 sc1\n\t"
                   "s_waitcnt vmcnt(0)"
                   : "=&v"(gr[0]), "=&v"(gr[1]), "=&v"(gr[2]), "=&v"(gr[3]), "=&v"(gr[4]), "=&v"(gr[5]), "=&v"(gr[6])
                   : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "v"(off[6]), "s"(base)
                   : "memory");
   if (N == 8)
      asm volatile("s_nop 4\n\t"
                   "global_load_dwordx4 %0, %8, %16 sc1\n\t"
                   "global_load_dwordx4 %1, %9, %16 sc1\n\t"
                   "global_load_dwordx4 %2, %10, %16 sc1\n\t"
                   "global_load_dwordx4 %3, %11, %16 sc1\n\t"
                   "global_load_dwordx4 %4, %12, %16 sc1\n\t"
                   "global_load_dwordx4 %5, %13, %16 sc1\n\t"
                   "global_load_dwordx4 %6, %14, %16 sc1\n\t"
                   "global_load_dwordx4 %7, %15, %16 sc1\n\t"
                   "s_waitcnt vmcnt(0)"
                   : "=&v"(gr[0]), "=&v"(gr[1]), "=&v"(gr[2]), "=&v"(gr[3]), "=&v"(gr[4]), "=&v"(gr[5]), "=&v"(gr[6]), "=&v"(gr[7])
                   : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "v"(off[6]), "v"(off[7]), "s"(base)
                   : "memory");
}

// Between two sweeps of an exchange.  Inside a run of iterations the partials are there at the first or second sweep; a
// workgroup that reaches a locus before its partners (the job form: they may still be on their previous locus for
// milliseconds) must not keep hundreds of polling loads in flight meanwhile -- pollers take bandwidth from the workgroups
// that compute (MI355X_MICROARCH.md, polling-cost): after a few misses the pauses grow to ~3 us.
__device__ __forceinline__ void wide_poll_pause(unsigned spins)
{
   if (spins < 8) __builtin_amdgcn_s_sleep(1);
   else if (spins < 32) __builtin_amdgcn_s_sleep(16);
   else __builtin_amdgcn_s_sleep(127);
}

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

// calls fn(integral_constant<int, 0>) ... fn(integral_constant<int, N - 1>): a loop whose index is a constant
// expression in the body (block shapes differ; the register tile must be indexed by constants)
template <class Fn, int... B>
__device__ __forceinline__ void wide_for_each(Fn &fn, std::integer_sequence<int, B...>)
{
   (fn(std::integral_constant<int, B>{}), ...);
}

template <int LB_CL, int CPL, int R, int RL, int RBLK, int LBLK, bool BIAS>
__device__ __forceinline__ void em_wide_body(const WideArgs &g, const int di, const int w)
{
   constexpr int CL = 1 << LB_CL, GR = 64 / CL, NPAD = CL * CPL;
   constexpr int ROWSTEP = kWideWaves * GR; // rows of the block between a lane's consecutive row slots
   constexpr int RT = R + RL;               // a lane's row slots: the first R in registers, the others in LDS
   // the slots are worked on in blocks that share a v_rcp_f64: register blocks of RBLK rows, then LDS blocks of LBLK rows
   constexpr int NBR = (R + RBLK - 1) / RBLK, NBL = (RL + LBLK - 1) / LBLK, NBLK = NBR + NBL;
   static_assert((RBLK == 4 || RBLK == 2 || RBLK == 1) && (LBLK == 2 || LBLK == 1) && NPAD <= kWideThreads && NBLK <= 32, "layout");
   extern __shared__ double s_dyn[];
   double *phi = s_dyn;                       // [NPAD] theta_j * scale_j of the iteration about to run
   double *accw = phi + NPAD;                 // [kWideWaves][NPAD] the waves' partial column sums
   double *s_part = accw + kWideWaves * NPAD; // [kWideWaves] the owner waves' shares of ||next - theta||^2
   double *s_misc = s_part + kWideWaves;      // 0: total count, 1: rows kept, 3: abort, 4 + parity: zero-denominator flag of an iteration
   int *s_zf = (int *)(s_misc + 8);           // [kWideWaves] "a kept row of this wave had a zero denominator"
   double *s_row = (double *)(s_zf + 16);     // [RT * ROWSTEP] per row of the block: its count, -1 for a row outside the problem
   u32x4 *stage = (u32x4 *)(s_row + RT * ROWSTEP); // [2 NPAD] granules of the two-level exchange
   double *s_F = (double *)(stage + 2 * NPAD);     // [RL][CPL][kWideThreads] the lanes' rows beyond the register tile
   const EmArgs &a = g.a;
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   set_fp64_flush_denormals();
   const WideDesc d = g.table[di];
   const int G = d.n_blocks;
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
   const int row_slot0 = wave * GR + b; // slot r of this lane is row r * ROWSTEP + row_slot0 of the block

   // ---- my rows: block w owns rows [w * rows_per_block, ...); the lane's slot r is row r * ROWSTEP + wave * GR + b of it
   const int row_lo = w * d.rows_per_block;
   const int row_hi = min(nrow, row_lo + d.rows_per_block);
   double F[R][CPL];
   double csum[CPL];
#pragma unroll
   for (int k = 0; k < CPL; ++k) csum[k] = 0.0;
   double tot = 0.0;
   int kept = 0;
   unsigned blk_live = 0;  // bit per block: some lane of this wave has a kept row in it
   unsigned blk_mixed = 0; // bit per block: some lane's row is kept and some lane's is not
   // (the LDS slots first: their values pass through registers the tile does not occupy yet)
#pragma unroll
   for (int rr = 0; rr < RT; ++rr) {
      const int r = rr < RL ? R + rr : rr - RL;
      if (rr == RL) __builtin_amdgcn_sched_barrier(0);
      const int i = row_lo + r * ROWSTEP + row_slot0;
      const bool valid = i < row_hi;
      const int ic = valid ? i : row_lo; // (row_lo < nrow: every block owns at least one row)
      int cnt = a.count[r0 + ic];
      cnt = valid ? cnt : 0;
      double x[CPL];
      double mx = 0.0;
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
         const int j = k * CL + c; // the lane's k-th column: consecutive column lanes, consecutive columns (LDS banks)
         const bool ok = valid && j < niso;
         double v = Fg[(int64_t)ic * niso + (j < niso ? j : 0)]; // unconditional load, then select: no branch per element
         // config 5: the biased weight F_ij * 2^(row_bias_i * iso_bias_j) is the operand from here on, as in the tile kernels
         // (em_device.h).  An instantiation of its own: with the factors as a run-time branch of the one kernel the plain
         // path lost 3 % (C3-T 38.2 -> 39.3 ms: two more pointers across the unrolled load)
         if (BIAS) v *= bias_factor(a.row_bias[r0 + ic] * a.iso_bias[iso_base + (j < niso ? j : 0)]);
         v = ok ? v : 0.0;
         x[k] = v;
         mx = fmax(mx, v);
      }
      mx = wide_col_lanes_max<LB_CL>(mx);
      const bool keep = valid && mx > kRowEps; // estimate.cpp:380
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
         const double v = keep ? x[k] : 0.0;
         if (r < R) F[r < R ? r : 0][k] = v;
         else s_F[((r - R) * CPL + k) * kWideThreads + tid] = v;
         csum[k] += v;
      }
      if (c == 0) {
         s_row[r * ROWSTEP + row_slot0] = keep ? (double)cnt : -1.0;
         tot += (double)cnt; // theta_0 counts ALL rows (:374-375); one lane per row
      }
      kept |= keep ? 1 : 0;
      const int blk = r < R ? r / RBLK : NBR + (r - R) / LBLK;
      if (wave_any(keep)) blk_live |= 1u << blk;
      if (wave_any(!keep)) blk_mixed |= 1u << blk;
   }
   blk_live = (unsigned)__builtin_amdgcn_readfirstlane((int)blk_live);
   blk_mixed = (unsigned)__builtin_amdgcn_readfirstlane((int)blk_mixed); // (block 0 runs whether it is live or not)

   // The exchange: the thread that owns item `tid` (tid < n_items) hands in its workgroup's partial `s` and the
   // workgroup's flag bit and gets the sum of the G workgroups' partials, added in workgroup order (identical in
   // every workgroup), and the OR of their flags.  Called by ALL threads.  G == 1: nothing travels.
   //   up to kWideSweep workgroups: the owner reads the G partials of its item itself, all in flight together;
   //   more: two levels -- G x G x npad granules per exchange would be tens of MB per iteration chip-wide -- workgroup
   //   w sums the columns [w << lb_slice, ...) (one granule per thread, staged in LDS, a workgroup barrier, added in
   //   workgroup order) and publishes the totals, then every owner reads the ONE total of its item.
   bool aborted = false;
   // one granule, polled until it carries this exchange's tag
   auto fetch = [&](const char *p, u32x4 &gr) {
      for (unsigned spins = 0;; ++spins) {
         asm volatile("s_nop 4\n\t"
                   "global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(gr) : "v"(p) : "memory");
         if (granule_ready(gr, round)) break;
         if (spins > kWideSpinLimit || ((spins & 63) == 63 && __hip_atomic_load(g_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            aborted = true;
            break;
         }
         wide_poll_pause(spins);
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
            // granule k of my column: buf1 + (k * NPAD + tid) * 16 -- a 32-bit offset per granule on a scalar base
            unsigned off[kWideSweep];
#pragma unroll
            for (int k = 0; k < kWideSweep; ++k) off[k] = (unsigned)(((k < G ? k : G - 1) * NPAD + tid) * 16);
            double sum = 0.0;
            unsigned fl = 0;
            for (unsigned spins = 0;; ++spins) {
               // every sweep fetches all G granules (a second sweep is the exception): nothing of a sweep outlives it
               u32x4 gr[kWideSweep];
               switch (G) {
               case 2: granule_sweep<2>(buf1, off, gr); break;
               case 3: granule_sweep<3>(buf1, off, gr); break;
               case 4: granule_sweep<4>(buf1, off, gr); break;
               case 5: granule_sweep<5>(buf1, off, gr); break;
               case 6: granule_sweep<6>(buf1, off, gr); break;
               case 7: granule_sweep<7>(buf1, off, gr); break;
               default: granule_sweep<8>(buf1, off, gr); break;
               }
               bool all = true;
#pragma unroll
               for (int k = 0; k < kWideSweep; ++k)
                  if (k < G) all = all && granule_ready(gr[k], round);
               if (all) {
#pragma unroll
                  for (int k = 0; k < kWideSweep; ++k)
                     if (k < G) {
                        sum += granule_value(gr[k]);
                        fl |= granule_flag(gr[k], round);
                     }
                  break;
               }
               // give up: after the spin limit, or when another workgroup has (the error word lives in host memory)
               if (spins > kWideSpinLimit || ((spins & 63) == 63 && __hip_atomic_load(g_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                  aborted = true;
                  break;
               }
               wide_poll_pause(spins);
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
      auto block = [&](auto tag) {
         constexpr int BLK = decltype(tag)::value;
         constexpr bool IN_LDS = BLK >= NBR;
         constexpr int S0 = IN_LDS ? R + (BLK - NBR) * LBLK : BLK * RBLK;                       // the block's first row slot
         constexpr int NB = IN_LDS ? (RL - (BLK - NBR) * LBLK < LBLK ? RL - (BLK - NBR) * LBLK : LBLK)
                                   : (R - BLK * RBLK < RBLK ? R - BLK * RBLK : RBLK);           // its rows
         if (BLK > 0 && !((blk_live >> BLK) & 1u)) return; // no lane of this wave has a kept row in these slots: nothing to add
         double cnt[NB];
#pragma unroll
         for (int q = 0; q < NB; ++q) cnt[q] = s_row[(S0 + q) * ROWSTEP + row_slot0];
         // (an LDS block's reads stay behind the block before it: hoisted to the top of the pass -- nothing they depend
         // on -- they would all be live at once, RL x CPL doubles the tile has no registers for)
         if (IN_LDS) __builtin_amdgcn_sched_barrier(0);
         double fb[NB][CPL]; // the block's weights: the register tile itself, or the lane's LDS slots read once for both steps
#pragma unroll
         for (int q = 0; q < NB; ++q) {
#pragma unroll
            for (int k = 0; k < CPL; ++k) fb[q][k] = IN_LDS ? s_F[((S0 + q - R) * CPL + k) * kWideThreads + tid] : F[IN_LDS ? 0 : S0 + q][k];
         }
         double dd[4];
#pragma unroll
         for (int q = 0; q < NB; ++q) {
            double part = 0.0;
#pragma unroll
            for (int k = 0; k < CPL; ++k) part = __builtin_fma(fb[q][k], ph[k], part); // :450
            dd[q] = part;
         }
         wide_col_lanes_sum<LB_CL, NB>(dd);
         // The matrix instruction's result wants six wait states before a vector instruction reads it.  The compiler
         // counts them along the fall-through path only: with a branch between the two (the mixed-block test below) the
         // taken path's first instruction read the sums early -- NaN in every theta of one layout.  Eight wait states
         // tied to the sums, so that nothing that reads them can come before
         if (NB == 4) asm volatile("s_nop 7" : "+v"(dd[0]), "+v"(dd[1]), "+v"(dd[2]), "+v"(dd[3]));
         else if (NB == 3) asm volatile("s_nop 7" : "+v"(dd[0]), "+v"(dd[1]), "+v"(dd[2]));
         else if (NB == 2) asm volatile("s_nop 7" : "+v"(dd[0]), "+v"(dd[1]));
         else asm volatile("s_nop 7" : "+v"(dd[0]));
         // a row outside the problem (dropped by init(), or beyond the block's rows) has weights 0 and "count" -1: its
         // denominator is 0 + 1 and its weight 0 / 1.  Only a block with such a row in some lane looks (a wave-uniform
         // bit per block): elsewhere the denominators are the sums and the counts the counts
         double de[4], inv[4];
         if ((blk_mixed >> BLK) & 1u) {
#pragma unroll
            for (int q = 0; q < NB; ++q) {
               const bool out = cnt[q] < 0.0;
               de[q] = out ? 1.0 : dd[q];
               cnt[q] = out ? 0.0 : cnt[q];
            }
         } else {
#pragma unroll
            for (int q = 0; q < NB; ++q) de[q] = dd[q];
         }
         double prod;
         if (NB == 4) prod = (de[0] * de[1]) * (de[2] * de[3]);
         else if (NB == 3) prod = (de[0] * de[1]) * de[2];
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
               zf |= de[q] == 0.0; // :451 (a kept row's: the others' denominators are 1)
               inv[q] = newton_rcp(de[q]);
            }
         }
#pragma unroll
         for (int q = 0; q < NB; ++q) {
            const double wgt = cnt[q] * inv[q];
            if (BLK == 0 && q == 0) { // the first row starts the column partials
#pragma unroll
               for (int k = 0; k < CPL; ++k) acc[k] = wgt * fb[0][k];
            } else {
#pragma unroll
               for (int k = 0; k < CPL; ++k) acc[k] = __builtin_fma(wgt, fb[q][k], acc[k]);
            }
         }
         // one block's temporaries at a time (the tile leaves few registers): the empty asm pins the column partials
         // here, so the block's updates cannot be sunk behind the later blocks' reciprocals
#pragma unroll
         for (int k = 0; k < CPL; ++k) asm volatile("" : "+v"(acc[k]));
      };
      wide_for_each(block, std::make_integer_sequence<int, NBLK>{});
      wide_row_lanes_sum<LB_CL>(acc);
      if (b == 0) {
#pragma unroll
         for (int k = 0; k < CPL; ++k) accw[wave * NPAD + k * CL + c] = acc[k];
      }
      {
         const int zw = wave_any(zf) ? 1 : 0;
         if (lane == 0) s_zf[wave] = zw;
      }
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
         if (s_misc[4 + ((it - 1) & 1)] != 0.0) { // run() == false, _theta untouched (:451-453)
            st = kStDenomZero;
            theta0_out = true;
            it_done = it - 1;
            break;
         }
         if (dsum <= kThetaLimitSq) { // sqrt(d2) < 1e-2 (:479-480), theta NOT updated: the value of before
            st = kStOk;
            prev_out = true;
            it_done = it - 1;
            break;
         }
      }
      // ---- the owner of column tid: the locus-wide sum, next_theta (:454-464), its share of ||next - theta||^2
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
// One launch serves loci of all widths (a round's workgroups must all be resident together, and cooperative
// launches do not overlap, so rounds should be few and full): the workgroup looks up its locus and jumps to the
// instantiation its width needs.  (Round 5 measured two forms of ONE launch for the whole tail -- every resident workgroup
// working through a list of (locus, part) jobs, scheduled on the host or taken from a queue on the device -- against the
// rounds: 41.6 and 39.9 ms against 39.0 on C3-T.  A locus starts when the LAST of its workgroups is free, and rounds of
// loci of like workgroup counts, all of which run the same 1000 iterations, keep the workgroups aligned; a queue lets them
// drift apart.  profiles/EXPERIMENTS_r05.md.)
#ifdef SB_WIDE_ONLY // diagnostic: one instantiation per build (register report)
#define SB_WIDE_CASE(ID) \
   case ID: if (ID == SB_WIDE_ONLY) em_wide_body<wide_layout(ID).lb_cl, wide_layout(ID).cpl, wide_layout(ID).r, wide_layout(ID).rl, wide_layout(ID).rblk, wide_layout(ID).lblk, BIAS>(g, di, w); break;
#else
#define SB_WIDE_CASE(ID) \
   case ID: em_wide_body<wide_layout(ID).lb_cl, wide_layout(ID).cpl, wide_layout(ID).r, wide_layout(ID).rl, wide_layout(ID).rblk, wide_layout(ID).lblk, BIAS>(g, di, w); break;
#endif
template <bool BIAS>
__global__ __launch_bounds__(kWideThreads) void em_wide_kernel(WideArgs g)
{
   int di = 0;
   for (int k = 1; k < g.n_desc; ++k)
      if (g.table[k].first_block <= (int)blockIdx.x) di = k;
   di = __builtin_amdgcn_readfirstlane(di);
   const int w = (int)blockIdx.x - g.table[di].first_block;
   switch (g.table[di].layout) {
      SB_WIDE_CASE(0)
      SB_WIDE_CASE(1)
      SB_WIDE_CASE(2)
      SB_WIDE_CASE(3)
      SB_WIDE_CASE(4)
      SB_WIDE_CASE(5)
      SB_WIDE_CASE(6)
      SB_WIDE_CASE(7)
      SB_WIDE_CASE(8)
      SB_WIDE_CASE(9)
      SB_WIDE_CASE(10)
      SB_WIDE_CASE(11)
      SB_WIDE_CASE(12)
   }
}
#undef SB_WIDE_CASE
#endif

hipError_t launch_wide(const WideArgs &g, int n_blocks, size_t lds_bytes, hipStream_t s);

} // namespace sb
