// tools/microbench.hip -- instruction-cost probes behind the EM kernels' design (gfx950).
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/microbench tools/microbench.hip && /tmp/microbench
//
// What a lone wave and 2 / 4 co-resident waves per SIMD pay per instruction for the few
// instruction kinds an EM iteration is made of (fp64 FMA, v_rcp_f64, DPP moves, the
// lane-swap forms, ds_swizzle, fp64 MFMA), plus the accuracy of v_rcp_f64 and of the two
// division forms built on it.  Diagnostic only: nothing here is linked into libsbgpu.so.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CHECK(x)                                                                          \
   do {                                                                                   \
      hipError_t e_ = (x);                                                                \
      if (e_ != hipSuccess) {                                                             \
         std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
         std::exit(1);                                                                    \
      }                                                                                   \
   } while (0)

enum Test {
   kFmaDep,      // one dependent chain of v_fma_f64
   kFmaInd,      // 8 independent chains
   kAddInd,      // v_add_f64, 8 chains
   kMulInd,      // v_mul_f64, 8 chains
   kRcpInd,      // v_rcp_f64, 8 independent
   kFma32Ind,    // v_fma_f32, 8 chains
   kDppInd,      // v_mov_b32 dpp quad_perm, 8 independent
   kDppDep,      // dependent chain of the same
   kButterfly,   // x += xor1(x) as 2 dpp movs + v_add_f64, dependent
   kButterfly4,  // 4 independent butterflies interleaved
   kSwap32,      // v_permlane32_swap_b32, dependent
   kSwap16,      // v_permlane16_swap_b32, dependent
   kSwizzle,     // ds_swizzle_b32 + wait, dependent
   kCndmask,     // v_cndmask_b32 independent
   kCmp,         // v_cmp_eq_f64 independent (to sgpr pairs)
   kMfma16Dep,   // v_mfma_f64_16x16x4_f64, dependent accumulator
   kMfma16Ind,   // 4 independent accumulators
   kMfma4Dep,    // v_mfma_f64_4x4x4_4b_f64 dependent
   kMfma4Ind,    // 4 independent
   kDivFull,     // rcp + 2 Newton + mul (fast_div of em_device.h), 4 independent
   kDivShort,    // rcp + mul + residual fma + fma, 4 independent
   kLdsRound,    // block-form exchange: ds_write_b64, barrier, 4 x ds_read_b64, adds (256 threads)
   kCndmask64,   // v_cndmask_b32_e64 v, 0, v, s[a:b] (what `x = p ? x : 0` compiles to), 8 independent
   kMovInd,      // v_mov_b32, 8 independent
   kMaxInd,      // v_max_f64, 8 independent
   kRowSelect,   // one row of the E-step as compiled today: 4 fma, rcp, 2 Newton, mul, 2 cndmask, 4 fma
   kRowAdd,      // the same with the select replaced by d + inactive (one v_add_f64)
   kNumTests
};
static const char *kNames[kNumTests] = {
   "v_fma_f64 dependent chain", "v_fma_f64 8 independent", "v_add_f64 8 independent", "v_mul_f64 8 independent",
   "v_rcp_f64 8 independent", "v_fma_f32 8 independent", "v_mov_b32 dpp 8 independent", "v_mov_b32 dpp dependent",
   "butterfly step (2 dpp + v_add_f64) dependent", "butterfly step x4 interleaved (per step)",
   "v_permlane32_swap dependent", "v_permlane16_swap dependent", "ds_swizzle + wait dependent",
   "v_cndmask_b32 8 independent", "v_cmp_eq_f64 8 independent", "v_mfma_f64_16x16x4 dependent",
   "v_mfma_f64_16x16x4 4 independent", "v_mfma_f64_4x4x4_4b dependent", "v_mfma_f64_4x4x4_4b 4 independent",
   "division rcp+2 Newton+mul (6 instr) x4 (per division)", "division rcp+mul+2 fma (4 instr) x4 (per division)",
   "LDS exchange round, 4 waves (per round)", "v_cndmask_b32_e64 v,0,v,s[] 8 independent", "v_mov_b32 8 independent",
   "v_max_f64 8 independent", "E-step row with select (per row)", "E-step row with add (per row)"};

typedef double d4 __attribute__((ext_vector_type(4)));

template <int T>
__global__ __launch_bounds__(256) void probe(unsigned long long *out, double *sink, int iters)
{
   double a = 1.0000001 + threadIdx.x * 1e-9, b = 0.9999999, c = 1e-9;
   double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
   int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
   float f0 = (float)a, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
   float fb = 0.999f, fc = 1e-3f;
   d4 m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, m3 = m0;
   __shared__ double lds[2][4][64];
   const unsigned long long t0 = __builtin_amdgcn_s_memtime();
   for (int it = 0; it < iters; ++it) {
      if (T == kFmaDep) {
#pragma unroll
         for (int u = 0; u < 16; ++u) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x0) : "v"(b), "v"(c));
      } else if (T == kFmaInd || T == kAddInd || T == kMulInd) {
#pragma unroll
         for (int u = 0; u < 2; ++u) {
#define OP8(INS)                                                                                         \
   asm volatile(INS " %0, %0, %8, %9\n" INS " %1, %1, %8, %9\n" INS " %2, %2, %8, %9\n" INS " %3, %3, %8, %9\n" \
                INS " %4, %4, %8, %9\n" INS " %5, %5, %8, %9\n" INS " %6, %6, %8, %9\n" INS " %7, %7, %8, %9"  \
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)            \
                : "v"(b), "v"(c))
            if (T == kFmaInd) OP8("v_fma_f64");
#undef OP8
#define OP8B(INS)                                                                                        \
   asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n"          \
                INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8"            \
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)        \
                : "v"(b))
            if (T == kAddInd) OP8B("v_add_f64");
            if (T == kMulInd) OP8B("v_mul_f64");
#undef OP8B
         }
      } else if (T == kRcpInd) {
#pragma unroll
         for (int u = 0; u < 2; ++u)
            asm volatile("v_rcp_f64 %0, %8\nv_rcp_f64 %1, %8\nv_rcp_f64 %2, %8\nv_rcp_f64 %3, %8\n"
                         "v_rcp_f64 %4, %8\nv_rcp_f64 %5, %8\nv_rcp_f64 %6, %8\nv_rcp_f64 %7, %8"
                         : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3), "=v"(x4), "=v"(x5), "=v"(x6), "=v"(x7)
                         : "v"(a));
      } else if (T == kFma32Ind) {
#pragma unroll
         for (int u = 0; u < 2; ++u)
            asm volatile("v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\n"
                         "v_fma_f32 %3, %3, %8, %9\nv_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\n"
                         "v_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)
                         : "v"(fb), "v"(fc));
      } else if (T == kDppInd) {
#pragma unroll
         for (int u = 0; u < 2; ++u)
            asm volatile("v_mov_b32_dpp %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %3, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %4, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %5, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %6, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %7, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                         : "=v"(i0), "=v"(i1), "=v"(i2), "=v"(i3), "=v"(i4), "=v"(i5), "=v"(i6), "=v"(i7)
                         : "v"((int)threadIdx.x));
      } else if (T == kDppDep) {
#pragma unroll
         for (int u = 0; u < 16; ++u) {
            int t;
            asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(i0));
            i0 = t;
         }
      } else if (T == kButterfly) {
#pragma unroll
         for (int u = 0; u < 16; ++u) {
            int lo = __double2loint(x0), hi = __double2hiint(x0);
            lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true);
            hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
            double y = __hiloint2double(hi, lo);
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(x0) : "v"(y));
         }
      } else if (T == kButterfly4) {
#pragma unroll
         for (int u = 0; u < 4; ++u) {
            double *xs[4] = {&x0, &x1, &x2, &x3};
            double ys[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
               int lo = __double2loint(*xs[q]), hi = __double2hiint(*xs[q]);
               lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true);
               hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
               ys[q] = __hiloint2double(hi, lo);
            }
            asm volatile("v_add_f64 %0, %0, %4\nv_add_f64 %1, %1, %5\nv_add_f64 %2, %2, %6\nv_add_f64 %3, %3, %7"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)
                         : "v"(ys[0]), "v"(ys[1]), "v"(ys[2]), "v"(ys[3]));
         }
      } else if (T == kSwap32 || T == kSwap16) {
#pragma unroll
         for (int u = 0; u < 16; ++u) {
            if (T == kSwap32) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(i0), "+v"(i1));
            else asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(i0), "+v"(i1));
         }
      } else if (T == kSwizzle) {
#pragma unroll
         for (int u = 0; u < 16; ++u) {
            i0 = __builtin_amdgcn_ds_swizzle(i0, 0x401F);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
         }
      } else if (T == kCndmask) {
#pragma unroll
         for (int u = 0; u < 2; ++u)
            asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\n"
                         "v_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\n"
                         "v_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7)
                         : "v"((int)threadIdx.x)
                         : "vcc");
      } else if (T == kCmp) {
#pragma unroll
         for (int u = 0; u < 2; ++u)
            asm volatile("v_cmp_eq_f64 s[40:41], %0, %1\nv_cmp_eq_f64 s[42:43], %0, %1\nv_cmp_eq_f64 s[44:45], %0, %1\n"
                         "v_cmp_eq_f64 s[46:47], %0, %1\nv_cmp_eq_f64 s[48:49], %0, %1\nv_cmp_eq_f64 s[50:51], %0, %1\n"
                         "v_cmp_eq_f64 s[52:53], %0, %1\nv_cmp_eq_f64 s[54:55], %0, %1"
                         :
                         : "v"(x0), "v"(x1)
                         : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53",
                           "s54", "s55");
      } else if (T == kMfma16Dep) {
#pragma unroll
         for (int u = 0; u < 8; ++u) m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, m0, 0, 0, 0);
      } else if (T == kMfma16Ind) {
#pragma unroll
         for (int u = 0; u < 2; ++u) {
            m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, m0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, m1, 0, 0, 0);
            m2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, m2, 0, 0, 0);
            m3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, m3, 0, 0, 0);
         }
      } else if (T == kMfma4Dep) {
#pragma unroll
         for (int u = 0; u < 8; ++u) x0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, x0, 0, 0, 0);
      } else if (T == kMfma4Ind) {
#pragma unroll
         for (int u = 0; u < 2; ++u) {
            x0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, x1, 0, 0, 0);
            x2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, x2, 0, 0, 0);
            x3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, x3, 0, 0, 0);
         }
      } else if (T == kDivFull || T == kDivShort) {
         double *xs[4] = {&x0, &x1, &x2, &x3};
#pragma unroll
         for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
               const double d = *xs[q], n = a;
               double r = __builtin_amdgcn_rcp(d);
               if (T == kDivFull) {
                  double e = __builtin_fma(-d, r, 1.0);
                  r = __builtin_fma(r, e, r);
                  e = __builtin_fma(-d, r, 1.0);
                  r = __builtin_fma(r, e, r);
                  *xs[q] = n * r + 1.5;
               } else {
                  const double w = n * r;
                  const double e = __builtin_fma(-d, w, n);
                  *xs[q] = __builtin_fma(e, r, w) + 1.5;
               }
            }
         }
      } else if (T == kCndmask64) {
         const unsigned long long msk = 0x5555555555555555ull ^ (unsigned long long)iters;
#pragma unroll
         for (int u = 0; u < 2; ++u)
            asm volatile("v_cndmask_b32_e64 %0, 0, %0, %8\nv_cndmask_b32_e64 %1, 0, %1, %8\nv_cndmask_b32_e64 %2, 0, %2, %8\n"
                         "v_cndmask_b32_e64 %3, 0, %3, %8\nv_cndmask_b32_e64 %4, 0, %4, %8\nv_cndmask_b32_e64 %5, 0, %5, %8\n"
                         "v_cndmask_b32_e64 %6, 0, %6, %8\nv_cndmask_b32_e64 %7, 0, %7, %8"
                         : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7)
                         : "s"(msk));
      } else if (T == kMovInd) {
#pragma unroll
         for (int u = 0; u < 2; ++u)
            asm volatile("v_mov_b32 %0, %8\nv_mov_b32 %1, %8\nv_mov_b32 %2, %8\nv_mov_b32 %3, %8\n"
                         "v_mov_b32 %4, %8\nv_mov_b32 %5, %8\nv_mov_b32 %6, %8\nv_mov_b32 %7, %8"
                         : "=v"(i0), "=v"(i1), "=v"(i2), "=v"(i3), "=v"(i4), "=v"(i5), "=v"(i6), "=v"(i7)
                         : "v"((int)threadIdx.x));
      } else if (T == kMaxInd) {
#pragma unroll
         for (int u = 0; u < 2; ++u)
            asm volatile("v_max_f64 %0, %0, %8\nv_max_f64 %1, %1, %8\nv_max_f64 %2, %2, %8\nv_max_f64 %3, %3, %8\n"
                         "v_max_f64 %4, %4, %8\nv_max_f64 %5, %5, %8\nv_max_f64 %6, %6, %8\nv_max_f64 %7, %7, %8"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                         : "v"(b));
      } else if (T == kRowSelect || T == kRowAdd) {
         // 8 rows of a 4-column tile: d = F . phi; w = n / d (masked); acc += w F
         const double F0 = a, F1 = b, F2 = c + 1, F3 = a * b;
         const bool act = (threadIdx.x & 1) == 0 || iters > 5;
         const double ina = act ? 0.0 : 1.0;
         double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
#pragma unroll
         for (int r = 0; r < 8; ++r) {
            double d = F0 * x0;
            d = __builtin_fma(F1, x1, d);
            d = __builtin_fma(F2, x2, d);
            d = __builtin_fma(F3, x3, d);
            if (T == kRowAdd) d += ina;
            double rr = __builtin_amdgcn_rcp(d);
            double e = __builtin_fma(-d, rr, 1.0);
            rr = __builtin_fma(rr, e, rr);
            e = __builtin_fma(-d, rr, 1.0);
            rr = __builtin_fma(rr, e, rr);
            double w = x4 * rr;
            if (T == kRowSelect) w = act ? w : 0.0;
            acc0 = __builtin_fma(w, F0, acc0);
            acc1 = __builtin_fma(w, F1, acc1);
            acc2 = __builtin_fma(w, F2, acc2);
            acc3 = __builtin_fma(w, F3, acc3);
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)); // rows stay separate computations
            x0 += 1e-9 * r;
         }
         x0 = acc0 * 1e-3 + 1.0;
         x1 = acc1 * 1e-3 + 1.0;
         x2 = acc2 * 1e-3 + 1.0;
         x3 = acc3 * 1e-3 + 1.0;
      } else if (T == kLdsRound) {
         const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
         lds[it & 1][wave][lane] = x0;
         __syncthreads();
         double s = lds[it & 1][0][lane];
         s += lds[it & 1][1][lane];
         s += lds[it & 1][2][lane];
         s += lds[it & 1][3][lane];
         x0 = s * 0.25;
      }
   }
   const unsigned long long t1 = __builtin_amdgcn_s_memtime();
   const int wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
   if ((threadIdx.x & 63) == 0) out[wave_id] = t1 - t0;
   // keep every value alive
   double s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (double)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7) +
              (double)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7) + m0[0] + m0[1] + m0[2] + m0[3] + m1[0] + m2[0] + m3[0];
   if (s == 123.456) sink[0] = s;
}

static int instr_per_iter(int t)
{
   switch (t) {
   case kFmaDep: return 16;
   case kFmaInd: case kAddInd: case kMulInd: case kRcpInd: case kFma32Ind: case kDppInd: case kCndmask: case kCmp: return 16;
   case kDppDep: return 16;
   case kButterfly: return 16;
   case kButterfly4: return 16;
   case kSwap32: case kSwap16: case kSwizzle: return 16;
   case kMfma16Dep: case kMfma16Ind: case kMfma4Dep: case kMfma4Ind: return 8;
   case kDivFull: case kDivShort: return 8;
   case kLdsRound: return 1;
   case kCndmask64: case kMovInd: case kMaxInd: return 16;
   case kRowSelect: case kRowAdd: return 8;
   }
   return 1;
}

template <int T>
static void run(int waves_per_simd, unsigned long long *d_out, double *d_sink, int n_cu)
{
   const int threads = (T == kLdsRound) ? 256 : 64;
   const int waves = n_cu * 4 * waves_per_simd;
   const int blocks = waves * 64 / threads;
   const int iters = 2000;
   hipLaunchKernelGGL((probe<T>), dim3(blocks), dim3(threads), 0, 0, d_out, d_sink, iters); // warm-up
   CHECK(hipDeviceSynchronize());
   hipLaunchKernelGGL((probe<T>), dim3(blocks), dim3(threads), 0, 0, d_out, d_sink, iters);
   CHECK(hipDeviceSynchronize());
   std::vector<unsigned long long> h(waves);
   CHECK(hipMemcpy(h.data(), d_out, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
   std::sort(h.begin(), h.end());
   const double per = (double)instr_per_iter(T) * iters;
   std::printf("%-58s waves/SIMD %d: median %7.2f  min %7.2f  max %7.2f cycles\n", kNames[T], waves_per_simd,
               h[waves / 2] / per, h[0] / per, h[waves - 1] / per);
}

template <int T>
static void run_all(unsigned long long *d_out, double *d_sink, int n_cu)
{
   for (int w : {1, 2, 4}) {
      if (T == kLdsRound && w == 4) continue;
      run<T>(w, d_out, d_sink, n_cu);
   }
}

// ---- accuracy of v_rcp_f64 and the division forms
__global__ void div_accuracy(const double *n, const double *d, double *rcp, double *full, double *shrt, int count)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= count) return;
   const double dd = d[i], nn = n[i];
   double r = __builtin_amdgcn_rcp(dd);
   rcp[i] = r;
   {
      double e = __builtin_fma(-dd, r, 1.0);
      double r1 = __builtin_fma(r, e, r);
      e = __builtin_fma(-dd, r1, 1.0);
      r1 = __builtin_fma(r1, e, r1);
      full[i] = nn * r1;
   }
   {
      const double w = nn * r;
      const double e = __builtin_fma(-dd, w, nn);
      shrt[i] = __builtin_fma(e, r, w);
   }
}

// which lanes a v_mfma_f64_4x4x4_4b_f64 sums: A = 1, B = lane id  ->  D per lane
__global__ void mfma_layout(double *out)
{
   const double lane = (double)threadIdx.x;
   out[threadIdx.x] = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, lane, 0.0, 0, 0, 0);
   out[64 + threadIdx.x] = __builtin_amdgcn_mfma_f64_4x4x4f64(lane, 1.0, 0.0, 0, 0, 0);
}
static void mfma_layout_probe()
{
   double *d;
   CHECK(hipMalloc(&d, 128 * sizeof(double)));
   hipLaunchKernelGGL(mfma_layout, dim3(1), dim3(64), 0, 0, d);
   double h[128];
   CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
   std::printf("v_mfma_f64_4x4x4_4b: D(lane) for A = 1, B = lane id:\n");
   for (int i = 0; i < 64; ++i) std::printf("%4.0f%s", h[i], (i & 15) == 15 ? "\n" : "");
   std::printf("v_mfma_f64_4x4x4_4b: D(lane) for A = lane id, B = 1:\n");
   for (int i = 0; i < 64; ++i) std::printf("%4.0f%s", h[64 + i], (i & 15) == 15 ? "\n" : "");
}

int main()
{
   hipDeviceProp_t prop;
   CHECK(hipGetDeviceProperties(&prop, 0));
   const int n_cu = prop.multiProcessorCount;
   std::printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, n_cu, prop.clockRate);
   unsigned long long *d_out;
   double *d_sink;
   CHECK(hipMalloc(&d_out, sizeof(unsigned long long) * n_cu * 4 * 8));
   CHECK(hipMalloc(&d_sink, 64));
   run_all<kFmaDep>(d_out, d_sink, n_cu);
   run_all<kFmaInd>(d_out, d_sink, n_cu);
   run_all<kAddInd>(d_out, d_sink, n_cu);
   run_all<kMulInd>(d_out, d_sink, n_cu);
   run_all<kRcpInd>(d_out, d_sink, n_cu);
   run_all<kFma32Ind>(d_out, d_sink, n_cu);
   run_all<kDppInd>(d_out, d_sink, n_cu);
   run_all<kDppDep>(d_out, d_sink, n_cu);
   run_all<kButterfly>(d_out, d_sink, n_cu);
   run_all<kButterfly4>(d_out, d_sink, n_cu);
   run_all<kSwap32>(d_out, d_sink, n_cu);
   run_all<kSwap16>(d_out, d_sink, n_cu);
   run_all<kSwizzle>(d_out, d_sink, n_cu);
   run_all<kCndmask>(d_out, d_sink, n_cu);
   run_all<kCmp>(d_out, d_sink, n_cu);
   run_all<kMfma16Dep>(d_out, d_sink, n_cu);
   run_all<kMfma16Ind>(d_out, d_sink, n_cu);
   run_all<kMfma4Dep>(d_out, d_sink, n_cu);
   run_all<kMfma4Ind>(d_out, d_sink, n_cu);
   run_all<kDivFull>(d_out, d_sink, n_cu);
   run_all<kDivShort>(d_out, d_sink, n_cu);
   run_all<kLdsRound>(d_out, d_sink, n_cu);
   run_all<kCndmask64>(d_out, d_sink, n_cu);
   run_all<kMovInd>(d_out, d_sink, n_cu);
   run_all<kMaxInd>(d_out, d_sink, n_cu);
   run_all<kRowSelect>(d_out, d_sink, n_cu);
   run_all<kRowAdd>(d_out, d_sink, n_cu);
   mfma_layout_probe();

   // accuracy
   const int N = 1 << 20;
   std::mt19937_64 rng(7);
   std::vector<double> hn(N), hd(N);
   for (int i = 0; i < N; ++i) {
      const double e1 = std::uniform_real_distribution<double>(-20, 20)(rng), e2 = std::uniform_real_distribution<double>(-20, 20)(rng);
      hn[i] = std::uniform_real_distribution<double>(1, 2)(rng) * std::pow(2.0, e1);
      hd[i] = std::uniform_real_distribution<double>(1, 2)(rng) * std::pow(2.0, e2);
   }
   double *dn, *dd, *dr, *df, *ds;
   for (double **p : {&dn, &dd, &dr, &df, &ds}) CHECK(hipMalloc(p, N * sizeof(double)));
   CHECK(hipMemcpy(dn, hn.data(), N * sizeof(double), hipMemcpyHostToDevice));
   CHECK(hipMemcpy(dd, hd.data(), N * sizeof(double), hipMemcpyHostToDevice));
   hipLaunchKernelGGL(div_accuracy, dim3(N / 256), dim3(256), 0, 0, dn, dd, dr, df, ds, N);
   CHECK(hipDeviceSynchronize());
   std::vector<double> hr(N), hf(N), hs(N);
   CHECK(hipMemcpy(hr.data(), dr, N * sizeof(double), hipMemcpyDeviceToHost));
   CHECK(hipMemcpy(hf.data(), df, N * sizeof(double), hipMemcpyDeviceToHost));
   CHECK(hipMemcpy(hs.data(), ds, N * sizeof(double), hipMemcpyDeviceToHost));
   double er = 0, ef = 0, es = 0;
   long nf = 0, ns = 0;
   for (int i = 0; i < N; ++i) {
      const long double q = (long double)hn[i] / (long double)hd[i], rr = 1.0L / (long double)hd[i];
      er = std::max(er, (double)fabsl(((long double)hr[i] - rr) / rr));
      ef = std::max(ef, (double)fabsl(((long double)hf[i] - q) / q));
      es = std::max(es, (double)fabsl(((long double)hs[i] - q) / q));
      const double ieee = hn[i] / hd[i];
      nf += hf[i] != ieee;
      ns += hs[i] != ieee;
   }
   std::printf("v_rcp_f64 max relative error            %.3e (2^%.1f)\n", er, std::log2(er));
   std::printf("rcp + 2 Newton + mul   max rel error    %.3e, differs from IEEE n/d in %ld of %d\n", ef, nf, N);
   std::printf("rcp + mul + 2 fma      max rel error    %.3e, differs from IEEE n/d in %ld of %d\n", es, ns, N);
   return 0;
}
