// tools/probe_mfma_layout.hip -- which lanes does v_mfma_f64_4x4x4_4b sum over?  Every lane contributes 2^lane, so a
// result read as an integer is the set of lanes it sums.  Build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/probe_mfma tools/probe_mfma_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(double *out)
{
   const int l = threadIdx.x;
   const double x = (double)(1ull << l);
   // (0) data in B, ones in A
   out[0 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, x, 0.0, 0, 0, 0);
   // (1) data in A, ones in B
   const double t = __builtin_amdgcn_mfma_f64_4x4x4f64(x, 1.0, 0.0, 0, 0, 0);
   out[1 * 64 + l] = t;
   // (2) the pair: data in A / ones in B, then ones in A / result in B
   out[2 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, t, 0.0, 0, 0, 0);
   // (3) data in B, A = 1 where bit 1 of (lane % 4) equals bit 1 of (lane / 16): sums over lane bit 4 only?
   const double m4 = (((l & 3) >> 1) == ((l >> 4) >> 1)) ? 1.0 : 0.0;
   out[3 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(m4, x, 0.0, 0, 0, 0);
   // (4) same with bit 0: sums over lane bit 5 only?
   const double m5 = (((l & 3) & 1) == ((l >> 4) & 1)) ? 1.0 : 0.0;
   out[4 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(m5, x, 0.0, 0, 0, 0);
}
int main()
{
   double *d;
   hipMalloc(&d, 5 * 64 * sizeof(double));
   hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
   double h[5 * 64];
   hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
   const char *names[5] = {"A=1, B=x", "A=x, B=1", "pair", "A=mask(bit 4 only), B=x", "A=mask(bit 5 only), B=x"};
   for (int t = 0; t < 5; ++t) {
      printf("%s\n", names[t]);
      for (int l = 0; l < 64; ++l) {
         const uint64_t m = (uint64_t)h[t * 64 + l];
         printf("  lane %2d <-", l);
         for (int s = 0; s < 64; ++s) if (m >> s & 1) printf(" %d", s);
         printf("\n");
      }
   }
   return 0;
}
