// strawberry_amd/csrc/bitonic_big.h -- a bitonic sort of (key, index) pairs that live in GLOBAL memory, by one workgroup,
// for the clusters the LDS kernels of collapse_device.h / matepair_device.h do not hold.
//
// The plain network over global memory costs log2(n)(log2(n)+1)/2 passes through the caches with a barrier each (153 for
// 2^17 elements).  Every compare-exchange at distance j < CH stays inside an aligned chunk of CH elements, so all such
// passes of a stage run on a chunk that is loaded into LDS once: what is left in global memory are the passes at distance
// >= CH -- 15 for 2^17 elements and CH = 4096.  Ascending by (key, index); n2 a power of two.
#pragma once

#include <hip/hip_runtime.h>

namespace sb {

template <int THREADS, int CH>
__device__ inline void bitonic_sort_global(unsigned long long *key, int *idx, int n2, unsigned long long *lk /*[CH]*/, int *li /*[CH]*/)
{
   const int tid = threadIdx.x;
   const int ch = n2 < CH ? n2 : CH; // elements per chunk
   auto cx = [](unsigned long long &ka, int &ia, unsigned long long &kb, int &ib, bool up) {
      const bool greater = ka > kb || (ka == kb && ia > ib);
      if (greater == up) {
         const unsigned long long tk = ka;
         ka = kb, kb = tk;
         const int ti = ia;
         ia = ib, ib = ti;
      }
   };
   // the passes of stage k2 at distances j_hi, j_hi / 2, ... 1 (j_hi < ch), chunk by chunk in LDS
   auto chunks_in_lds = [&](int k2_first, int k2_last, bool all_distances) {
      for (int c0 = 0; c0 < n2; c0 += ch) {
         for (int t = tid; t < ch; t += THREADS) lk[t] = key[c0 + t], li[t] = idx[c0 + t];
         __syncthreads();
         for (int k2 = k2_first; k2 <= k2_last; k2 <<= 1)
            for (int j = (all_distances ? k2 : ch) >> 1; j > 0; j >>= 1) {
               for (int t = tid; t < ch; t += THREADS) {
                  const int p = t ^ j;
                  if (p > t) cx(lk[t], li[t], lk[p], li[p], ((c0 + t) & k2) == 0);
               }
               __syncthreads();
            }
         for (int t = tid; t < ch; t += THREADS) key[c0 + t] = lk[t], idx[c0 + t] = li[t];
         __syncthreads();
      }
   };
   chunks_in_lds(2, ch, true); // stages 2 .. ch entirely inside the chunks
   for (int k2 = ch << 1; k2 <= n2; k2 <<= 1) {
      for (int j = k2 >> 1; j >= ch; j >>= 1) { // across chunks: through global memory
         for (int i = tid; i < n2; i += THREADS) {
            const int p = i ^ j;
            if (p > i) {
               unsigned long long ki = key[i], kp = key[p];
               int ii = idx[i], ip = idx[p];
               const unsigned long long k0 = ki;
               const int i0 = ii;
               cx(ki, ii, kp, ip, (i & k2) == 0);
               if (ki != k0 || ii != i0) key[i] = ki, idx[i] = ii, key[p] = kp, idx[p] = ip;
            }
         }
         __syncthreads();
      }
      chunks_in_lds(k2, k2, false); // ... and the stage's distances below ch
   }
}

} // namespace sb
