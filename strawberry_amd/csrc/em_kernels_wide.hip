// strawberry_amd/csrc/em_kernels_wide.hip -- the cooperative multi-workgroup kernel for wide loci
#include "em_wide.h"

namespace sb {
hipError_t launch_wide(int nslot, const WideArgs &g, int n_blocks, size_t lds_bytes, hipStream_t s)
{
   WideArgs arg = g;
   void *params[] = {&arg};
   const void *fn = nullptr;
   if (nslot <= 2) fn = (const void *)em_wide_kernel<2>;
   else if (nslot <= 4) fn = (const void *)em_wide_kernel<4>;
   else fn = (const void *)em_wide_kernel<8>;
   // cooperative: all workgroups resident, or the launch fails -- the barrier among a locus' workgroups needs it
   return hipLaunchCooperativeKernel(fn, dim3((unsigned)n_blocks), dim3(kWideThreads), params, (unsigned)lds_bytes, s);
}
} // namespace sb
