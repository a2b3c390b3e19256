// strawberry_amd/csrc/em_kernels_wide.hip -- the cooperative multi-workgroup kernel for wide loci
#define SB_COMPILE_WIDE_KERNEL 1
#include "em_wide.h"

namespace sb {
hipError_t launch_wide(const WideArgs &g, int n_blocks, size_t lds_bytes, hipStream_t s)
{
   WideArgs arg = g;
   void *params[] = {&arg};
   // cooperative: all workgroups resident, or the launch fails -- the exchange among a locus' workgroups needs it
   return hipLaunchCooperativeKernel(arg.a.row_bias ? (const void *)em_wide_kernel<true> : (const void *)em_wide_kernel<false>,
                                     dim3((unsigned)n_blocks), dim3(kWideThreads), params, (unsigned)lds_bytes, s);
}
} // namespace sb
