// strawberry_amd/csrc/binseq_api.hip -- sbgpu_binseq_device / sbgpu_binseq_host (include/sbgpu.h):
// launch of the per-bin sequence statistics kernel (binseq_device.h).
#include <hip/hip_runtime.h>

#include <cmath>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/sbgpu.h"
#include "api_internal.h"
#include "binseq_device.h"

using sb::api_fail;

namespace {
// g_binseq_log_n on the current device, once per device and process: log(n) by the host's libm
std::mutex g_log_mutex;
uint64_t g_log_done = 0; // bit d: device d has its table
int ensure_log_table(const sbgpu_ctx_t *c)
{
   const int dev = sb::ctx_device(c); // the tables live on the context's device, whatever the caller's current one is
   hipError_t e = hipSetDevice(dev);
   if (e != hipSuccess) return api_fail(SBGPU_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
   std::lock_guard<std::mutex> lock(g_log_mutex);
   if (dev < 64 && ((g_log_done >> dev) & 1)) return SBGPU_OK;
   std::vector<double> t(sb::kBinSeqChunk + 1, 0.0);
   for (int n = 1; n <= sb::kBinSeqChunk; ++n) t[n] = std::log((double)n);
   e = hipMemcpyToSymbol(HIP_SYMBOL(sb::g_binseq_log_n), t.data(), t.size() * sizeof(double));
   if (e != hipSuccess) return api_fail(SBGPU_EHIP, std::string("hipMemcpyToSymbol(log table): ") + hipGetErrorString(e));
   uint8_t lut[256] = {0}; // include/kmer.h:91-124 (ToDna2, ToDna)
   lut['C'] = lut['c'] = 1 | 4;
   lut['G'] = lut['g'] = 2 | 4;
   lut['T'] = lut['t'] = 3;
   lut[1] = lut[2] = 4;
   e = hipMemcpyToSymbol(HIP_SYMBOL(sb::g_binseq_lut), lut, sizeof(lut));
   if (e != hipSuccess) return api_fail(SBGPU_EHIP, std::string("hipMemcpyToSymbol(base table): ") + hipGetErrorString(e));
   if (dev < 64) g_log_done |= 1ull << dev;
   return SBGPU_OK;
}
} // namespace

extern "C" {

int sbgpu_binseq_device(sbgpu_ctx_t *c, const uint8_t *d_genome, int64_t genome_start, int64_t genome_len, int64_t n_bins,
                        const int64_t *d_seg_off, const uint32_t *d_seg_left, const uint32_t *d_seg_right, double *d_gc,
                        double *d_entropy, uint8_t *d_flags, int32_t *d_error, void *stream)
{
   if (!c) return api_fail(SBGPU_EINVAL, "sbgpu_binseq_device: null context");
   if (n_bins == 0) return SBGPU_OK;
   if (n_bins < 0 || genome_len < 0 || genome_start < 0) return api_fail(SBGPU_EINVAL, "sbgpu_binseq_device: bad counts");
   if (!d_genome || !d_seg_off || !d_seg_left || !d_seg_right || !d_gc || !d_entropy || !d_flags || !d_error)
      return api_fail(SBGPU_EINVAL, "sbgpu_binseq_device: null device pointer");
   if (n_bins > 0x7fffffff) return api_fail(SBGPU_ESHAPE, "sbgpu_binseq_device: more than 2^31 - 1 bins in one call");
   if (genome_len > 0xffffffffll) return api_fail(SBGPU_ESHAPE, "sbgpu_binseq_device: a genome window of 2^32 bases or more");
   if (int rc = ensure_log_table(c)) return rc;
   sb::BinSeqArgs a;
   a.genome = d_genome;
   a.genome_start = genome_start;
   a.genome_len = genome_len;
   a.n_bins = n_bins;
   a.seg_off = d_seg_off;
   a.seg_left = d_seg_left;
   a.seg_right = d_seg_right;
   a.gc = d_gc;
   a.entropy = d_entropy;
   a.flags = d_flags;
   a.error = d_error;
   hipLaunchKernelGGL(sb::binseq_kernel, dim3((unsigned)n_bins), dim3(64), 0, (hipStream_t)stream, a);
   hipError_t e = hipGetLastError();
   if (e != hipSuccess) return api_fail(SBGPU_EHIP, std::string("binseq_kernel: ") + hipGetErrorString(e));
   return SBGPU_OK;
}

int sbgpu_binseq_host(sbgpu_ctx_t *c, const uint8_t *genome, int64_t genome_start, int64_t genome_len, int64_t n_bins,
                      const int64_t *seg_off, const uint32_t *seg_left, const uint32_t *seg_right, double *gc_out,
                      double *entropy_out, uint8_t *flags_out)
{
   if (!c) return api_fail(SBGPU_EINVAL, "sbgpu_binseq_host: null context");
   if (n_bins == 0) return SBGPU_OK;
   if (n_bins < 0 || genome_len < 0 || genome_start < 0) return api_fail(SBGPU_EINVAL, "sbgpu_binseq_host: bad counts");
   if (!genome || !seg_off || !gc_out || !entropy_out || !flags_out)
      return api_fail(SBGPU_EINVAL, "sbgpu_binseq_host: null argument");
   if (seg_off[0] != 0) return api_fail(SBGPU_EINVAL, "sbgpu_binseq_host: offsets must start at 0");
   for (int64_t b = 0; b < n_bins; ++b)
      if (seg_off[b + 1] < seg_off[b]) return api_fail(SBGPU_EINVAL, "sbgpu_binseq_host: decreasing seg_off");
   const int64_t n_seg = seg_off[n_bins];
   if (n_seg && (!seg_left || !seg_right)) return api_fail(SBGPU_EINVAL, "sbgpu_binseq_host: null coordinate array");
   for (int64_t b = 0; b < n_bins; ++b) {
      int64_t len = 0;
      for (int64_t s = seg_off[b]; s < seg_off[b + 1]; ++s) {
         const int64_t l = seg_left[s], r = seg_right[s];
         if (r < l || l < genome_start || r - genome_start >= genome_len)
            return api_fail(SBGPU_EINVAL, "sbgpu_binseq_host: a segment lies outside the genome window");
         len += r - l + 1;
      }
      if (len > 0x7fffffff) return api_fail(SBGPU_ESHAPE, "sbgpu_binseq_host: a bin longer than 2^31 - 1 bases");
   }
   struct Part {
      const void *src;
      size_t bytes, off;
   };
   Part parts[] = {{seg_off, (size_t)(n_bins + 1) * 8, 0}, {seg_left, (size_t)n_seg * 4, 0}, {seg_right, (size_t)n_seg * 4, 0},
                   {genome, (size_t)genome_len, 0}};
   size_t total = 0;
   for (Part &p : parts) {
      p.off = total;
      total += (p.bytes + 255) & ~(size_t)255;
   }
   const size_t o_gc = total, o_ent = o_gc + (((size_t)n_bins * 8 + 255) & ~(size_t)255),
                o_fl = o_ent + (((size_t)n_bins * 8 + 255) & ~(size_t)255), o_err = o_fl + (((size_t)n_bins + 255) & ~(size_t)255);
   total = o_err + 256;
   char *d = nullptr;
   hipError_t e = hipMalloc(&d, total);
   if (e != hipSuccess)
      return api_fail(e == hipErrorOutOfMemory ? SBGPU_ENOMEM : SBGPU_EHIP, std::string("hipMalloc: ") + hipGetErrorString(e));
   hipStream_t s = sb::ctx_stream(c);
   auto bail = [&](hipError_t err, const char *what) {
      (void)hipFree(d);
      return api_fail(SBGPU_EHIP, std::string(what) + ": " + hipGetErrorString(err));
   };
   for (Part &p : parts)
      if (p.bytes && (e = hipMemcpyAsync(d + p.off, p.src, p.bytes, hipMemcpyHostToDevice, s)) != hipSuccess)
         return bail(e, "hipMemcpyAsync(H2D)");
   if ((e = hipMemsetAsync(d + o_err, 0, 4, s)) != hipSuccess) return bail(e, "hipMemsetAsync");
   int rc = sbgpu_binseq_device(c, (const uint8_t *)(d + parts[3].off), genome_start, genome_len, n_bins,
                                (const int64_t *)(d + parts[0].off), (const uint32_t *)(d + parts[1].off),
                                (const uint32_t *)(d + parts[2].off), (double *)(d + o_gc), (double *)(d + o_ent),
                                (uint8_t *)(d + o_fl), (int32_t *)(d + o_err), s);
   if (rc != SBGPU_OK) {
      (void)hipFree(d);
      return rc;
   }
   int32_t err = 0;
   if ((e = hipMemcpyAsync(gc_out, d + o_gc, (size_t)n_bins * 8, hipMemcpyDeviceToHost, s)) != hipSuccess ||
       (e = hipMemcpyAsync(entropy_out, d + o_ent, (size_t)n_bins * 8, hipMemcpyDeviceToHost, s)) != hipSuccess ||
       (e = hipMemcpyAsync(flags_out, d + o_fl, (size_t)n_bins, hipMemcpyDeviceToHost, s)) != hipSuccess ||
       (e = hipMemcpyAsync(&err, d + o_err, 4, hipMemcpyDeviceToHost, s)) != hipSuccess)
      return bail(e, "hipMemcpyAsync(D2H)");
   if ((e = hipStreamSynchronize(s)) != hipSuccess) return bail(e, "hipStreamSynchronize");
   (void)hipFree(d);
   if (err) return api_fail(SBGPU_EINVAL, "sbgpu_binseq_host: the kernel rejected a bin (segment outside the genome window)");
   return SBGPU_OK;
}

} // extern "C"
