// strawberry_amd/csrc/sbgpu_api.hip -- the extern "C" surface of libsbgpu.so
// (include/sbgpu.h): context, plan upload, kernel dispatch, abundance epilogue.
// gfx950 only; there is no CPU fallback: every entry point fails with SBGPU_EHIP /
// SBGPU_ENODEV when HIP or the device is unavailable.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cmath>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/sbgpu.h"
#include "api_internal.h"
#include "binweight_device.h"
#include "em_device.h"
#include "em_wide.h"
#include "plan.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
   g_err = msg;
   return code;
}

#define HIP_TRY(expr)                                                                         \
   do {                                                                                       \
      hipError_t e_ = (expr);                                                                 \
      if (e_ != hipSuccess)                                                                   \
         return fail(SBGPU_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
   } while (0)

constexpr int kAuxStreams = 8;
constexpr int kMaxPhaseEvents = 6; // later phases of the wave kind that get timing events
// Which aux stream each kernel kind runs on.  HIP folds streams onto a few hardware queues
// (4 by default: GPU_MAX_HW_QUEUES) in creation order; measured on ROCm 7.2 the aux streams
// land on queues {2,3,4,4,3,2,1,4} and the null stream on 3, so the three kinds that run
// side by side (one wave kind, block, tall block) get streams 0, 2 and 6: three distinct queues
// that also avoid the caller's.
constexpr int kKindStream[sb::kNumKinds] = {0, 0, 0, 2, 6, 1};

} // namespace

struct sbgpu_ctx {
   int device = 0;
   int n_cu = 256;
   hipDeviceProp_t prop;
   hipStream_t stream = nullptr;            // the context's own stream
   hipStream_t aux[kAuxStreams] = {};       // size classes run concurrently on these
   hipEvent_t fork = nullptr;
   hipEvent_t join[kAuxStreams] = {};
   unsigned split_runs = 0; // sbgpu_em_run_device_split calls so far (their wave kinds alternate between two streams)
   int wave_alt = -1;       // the wave kinds' second stream: -1 not looked for yet, 0 none (no alternation), else its index
   hipEvent_t t0[sb::kNumKinds] = {}, t1[sb::kNumKinds] = {}; // per-kind kernel timing (sbgpu_set_timing)
   hipEvent_t tp[kMaxPhaseEvents + 1] = {};                   // ends of the wave kind's phases
   bool timed[sb::kNumKinds] = {};
   bool timing = false; // record the timing events (off by default: they cost a few microseconds per step)
   int n_phase_timed = 0;
   // grow-only device scratch of the multi-stage entry points (a hipMalloc / hipFree pair of a few GB per call costs
   // tens to hundreds of milliseconds): sb::ctx_scratch
   char *scratch[8] = {};
   size_t scratch_bytes[8] = {};
   int32_t *d_pdf_support = nullptr; // [5] device: support of the insert-size table of the bin-weight launch in flight (pdf_support_kernel)
   int32_t *wide_error = nullptr; // pinned host word the wide-locus kernel raises when a barrier times out
   // kernel stages of the chain entry points, bracketed by events while `timing` is on (sb::ctx_stage_begin / _end)
   static constexpr int kMaxStages = 16;
   hipEvent_t stage_ev[kMaxStages][2] = {};
   const char *stage_name[kMaxStages] = {};
   int n_stages = 0;
   bool stage_open = false;
   // host-side helpers of the grouping (sb::ctx_pinned, ctx_copy_stream, ctx_event, ctx_pairs_hint)
   char *pinned[4] = {};
   size_t pinned_bytes[4] = {};
   hipStream_t copy_stream = nullptr;
   hipEvent_t order_ev[6] = {};
   size_t pairs_hint = 0;
   sb::ResidentAnnotation *resident = nullptr; // sbgpu_annotation_pin
};

namespace sb {
int api_fail(int code, const std::string &msg) { return fail(code, msg); }
hipStream_t ctx_stream(const sbgpu_ctx_t *ctx) { return ctx->stream; }
hipStream_t ctx_aux_stream(const sbgpu_ctx_t *ctx, int i) { return (i >= 0 && i < kAuxStreams) ? ctx->aux[i] : ctx->stream; }
int ctx_cu_count(const sbgpu_ctx_t *ctx) { return ctx->n_cu; }
int ctx_device(const sbgpu_ctx_t *ctx) { return ctx->device; }
void ctx_stage_reset(sbgpu_ctx_t *ctx) { ctx->n_stages = 0, ctx->stage_open = false; }
void ctx_stage_begin(sbgpu_ctx_t *ctx, const char *name, hipStream_t s)
{
   if (!ctx->timing || ctx->stage_open || ctx->n_stages >= sbgpu_ctx::kMaxStages) return;
   hipEvent_t *ev = ctx->stage_ev[ctx->n_stages];
   for (int i = 0; i < 2; ++i)
      if (!ev[i] && hipEventCreate(&ev[i]) != hipSuccess) return;
   if (hipEventRecord(ev[0], s) != hipSuccess) return;
   ctx->stage_name[ctx->n_stages] = name;
   ctx->stage_open = true;
}
void ctx_stage_end(sbgpu_ctx_t *ctx, hipStream_t s)
{
   if (!ctx->stage_open) return;
   (void)hipEventRecord(ctx->stage_ev[ctx->n_stages][1], s);
   ctx->stage_open = false;
   ++ctx->n_stages;
}
hipError_t ctx_pinned(sbgpu_ctx_t *ctx, int slot, size_t bytes, char **out)
{
   *out = nullptr;
   if (slot < 0 || slot >= 4) return hipErrorInvalidValue;
   if (bytes < 4096) bytes = 4096;
   if (ctx->pinned_bytes[slot] < bytes) {
      if (ctx->pinned[slot]) (void)hipHostFree(ctx->pinned[slot]);
      ctx->pinned[slot] = nullptr;
      ctx->pinned_bytes[slot] = 0;
      const size_t want = bytes + bytes / 4;
      hipError_t e = hipHostMalloc((void **)&ctx->pinned[slot], want, hipHostMallocDefault);
      if (e != hipSuccess) return e;
      ctx->pinned_bytes[slot] = want;
   }
   *out = ctx->pinned[slot];
   return hipSuccess;
}
hipError_t ctx_copy_stream(sbgpu_ctx_t *ctx, hipStream_t *out)
{
   if (!ctx->copy_stream) {
      hipError_t e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking);
      if (e != hipSuccess) return e;
   }
   *out = ctx->copy_stream;
   return hipSuccess;
}
hipError_t ctx_event(sbgpu_ctx_t *ctx, int which, hipEvent_t *out)
{
   if (which < 0 || which >= 6) return hipErrorInvalidValue;
   if (!ctx->order_ev[which]) {
      hipError_t e = hipEventCreateWithFlags(&ctx->order_ev[which], hipEventDisableTiming);
      if (e != hipSuccess) return e;
   }
   *out = ctx->order_ev[which];
   return hipSuccess;
}
namespace {
struct DevBlock {
   char *p;
   size_t cap;
   int device;
};
struct DevPool {
   std::mutex m;
   std::vector<DevBlock> blocks;
   size_t bytes = 0;
   size_t limit = 0; // idle bytes the pool may hold; 0: not decided yet
};
// How much idle memory the pool may hold.  The arenas of a sample-sized call are tens of GB (a chain sample's alignment
// records decode into 25 GB of reads, pair into 15 GB of pairs, ...): with a 4 GB pool every one of them went through
// hipMalloc and hipFree in every call -- 2.9 of a step's 4.1 seconds at 3.9e8 records.  The part is built around 288 GB:
// half of the device's memory may idle here (SBGPU_POOL_GB overrides; an allocation that fails -- the library's own
// -- lets the idle blocks go and tries again, see dev_take).
size_t dev_pool_limit(DevPool &pool)
{
   if (pool.limit) return pool.limit;
   size_t lim = (size_t)4 << 30;
   if (const char *e = std::getenv("SBGPU_POOL_GB")) {
      lim = (size_t)std::max(0.0, std::atof(e) * 1073741824.0);
   } else {
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b / 2 > lim) lim = total_b / 2;
      else (void)hipGetLastError();
   }
   pool.limit = lim ? lim : 1;
   return pool.limit;
}
DevPool &dev_pool()
{
   static DevPool *p = new DevPool(); // never destroyed: handles may be released during static destruction
   return *p;
}
} // namespace
hipError_t dev_take(size_t bytes, char **out, size_t *capacity)
{
   *out = nullptr;
   *capacity = 0;
   if (bytes < 256) bytes = 256;
   int device = 0;
   hipError_t e = hipGetDevice(&device);
   if (e != hipSuccess) return e;
   DevPool &pool = dev_pool();
   {
      std::lock_guard<std::mutex> g(pool.m);
      size_t best = pool.blocks.size();
      for (size_t i = 0; i < pool.blocks.size(); ++i) {
         const DevBlock &b = pool.blocks[i];
         if (b.device == device && b.cap >= bytes && b.cap <= 2 * bytes + (1u << 20) && (best == pool.blocks.size() || b.cap < pool.blocks[best].cap)) best = i;
      }
      if (best < pool.blocks.size()) {
         *out = pool.blocks[best].p;
         *capacity = pool.blocks[best].cap;
         pool.bytes -= pool.blocks[best].cap;
         pool.blocks.erase(pool.blocks.begin() + (long)best);
         return hipSuccess;
      }
   }
   const size_t want = bytes + bytes / 16; // head-room: the next request is often a little larger
   e = hipMalloc(out, want);
   if (e != hipSuccess) {
      // out of memory with blocks idle in the pool: let them go and try once more
      std::vector<DevBlock> idle;
      {
         std::lock_guard<std::mutex> g(pool.m);
         idle.swap(pool.blocks);
         pool.bytes = 0;
      }
      for (const DevBlock &b : idle) (void)hipFree(b.p);
      (void)hipGetLastError();
      e = hipMalloc(out, want);
      if (e != hipSuccess) return e;
   }
   *capacity = want;
   return hipSuccess;
}
size_t dev_release_idle()
{
   std::vector<DevBlock> idle;
   size_t bytes = 0;
   {
      DevPool &pool = dev_pool();
      std::lock_guard<std::mutex> g(pool.m);
      idle.swap(pool.blocks);
      bytes = pool.bytes;
      pool.bytes = 0;
   }
   for (const DevBlock &b : idle) (void)hipFree(b.p);
   return bytes;
}
namespace {
thread_local int g_give_without_device_sync = 0;
}
DevGiveStreamSynced::DevGiveStreamSynced() { ++g_give_without_device_sync; }
DevGiveStreamSynced::~DevGiveStreamSynced() { --g_give_without_device_sync; }
void dev_give(char *block, size_t capacity)
{
   if (!block) return;
   // The block belongs to the device it was allocated on, whichever device is current in the calling thread (a process
   // that drives several GPUs destroys handles while another device is current): ask the runtime, wait for THAT device,
   // pool the block under it, and leave the caller's current device as it was.
   int current = 0, owner = -1;
   const bool have_current = hipGetDevice(&current) == hipSuccess;
   hipPointerAttribute_t attr;
   if (hipPointerGetAttributes(&attr, block) == hipSuccess) owner = attr.device;
   else (void)hipGetLastError();
   if (owner < 0) owner = have_current ? current : 0;
   const bool switched = have_current && owner != current && hipSetDevice(owner) == hipSuccess;
   // hipFree waits for the device before it lets memory go, and callers have relied on that (a handle destroyed while a
   // kernel on the CALLER's stream still reads its arena): a block that goes back to the pool waits the same way.  On an
   // idle device -- every call site has synchronised its own stream already -- this costs ~10 us.
   if (!g_give_without_device_sync) (void)hipDeviceSynchronize();
   bool pooled = false;
   {
      DevPool &pool = dev_pool();
      std::lock_guard<std::mutex> g(pool.m);
      if (pool.blocks.size() < 24 && pool.bytes + capacity <= dev_pool_limit(pool)) {
         pool.blocks.push_back({block, capacity, owner});
         pool.bytes += capacity;
         pooled = true;
      }
   }
   if (!pooled) (void)hipFree(block);
   if (switched) (void)hipSetDevice(current);
}
const ResidentAnnotation *ctx_resident_annotation(const sbgpu_ctx_t *ctx) { return ctx->resident; }
void ctx_set_resident_annotation(sbgpu_ctx_t *ctx, ResidentAnnotation *r)
{
   if (ctx->resident) {
      (void)hipStreamSynchronize(ctx->stream);
      dev_give(ctx->resident->arena, ctx->resident->capacity);
      delete ctx->resident;
   }
   ctx->resident = r;
}
size_t ctx_pairs_hint(const sbgpu_ctx_t *ctx) { return ctx->pairs_hint; }
void ctx_set_pairs_hint(sbgpu_ctx_t *ctx, size_t bytes) { ctx->pairs_hint = bytes; }
hipError_t ctx_scratch(sbgpu_ctx_t *ctx, int slot, size_t bytes, char **out)
{
   *out = nullptr;
   if (slot < 0 || slot >= 8) return hipErrorInvalidValue;
   if (bytes < 256) bytes = 256;
   if (ctx->scratch_bytes[slot] < bytes) {
      if (ctx->scratch[slot]) (void)hipFree(ctx->scratch[slot]);
      ctx->scratch[slot] = nullptr;
      ctx->scratch_bytes[slot] = 0;
      const size_t want = bytes + bytes / 4; // head-room: the next call is often a little larger
      hipError_t e = hipMalloc(&ctx->scratch[slot], want);
      if (e != hipSuccess) {
         e = hipMalloc(&ctx->scratch[slot], bytes);
         if (e != hipSuccess) return e;
         ctx->scratch_bytes[slot] = bytes;
      } else {
         ctx->scratch_bytes[slot] = want;
      }
   }
   *out = ctx->scratch[slot];
   return hipSuccess;
}
bool ctx_take_wide_error(sbgpu_ctx_t *ctx)
{
   if (!ctx->wide_error || !*ctx->wide_error) return false;
   *ctx->wide_error = 0;
   return true;
}
} // namespace sb

struct KindLaunch {
   int first_class = 0, n_classes = 0; // range in plan->host.classes
   int n_blocks = 0;
   int block_threads = 0;
   sb::ClassDesc *d_table = nullptr;   // points into plan->d_tables
};

struct sbgpu_plan {
   sbgpu_ctx *ctx = nullptr;
   sb::HostPlan host;
   KindLaunch launches[sb::kNumKinds];
   char *d_arena = nullptr;            // one allocation (sb::dev_take), one upload: the arrays below point into it
   size_t arena_cap = 0;
   int64_t *d_row_off = nullptr, *d_iso_off = nullptr, *d_f_off = nullptr;
   int32_t *d_loci_all = nullptr;      // all class lists, concatenated (input of phase 0)
   int32_t *d_class_n = nullptr;       // loci per class (input count of phase 0)
   char *d_zero = nullptr;             // the later phases' survivor counts and batch totals: zeroed before every run
   size_t zero_bytes = 0;
   // later phases of the wave kind (plan.h: LatPhase): class table (first blocks filled on the device), survivor
   // counts, number of batches, survivor lists, and the route INTO the phase
   struct LatDev {
      sb::ClassDesc *d_table = nullptr;
      int32_t *d_counts = nullptr, *d_total = nullptr, *d_lists = nullptr, *d_route = nullptr;
      int n_classes = 0, it_limit = 0, grid = 0;
      bool repack = false;
   };
   std::vector<LatDev> lat;
   sb::ClassDesc *d_tables = nullptr;  // one descriptor per class
   std::vector<int64_t> loci_off;      // per class: offset into d_loci_all
   uint8_t *d_row_keep = nullptr;      // streaming path: init() row flags
   // SBGPU_GRAPH=1: the launches of sbgpu_em_run_device captured once per (plan, argument pointers, stream) and replayed
   // (an experiment: DESIGN.md section 8)
   mutable hipGraphExec_t graph_exec = nullptr;
   mutable const void *graph_key[6] = {};
   double *d_locus_sum = nullptr;      // abundance epilogue: kept-FPKM sum per workgroup of 256 loci
   unsigned *d_epi_ticket = nullptr;   // abundance epilogue: finished workgroups (the last one sums; it leaves 0 behind)
   size_t stream_lds_bytes = 0;
   // wide loci (kStream class) served by the cooperative multi-workgroup kernel, in launches ("rounds")
   struct WideRound {
      int first_desc = 0, n_desc = 0, n_blocks = 0;
      size_t lds_bytes = 0;
   };
   std::vector<WideRound> wide_rounds;
   sb::WideDesc *d_wide_table = nullptr; // all rounds' descriptors
   double *d_wide_bufs = nullptr;
   mutable unsigned wide_epoch = 0;    // bumped by every run: the tag space of its exchange granules
   int32_t n_wide_desc = 0, n_wide_loci = 0; // the first n_wide_loci of the stream class' list go to the wide kernel
};

namespace {

// ------------------------------------------------------------------ epilogue kernels
// LocusContext::estimate_abundances, /root/reference/src/estimate.cpp:314-355.
// One thread per locus; the per-locus FPKM sum runs in isoform order like the
// reference's loop (:315-336).  Returns the sum of the locus' kept FPKM.
__device__ double abundance_locus(int64_t l, const int64_t *iso_off, const double *theta, const int32_t *status,
                                  const int32_t *length, const sbgpu_abundance_params_t &p, double *fpkm, double *frac,
                                  int32_t *keep)
{
   const int64_t j0 = iso_off[l], j1 = iso_off[l + 1];
   if (status[l] == sb::kStInitEmpty) {
      // estimate_abundances() returns false: quantifyCluster returns no isoforms
      // (/root/reference/src/alignments.cpp:1524-1545)
      for (int64_t j = j0; j < j1; ++j) {
         fpkm[j] = 0.0;
         frac[j] = 0.0;
         keep[j] = 0;
      }
      return 0.0;
   }
   const double rpm = 1e6 / (double)p.total_mapped_reads; // :328
   // Both passes go four isoforms at a time: the loads of a group are issued together (a locus of 200 isoforms
   // otherwise pays one memory round trip per isoform and pass), the arithmetic stays in isoform order.
   double sum_fpkm = 0.0;
   for (int64_t jb = j0; jb < j1; jb += 4) {
      int32_t len[4];
      double th[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
         const bool in = jb + u < j1;
         len[u] = in ? length[jb + u] : 1;
         th[u] = in ? theta[jb + u] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
         if (jb + u >= j1) break;
         double kb;
         int32_t k = 1;
         double f = 0.0;
         if (p.effective_len_norm) { // :317-324
            kb = (double)len[u] - p.insert_mean;
            if (kb < 0) {
               k = 2; // "NA"
            } else {
               kb = 1e3 / kb;
            }
         } else {
            kb = 1e3 / (double)len[u]; // :326
         }
         if (k != 2) {
            f = th[u] * rpm * kb; // :329
            sum_fpkm += f;
         }
         fpkm[jb + u] = f;
         keep[jb + u] = k;
      }
   }
   double kept_sum = 0.0;
   for (int64_t jb = j0; jb < j1; jb += 4) {
      double f[4];
      int32_t kk[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
         const bool in = jb + u < j1;
         f[u] = in ? fpkm[jb + u] : 0.0;
         kk[u] = in ? keep[jb + u] : 0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
         if (jb + u >= j1) break;
         double fr = 0.0;
         int32_t k = kk[u];
         if (k != 2) fr = f[u] / sum_fpkm;                              // :342
         if (p.filter_by_expression && fr < p.min_isoform_frac) k = 0;  // :346-355
         frac[jb + u] = fr;
         keep[jb + u] = k;
         if (k) kept_sum += f[u];
      }
   }
   return kept_sum;
}

// The workgroup that finishes LAST adds the per-workgroup sums (fixed strided order + fixed tree: deterministic,
// whichever workgroup it is) and writes the total -- Sample::procSample's FPKM sum, alignments.cpp:1821-1824 --
// so no second launch is needed.  Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): the sums are
// stored and loaded with agent-scope (sc1) accesses, the storing lane waits for its store before it takes its
// ticket, and the workgroup whose ticket is the last one reads only after its add has returned.
__global__ __launch_bounds__(256) void abundance_kernel(int64_t n_loci, const int64_t *iso_off, const double *theta,
                                                        const int32_t *status, const int32_t *length,
                                                        sbgpu_abundance_params_t p, double *fpkm, double *frac,
                                                        int32_t *keep, double *block_sum, unsigned *ticket, double *total)
{
   const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   __shared__ double part[256];
   __shared__ int s_last;
   part[threadIdx.x] = l < n_loci ? abundance_locus(l, iso_off, theta, status, length, p, fpkm, frac, keep) : 0.0;
   __syncthreads();
   // this workgroup's 256 loci summed in a fixed tree: the global sum is deterministic
   for (int w = 128; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
      __syncthreads();
   }
   if (threadIdx.x == 0) {
      __hip_atomic_store(&block_sum[blockIdx.x], part[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = t == gridDim.x - 1;
   }
   __syncthreads();
   if (!s_last) return;
   double acc = 0.0;
   for (unsigned i = threadIdx.x; i < gridDim.x; i += 256)
      acc += __hip_atomic_load(&block_sum[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   part[threadIdx.x] = acc;
   __syncthreads();
   for (int w = 128; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
      __syncthreads();
   }
   if (threadIdx.x == 0) {
      *total = part[0];
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next launch
   }
}

// Deterministic sum of x[0..n) (the abundance kernel's per-workgroup sums) in one workgroup (fixed strided
// order + fixed tree), written to *out.  Sample::procSample, alignments.cpp:1821-1824.
__global__ __launch_bounds__(1024) void sum_kernel(int64_t n, const double *x, double *out)
{
   __shared__ double s[1024];
   double acc = 0.0;
   for (int64_t i = threadIdx.x; i < n; i += 1024) acc += x[i];
   s[threadIdx.x] = acc;
   __syncthreads();
   for (int w = 512; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
      __syncthreads();
   }
   if (threadIdx.x == 0) *out = s[0];
}

// Holds a stream back for `ticks` of the 100 MHz device clock (one wave, asleep most of the time).
__global__ void delay_kernel(unsigned long long ticks)
{
   const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
   while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// alignments.cpp:1825-1829
__global__ void tpm_kernel(int64_t n, const double *fpkm, const int32_t *keep, const double *total,
                           double *tpm)
{
   const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   const double t = *total;
   tpm[i] = keep[i] ? 1e6 * fpkm[i] / t : 0.0;
}

} // namespace

// ================================================================== C ABI
extern "C" {

const char *sbgpu_version(void) { return "libsbgpu 0.1 (gfx950, strawberry EM hot path)"; }

#ifndef SBGPU_BUILD_ID
#define SBGPU_BUILD_ID "unknown"
#endif
const char *sbgpu_build_id(void) { return SBGPU_BUILD_ID; }

const char *sbgpu_last_error(void) { return g_err.c_str(); }

int sbgpu_device_count(void)
{
   int n = 0;
   if (hipGetDeviceCount(&n) != hipSuccess) return 0;
   return n;
}

int sbgpu_init(int device, sbgpu_ctx_t **ctx_out)
{
   if (!ctx_out) return fail(SBGPU_EINVAL, "sbgpu_init: null ctx_out");
   *ctx_out = nullptr;
   int n = 0;
   if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(SBGPU_ENODEV, "sbgpu_init: no HIP device");
   if (device < 0 || device >= n) return fail(SBGPU_ENODEV, "sbgpu_init: device index out of range");
   HIP_TRY(hipSetDevice(device));
   sbgpu_ctx *c = new (std::nothrow) sbgpu_ctx();
   if (!c) return fail(SBGPU_ENOMEM, "sbgpu_init: out of host memory");
   c->device = device;
   hipError_t e = hipGetDeviceProperties(&c->prop, device);
   if (e != hipSuccess) {
      delete c;
      return fail(SBGPU_EHIP, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
   }
   if (std::strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
      std::string arch = c->prop.gcnArchName;
      delete c;
      return fail(SBGPU_ENODEV, "sbgpu_init: device is " + arch + ", libsbgpu is built for gfx950 only");
   }
   c->n_cu = c->prop.multiProcessorCount;
   e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
   // The few workgroups of the block and stream kinds run the longest loci: their streams get the higher queue
   // priority, so that they take their slots before the thousands of short wave-form workgroups do
   // (SBGPU_STREAM_PRIORITY=0 turns that off; A/B measurements).
   int prio_least = 0, prio_greatest = 0;
   (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
   const bool use_prio = !(sb::exp_env("SBGPU_STREAM_PRIORITY") && std::atoi(sb::exp_env("SBGPU_STREAM_PRIORITY")) == 0);
   for (int i = 0; e == hipSuccess && i < kAuxStreams; ++i) {
      const bool long_kind = i == kKindStream[sb::kNumKinds - 1] || i == kKindStream[sb::kNumKinds - 2] || i == kKindStream[sb::kNumKinds - 3];
      e = hipStreamCreateWithPriority(&c->aux[i], hipStreamNonBlocking, (use_prio && long_kind) ? prio_greatest : 0);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&c->join[i], hipEventDisableTiming);
   }
   if (e == hipSuccess) e = hipEventCreateWithFlags(&c->fork, hipEventDisableTiming);
   if (e == hipSuccess) e = hipMalloc((void **)&c->d_pdf_support, 8 * sizeof(int32_t));
   if (e == hipSuccess) e = hipHostMalloc((void **)&c->wide_error, sizeof(int32_t), hipHostMallocDefault);
   if (e == hipSuccess) *c->wide_error = 0;
   for (int k = 0; e == hipSuccess && k < sb::kNumKinds; ++k) {
      e = hipEventCreate(&c->t0[k]);
      if (e == hipSuccess) e = hipEventCreate(&c->t1[k]);
   }
   for (int i = 0; e == hipSuccess && i <= kMaxPhaseEvents; ++i) e = hipEventCreate(&c->tp[i]);
   if (e != hipSuccess) {
      sbgpu_finalize(c);
      return fail(SBGPU_EHIP, std::string("sbgpu_init: stream/event creation: ") + hipGetErrorString(e));
   }
   *ctx_out = c;
   return SBGPU_OK;
}

int sbgpu_finalize(sbgpu_ctx_t *c)
{
   if (!c) return SBGPU_OK;
   (void)hipSetDevice(c->device);
   for (int i = 0; i < kAuxStreams; ++i) {
      if (c->aux[i]) (void)hipStreamDestroy(c->aux[i]);
      if (c->join[i]) (void)hipEventDestroy(c->join[i]);
   }
   if (c->fork) (void)hipEventDestroy(c->fork);
   for (int k = 0; k < sb::kNumKinds; ++k) {
      if (c->t0[k]) (void)hipEventDestroy(c->t0[k]);
      if (c->t1[k]) (void)hipEventDestroy(c->t1[k]);
   }
   for (int i = 0; i <= kMaxPhaseEvents; ++i)
      if (c->tp[i]) (void)hipEventDestroy(c->tp[i]);
   if (c->stream) (void)hipStreamDestroy(c->stream);
   if (c->wide_error) (void)hipHostFree(c->wide_error);
   for (auto &ev : c->stage_ev)
      for (hipEvent_t e : ev)
         if (e) (void)hipEventDestroy(e);
   if (c->d_pdf_support) (void)hipFree(c->d_pdf_support);
   for (int i = 0; i < 8; ++i)
      if (c->scratch[i]) (void)hipFree(c->scratch[i]);
   for (int i = 0; i < 4; ++i)
      if (c->pinned[i]) (void)hipHostFree(c->pinned[i]);
   for (hipEvent_t ev : c->order_ev)
      if (ev) (void)hipEventDestroy(ev);
   if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
   sb::ctx_set_resident_annotation(c, nullptr);
   delete c;
   return SBGPU_OK;
}

int sbgpu_device_info(sbgpu_ctx_t *c, int64_t out[8])
{
   if (!c || !out) return fail(SBGPU_EINVAL, "sbgpu_device_info: null argument");
   out[0] = c->prop.multiProcessorCount;
   out[1] = c->prop.warpSize;
   out[2] = (int64_t)c->prop.maxSharedMemoryPerMultiProcessor;
   out[3] = c->prop.clockRate;
   out[4] = (int64_t)(c->prop.totalGlobalMem >> 20);
   out[5] = out[6] = out[7] = 0;
   return SBGPU_OK;
}

int64_t sbgpu_release_idle_memory(void) { return (int64_t)sb::dev_release_idle(); }

int sbgpu_synchronize(sbgpu_ctx_t *c, void *stream)
{
   if (!c) return fail(SBGPU_EINVAL, "sbgpu_synchronize: null ctx");
   HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
   if (c->wide_error && *c->wide_error) {
      *c->wide_error = 0;
      return fail(SBGPU_EHIP, "a barrier of the wide-locus EM kernel timed out: the loci it served have no result");
   }
   return SBGPU_OK;
}

int sbgpu_plan_destroy(sbgpu_plan_t *p)
{
   if (!p) return SBGPU_OK;
   if (p->ctx) (void)hipSetDevice(p->ctx->device);
   // every device array of the plan lives in this one allocation; it goes back to the pool (dev_give waits for the device,
   // as hipFree did: a run on the caller's stream may still be reading it)
   sb::dev_give(p->d_arena, p->arena_cap);
   if (p->graph_exec) (void)hipGraphExecDestroy(p->graph_exec);
   delete p;
   return SBGPU_OK;
}

int sbgpu_plan_create(sbgpu_ctx_t *c, int64_t n_loci, const int64_t *row_off, const int64_t *iso_off,
                      const int64_t *f_off, sbgpu_plan_t **plan_out)
{
   if (!c || !plan_out) return fail(SBGPU_EINVAL, "sbgpu_plan_create: null argument");
   *plan_out = nullptr;
   sbgpu_plan *p = new (std::nothrow) sbgpu_plan();
   if (!p) return fail(SBGPU_ENOMEM, "sbgpu_plan_create: out of host memory");
   p->ctx = c;
   const char *err = "";
   sb::PlanTuning tune;
   if (const char *e = sb::exp_env("SBGPU_WAVE_RMULT")) tune.wave_rmult = std::atoi(e);
   if (const char *e = sb::exp_env("SBGPU_MAX_WAVES")) tune.max_waves = std::atoll(e);
   if (const char *e = sb::exp_env("SBGPU_LIGHT_BLOCK")) tune.light_block = std::atoi(e) != 0;
   if (const char *e = sb::exp_env("SBGPU_ORDER")) tune.order_by_work = std::string(e) == "work";
   if (const char *e = sb::exp_env("SBGPU_CLASS_ORDER")) tune.classes_by_prediction = std::string(e) != "cost";
   // SBGPU_PHASES="32,128,512": iteration limits at which the wave kind's loci are suspended and continue in
   // lane-rich layouts ("0" or "": one phase); SBGPU_PHASE_LAMBDA="8,2,0.25": the later phases' lane weights
   auto parse_list = [](const char *ev, auto conv, auto *out) {
      std::string spec = ev;
      size_t pos = 0;
      while (pos < spec.size()) {
         size_t q = spec.find(',', pos);
         if (q == std::string::npos) q = spec.size();
         out->push_back(conv(spec.substr(pos, q - pos)));
         pos = q + 1;
      }
   };
   if (const char *e = sb::exp_env("SBGPU_PHASES")) {
      tune.phases_auto = false;
      parse_list(e, [](const std::string &t) { return std::atoi(t.c_str()); }, &tune.phase_limits);
   }
   // a lane weight "t" makes the phase re-pack its survivors into their phase-0 layouts instead ("tile" phases)
   if (const char *e = sb::exp_env("SBGPU_PHASE_LAMBDA"))
      parse_list(e, [](const std::string &t) { return (!t.empty() && (t[0] == 't' || t[0] == 'T')) ? -1.0 : std::atof(t.c_str()); }, &tune.phase_lambda);
   const bool timing = std::getenv("SBGPU_HOST_TIMING") != nullptr; // diagnostic: stage times on stderr
   auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
   double t_stage = now();
   auto stage = [&](const char *name) {
      if (timing) {
         const double t = now();
         std::fprintf(stderr, "sbgpu_plan_create: %-12s %.2f ms\n", name, (t - t_stage) * 1e3);
         t_stage = t;
      }
   };
   int rc = sb::build_host_plan(n_loci, row_off, iso_off, f_off, c->n_cu, tune, &p->host, &err);
   stage("size classes");
   if (rc != SBGPU_OK) {
      delete p;
      return fail(rc, err);
   }
   auto bail = [&](hipError_t e, const char *what) {
      sbgpu_plan_destroy(p);
      return fail(e == hipErrorOutOfMemory ? SBGPU_ENOMEM : SBGPU_EHIP, std::string(what) + ": " + hipGetErrorString(e));
   };
   hipError_t e = hipSetDevice(c->device);
   if (e != hipSuccess) return bail(e, "hipSetDevice");
   const size_t nb = (size_t)(n_loci + 1) * sizeof(int64_t);
   // ---- wide loci: the streaming class is served by the multi-workgroup kernel wherever a locus' rows fit
   // into n_cu workgroups; such loci move to the front of the class list, the rest keeps the streaming kernel
   std::vector<sb::WideDesc> wide_table;
   size_t wide_buf_doubles = 0;
   for (sb::SizeClass &sc : p->host.classes) {
      if (sc.kind != sb::kStream) continue;
      // (workgroups, locus): a round's time is its slowest locus', and an iteration costs the more the more
      // workgroups exchange partials, so loci of like workgroup counts share rounds: largest first
      std::vector<std::pair<int, int32_t>> wide;
      std::vector<int32_t> rest;
      // a locus' layout is the one that serves it with the fewest workgroups (em_wide.h: 16 / 32 / 64 column lanes x
      // 4-8 columns; rows per workgroup = register rows + LDS rows); the rows are then dealt evenly
      auto groups_of = [&](int32_t l, int &layout) -> int64_t {
         const int64_t nrow = row_off[l + 1] - row_off[l], niso = iso_off[l + 1] - iso_off[l];
         layout = sb::wide_layout_for(niso, nrow);
         if (layout < 0) return -1;
         const int64_t rpb = sb::wide_rows_per_block(layout);
         return std::max<int64_t>(1, (nrow + rpb - 1) / rpb);
      };
      static const bool no_wide = std::getenv("SBGPU_NO_WIDE") != nullptr;
      for (int32_t l : sc.loci) {
         int layout;
         const int64_t G = groups_of(l, layout);
         if (G > 0 && G <= c->n_cu && row_off[l + 1] > row_off[l] && !no_wide) wide.emplace_back((int)G, l);
         else rest.push_back(l);
      }
      std::stable_sort(wide.begin(), wide.end(), [](const std::pair<int, int32_t> &x, const std::pair<int, int32_t> &y) { return x.first > y.first; });
      sc.loci.clear();
      // Rounds: every launch holds at most n_cu workgroups, all resident.  First fit, most workgroups first: a locus goes
      // to the first round that has room (a round is as long as its slowest locus, and a locus of fewer workgroups is never
      // slower than one of more: filling the gaps of the early rounds with small loci costs nothing and can save the last,
      // mostly empty launch).  A round's descriptors are consecutive in the table, ordered by their first workgroup.
      std::vector<std::vector<std::pair<int, int32_t>>> members;
      std::vector<int> used;
      for (const auto &gl : wide) {
         size_t r = 0;
         while (r < used.size() && used[r] + gl.first > c->n_cu) ++r;
         if (r == used.size()) {
            used.push_back(0);
            members.emplace_back();
         }
         used[r] += gl.first;
         members[r].push_back(gl);
      }
      for (size_t r = 0; r < members.size(); ++r) {
         sbgpu_plan::WideRound round;
         round.first_desc = (int)wide_table.size();
         for (const auto &gl : members[r]) {
            const int32_t l = gl.second;
            const int G = gl.first;
            const int64_t nrow = row_off[l + 1] - row_off[l];
            int layout;
            (void)groups_of(l, layout);
            sb::WideDesc d;
            d.locus = l;
            d.first_block = round.n_blocks;
            d.n_blocks = G;
            d.rows_per_block = (int32_t)std::max<int64_t>(1, (nrow + G - 1) / G);
            d.npad = sb::wide_cols(layout);
            d.buf_off = (int64_t)wide_buf_doubles;
            d.layout = layout;
            d.lb_slice = sb::wide_lb_slice(d.npad, G);
            d.pad_ = 0;
            wide_buf_doubles += (size_t)4 * G * d.npad + (size_t)4 * d.npad; // two buffers of G x npad 16-byte granules (partials) + two of npad (totals)
            round.n_blocks += G;
            round.n_desc += 1;
            round.lds_bytes = std::max(round.lds_bytes, sb::wide_lds_bytes(layout));
            wide_table.push_back(d);
            sc.loci.push_back(l);
         }
         p->wide_rounds.push_back(round);
      }
      p->n_wide_loci = (int32_t)sc.loci.size();
      sc.loci.insert(sc.loci.end(), rest.begin(), rest.end());
      sc.n_blocks = (int)rest.size(); // what is left for the streaming kernel
   }
   p->n_wide_desc = (int32_t)wide_table.size();
   const size_t ncls = p->host.classes.size();
   const size_t ncls_alloc = ncls + 1;
   const size_t nlat = p->host.lat.size();
   int64_t n_phased = 0; // loci of the phased wave kind = capacity of every later phase's lists
   if (nlat)
      for (int32_t cap : p->host.lat[0].capacity) n_phased += cap;
   // ---- one device arena; its head (offsets, class lists, tables, class sizes) is staged on the host
   // and uploaded with a single copy, the rest is workspace
   auto up = [](size_t bytes) { return (bytes + 255) & ~(size_t)255; };
   size_t at = 0;
   const size_t o_row = at; at += up(nb);
   const size_t o_iso = at; at += up(nb);
   const size_t o_f = at; at += up(nb);
   const size_t o_loci = at; at += up((size_t)(n_loci + 1) * sizeof(int32_t));
   const size_t o_tab = at; at += up(ncls_alloc * sizeof(sb::ClassDesc));
   const size_t o_cn = at; at += up(ncls_alloc * sizeof(int32_t));
   const size_t o_wtab = at; at += up((wide_table.size() + 1) * sizeof(sb::WideDesc));
   std::vector<size_t> o_ltab(nlat), o_lroute(nlat), o_lcnt(nlat), o_ltot(nlat), o_llist(nlat);
   for (size_t i = 0; i < nlat; ++i) {
      o_ltab[i] = at; at += up((p->host.lat[i].classes.size() + 1) * sizeof(sb::ClassDesc));
      o_lroute[i] = at; at += up((size_t)(n_loci + 1) * sizeof(int32_t));
   }
   const size_t o_tick = at; at += up(sizeof(unsigned)); // (zero: it goes up with the staged head)
   const size_t staged = at;
   // zeroed before every run: every later phase's survivor counts and batch total
   const size_t o_cur = at;
   for (size_t i = 0; i < nlat; ++i) {
      o_lcnt[i] = at; at += up((p->host.lat[i].classes.size() + 1) * sizeof(int32_t));
      o_ltot[i] = at; at += up(sizeof(int32_t));
   }
   p->zero_bytes = at - o_cur;
   for (size_t i = 0; i < nlat; ++i) {
      o_llist[i] = at; at += up((size_t)(n_phased + 1) * sizeof(int32_t));
   }
   const size_t o_keep = at; at += up((size_t)p->host.n_rows + 1);
   const size_t o_sum = at; at += up((size_t)(n_loci + 1) * sizeof(double));
   const size_t o_wbuf = at; at += up((wide_buf_doubles + 1) * sizeof(double));
   if ((e = sb::dev_take(at, &p->d_arena, &p->arena_cap)) != hipSuccess) return bail(e, "hipMalloc(plan arena)");
   stage("hipMalloc");
   p->d_row_off = (int64_t *)(p->d_arena + o_row);
   p->d_iso_off = (int64_t *)(p->d_arena + o_iso);
   p->d_f_off = (int64_t *)(p->d_arena + o_f);
   p->d_loci_all = (int32_t *)(p->d_arena + o_loci);
   p->d_tables = (sb::ClassDesc *)(p->d_arena + o_tab);
   p->d_class_n = (int32_t *)(p->d_arena + o_cn);
   p->d_zero = p->d_arena + o_cur;
   p->lat.resize(nlat);
   for (size_t i = 0; i < nlat; ++i) {
      sbgpu_plan::LatDev &ld = p->lat[i];
      const sb::LatPhase &lp = p->host.lat[i];
      ld.d_table = (sb::ClassDesc *)(p->d_arena + o_ltab[i]);
      ld.d_route = (int32_t *)(p->d_arena + o_lroute[i]);
      ld.d_counts = (int32_t *)(p->d_arena + o_lcnt[i]);
      ld.d_total = (int32_t *)(p->d_arena + o_ltot[i]);
      ld.d_lists = (int32_t *)(p->d_arena + o_llist[i]);
      ld.n_classes = (int)lp.classes.size();
      ld.it_limit = lp.it_limit;
      ld.repack = lp.repack;
      // every batch gets a wave of its own up to 32768 waves; beyond that the waves stride over the batches
      ld.grid = (int)std::min<int64_t>(std::max<int64_t>(lp.max_blocks, 1), 32768);
   }
   p->d_row_keep = (uint8_t *)(p->d_arena + o_keep);
   p->d_locus_sum = (double *)(p->d_arena + o_sum);
   p->d_epi_ticket = (unsigned *)(p->d_arena + o_tick);
   p->d_wide_table = (sb::WideDesc *)(p->d_arena + o_wtab);
   p->d_wide_bufs = (double *)(p->d_arena + o_wbuf);
   std::vector<char> stage_buf(staged, 0);
   if (n_loci > 0) {
      std::memcpy(stage_buf.data() + o_row, row_off, nb);
      std::memcpy(stage_buf.data() + o_iso, iso_off, nb);
      std::memcpy(stage_buf.data() + o_f, f_off, nb);
   }
   int32_t *h_loci = (int32_t *)(stage_buf.data() + o_loci);
   sb::ClassDesc *table = (sb::ClassDesc *)(stage_buf.data() + o_tab);
   int32_t *h_cn = (int32_t *)(stage_buf.data() + o_cn);
   if (!wide_table.empty()) std::memcpy(stage_buf.data() + o_wtab, wide_table.data(), wide_table.size() * sizeof(sb::WideDesc));
   for (size_t i = 0; i < nlat; ++i) {
      const sb::LatPhase &lp = p->host.lat[i];
      sb::ClassDesc *lt = (sb::ClassDesc *)(stage_buf.data() + o_ltab[i]);
      int32_t loff = 0;
      for (size_t ci = 0; ci < lp.classes.size(); ++ci) {
         const sb::SizeClass &sc = lp.classes[ci];
         lt[ci].block_begin = 0; // phase_prepare_kernel fills it from the survivor counts
         lt[ci].n = lp.capacity[ci];
         lt[ci].loci_off = loff;
         lt[ci].shape = sc.layout | (sc.rmult << 8) | (sc.lbG << 16);
         loff += lp.capacity[ci];
      }
      std::memcpy(stage_buf.data() + o_lroute[i], lp.route.data(), lp.route.size() * sizeof(int32_t));
   }
   size_t off = 0;
   size_t max_stream_iso = 0;
   for (int k = 0; k < sb::kNumKinds; ++k) p->launches[k] = KindLaunch();
   for (size_t ci = 0; ci < ncls; ++ci) {
      const sb::SizeClass &sc = p->host.classes[ci];
      KindLaunch &kl = p->launches[sc.kind];
      if (kl.n_classes == 0) {
         kl.first_class = (int)ci;
         kl.d_table = p->d_tables + ci;
         kl.block_threads = sc.block_threads;
      }
      table[ci].block_begin = kl.n_blocks;
      table[ci].n = (int32_t)sc.loci.size();
      table[ci].loci_off = (int32_t)off;
      table[ci].shape = sc.layout | (sc.rmult << 8) | (sc.lbG << 16);
      h_cn[ci] = table[ci].n;
      kl.n_blocks += sc.n_blocks;
      kl.n_classes += 1;
      std::memcpy(h_loci + off, sc.loci.data(), sc.loci.size() * sizeof(int32_t));
      p->loci_off.push_back((int64_t)off);
      off += sc.loci.size();
      if (sc.kind == sb::kStream) {
         for (int32_t l : sc.loci) {
            size_t k = (size_t)(iso_off[l + 1] - iso_off[l]);
            if (k > max_stream_iso) max_stream_iso = k;
         }
      }
   }
   stage("staging");
   if ((e = hipMemcpy(p->d_arena, stage_buf.data(), staged, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy(plan)");
   // the wide kernel's exchange granules carry a per-run tag: start from a state no tag matches
   if (wide_buf_doubles && (e = hipMemset(p->d_wide_bufs, 0, wide_buf_doubles * sizeof(double))) != hipSuccess) return bail(e, "hipMemset(plan)");
   stage("upload");
   // streaming kernel LDS: (3 + NWAVE) * npad doubles, npad <= pow2ceil-padded niso
   size_t npad = 1;
   while (npad < max_stream_iso && npad < 64) npad <<= 1;
   if (max_stream_iso > 64) npad = 64 * ((max_stream_iso + 63) / 64);
   p->stream_lds_bytes = (3 + sb::kStreamThreads / 64) * npad * sizeof(double);
   *plan_out = p;
   return SBGPU_OK;
}

int sbgpu_plan_info(const sbgpu_plan_t *p, int64_t out[8])
{
   if (!p || !out) return fail(SBGPU_EINVAL, "sbgpu_plan_info: null argument");
   out[0] = p->host.n_loci;
   out[1] = p->host.n_rows;
   out[2] = p->host.n_iso;
   out[3] = p->host.n_elem;
   out[4] = (int64_t)p->host.classes.size();
   out[5] = p->host.n_stream_loci;
   out[6] = p->host.algorithmic_bytes;
   out[7] = 0;
   return SBGPU_OK;
}

int sbgpu_plan_classes(const sbgpu_plan_t *p, int64_t *out, int cap)
{
   if (!p) return fail(SBGPU_EINVAL, "sbgpu_plan_classes: null plan");
   int n = (int)p->host.classes.size();
   for (int i = 0; i < n && i < cap && out; ++i) {
      const sb::SizeClass &sc = p->host.classes[i];
      out[i * 6 + 0] = sc.kind;
      out[i * 6 + 1] = sc.CPL * sc.CL;
      out[i * 6 + 2] = sc.R;
      out[i * 6 + 3] = sc.G;
      out[i * 6 + 4] = (int64_t)sc.loci.size();
      out[i * 6 + 5] = (int64_t)sc.n_blocks * (sc.block_threads / 64);
   }
   return n;
}

} // extern "C"

// fp64 (the product path) and fp32 (BASELINE config 5's tolerance sweep) share the launch structure
// A side stream whose kernels run BESIDE those of the three streams the kinds use (0, 2, 6): four 200 us holds, one per stream,
// take ~200 us together when the candidate has a hardware queue of its own and ~400 when it shares one (HIP folds a process'
// streams onto 4 queues by creation order; which one a stream got cannot be asked).  0: none found.
static int find_wave_alt(sbgpu_ctx_t *c)
{
   const int used[3] = {kKindStream[sb::kWaveH], kKindStream[sb::kBlock], kKindStream[sb::kBlockTall]};
   const int cand[5] = {1, 3, 4, 5, 7};
   if (hipDeviceSynchronize() != hipSuccess) return 0;
   for (int pass = 0; pass < 2; ++pass) // (the first pass also loads the kernel: its times are not looked at)
      for (int ci = 0; ci < 5; ++ci) {
         const auto t0 = std::chrono::steady_clock::now();
         for (int u = 0; u < 3; ++u) hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, c->aux[used[u]], 200ull * 100ull);
         hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, c->aux[cand[ci]], 200ull * 100ull);
         if (hipDeviceSynchronize() != hipSuccess) return 0;
         const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
         if (pass == 1 && us < 330.0) return cand[ci];
      }
   return 0;
}

static int em_run_impl(sbgpu_ctx_t *c, const sbgpu_plan_t *p, const int32_t *d_count, const void *d_F_any, void *d_theta_any,
                       int32_t *d_status, int32_t *d_iters, void *stream, const bool f32, const void *d_row_bias = nullptr,
                       const void *d_iso_bias = nullptr, void *join_stream = nullptr)
{
   if (!c || !p) return fail(SBGPU_EINVAL, "sbgpu_em_run_device: null ctx/plan");
   if (p->host.n_loci == 0) return SBGPU_OK;
   if (!d_theta_any || !d_status || !d_iters || (!d_count && p->host.n_rows) || (!d_F_any && p->host.n_elem))
      return fail(SBGPU_EINVAL, "sbgpu_em_run_device: null device pointer");
   if (f32 && (p->launches[sb::kStream].n_classes > 0 || p->launches[sb::kWaveH].n_classes > 0))
      return fail(SBGPU_EUNSUPPORTED, "sbgpu_em_run_device_f32: the fp32 variant covers loci of up to 64 isoforms in the tile kernels only");
   if ((d_row_bias != nullptr) != (d_iso_bias != nullptr)) return fail(SBGPU_EINVAL, "sbgpu_em_run_device_bias: both bias arrays or none");
   // the bias factors are applied where a kernel loads its tile: the tile kernels and, round 5, the multi-workgroup kernel
   // of the wide loci; the streaming fallback (more than 512 isoforms or 256 workgroups) and the later phases do not
   bool stream_fallback = false;
   if (p->launches[sb::kStream].n_classes > 0)
      stream_fallback = (int32_t)p->host.classes[p->launches[sb::kStream].first_class].loci.size() > p->n_wide_loci;
   if (d_row_bias && (stream_fallback || !p->lat.empty()))
      return fail(SBGPU_EUNSUPPORTED, "sbgpu_em_run_device_bias: the bias factors are applied at tile load (tile kernels and the wide-locus kernel: "
                                      "no locus on the streaming fallback, no phases)");
   hipStream_t main = (hipStream_t)stream;
   sb::EmArgs a;
   a.row_bias = f32 ? nullptr : (const double *)d_row_bias;
   a.iso_bias = f32 ? nullptr : (const double *)d_iso_bias;
   a.row_off = p->d_row_off;
   a.iso_off = p->d_iso_off;
   a.f_off = p->d_f_off;
   a.count = d_count;
   a.F = (const double *)d_F_any;
   a.theta = (double *)d_theta_any;
   a.status = d_status;
   a.iters = d_iters;
   sb::EmArgsT<float> a32;
   a32.row_off = a.row_off, a32.iso_off = a.iso_off, a32.f_off = a.f_off, a32.count = a.count;
   a32.F = (const float *)d_F_any, a32.theta = (float *)d_theta_any, a32.status = d_status, a32.iters = d_iters;
   a32.row_bias = f32 ? (const float *)d_row_bias : nullptr, a32.iso_bias = f32 ? (const float *)d_iso_bias : nullptr;
   // Batches are dealt to the workgroups statically: nothing to reset between runs but the later phases' survivor
   // counts (only when the plan has phases)
   // (split runs: a plan with per-run state of its own -- the phases' survivor counts, the wide loci's barrier words -- must
   // not start while the previous run's kernels still use it: such a plan's runs wait for each other as the unsplit ones do)
   if (join_stream && (p->zero_bytes || p->n_wide_desc))
      for (int k = 0; k < sb::kNumKinds; ++k)
         if (p->launches[k].n_classes > 0) HIP_TRY(hipStreamWaitEvent(main, c->join[k], 0));
   if (p->zero_bytes) HIP_TRY(hipMemsetAsync(p->d_zero, 0, p->zero_bytes, main));
   // a locus the kernels never reach (a wide-locus barrier that timed out) must not look solved: with wide loci in
   // the plan, status starts at -1 = SBGPU_EM_UNSOLVED (0xFF bytes); every other kernel writes all its loci
   if (p->n_wide_desc) HIP_TRY(hipMemsetAsync(d_status, 0xFF, (size_t)p->host.n_loci * sizeof(int32_t), main));
   // one launch per kind (wave / block / stream); a single kind runs on the
   // caller's stream, several fork onto the aux streams and join back
   int kinds = 0;
   for (int k = 0; k < sb::kNumKinds; ++k) kinds += p->launches[k].n_classes > 0;
   const bool fork = kinds > 1;
   // SBGPU_WAVE_ON_MAIN=1: the wave kinds (stream slot 0: the kinds that end last) run on the caller's stream itself, so
   // that neither their start nor the epilogue behind them waits for an event to cross hardware queues; the other
   // kinds fork off and join as before
   static const bool wave_on_main = sb::exp_env("SBGPU_WAVE_ON_MAIN") && std::atoi(sb::exp_env("SBGPU_WAVE_ON_MAIN")) != 0;
   // Split runs alternate the wave kinds (the kinds that end last) between two streams on two hardware queues: the next run's
   // wave kernel -- its longest loci first -- starts while this run's last workgroups still work through theirs, instead of
   // behind them (C3: 0.76 -> 0.72 ms per step; a second stream that shares a hardware queue with the first one or with the
   // block kinds': 0.79).  Which stream that is depends on how the runtime folded the process' streams onto its hardware queues:
   // it is looked for once (find_wave_alt), and without one the runs do not alternate.
   if (join_stream && fork && c->wave_alt < 0) {
      c->wave_alt = find_wave_alt(c);
      if (std::getenv("SBGPU_HOST_TIMING")) std::fprintf(stderr, "[sbgpu] split runs: the wave kinds' second stream = %d (0: none)\n", c->wave_alt);
   }
   const int wave_slot = (join_stream && fork && c->wave_alt > 0 && !p->zero_bytes && !p->n_wide_desc && (c->split_runs++ & 1)) ? c->wave_alt : 0;
   auto stream_of = [&](int k) -> hipStream_t {
      if (!fork || (wave_on_main && kKindStream[k] == 0)) return main;
      return c->aux[kKindStream[k] == 0 ? wave_slot : kKindStream[k]];
   };
   if (fork) {
      HIP_TRY(hipEventRecord(c->fork, main));
      for (int k = 0; k < sb::kNumKinds; ++k)
         if (p->launches[k].n_classes > 0 && stream_of(k) != main) HIP_TRY(hipStreamWaitEvent(stream_of(k), c->fork, 0));
   }
   for (int k = 0; k < sb::kNumKinds; ++k) c->timed[k] = false;
   c->n_phase_timed = 0;
   const bool timing = c->timing;
   // longest iterations first: stream, block, wave
   // The wide-locus rounds are cooperative launches that want the chip to themselves (one 512-lane workgroup with all
   // of a CU's registers and up to 94 KB of its LDS per CU): beside the tile kinds' workgroups their residency is not
   // assured and the rounds take twice as long (measured: C3-T 44 -> 79 ms).  They run when the tile kinds are done.
   const bool wide_last = fork && p->n_wide_desc > 0;
   for (int pass = 0; pass < 2; ++pass)
   for (int k = sb::kNumKinds - 1; k >= 0; --k) {
      const KindLaunch &kl = p->launches[k];
      if (kl.n_classes == 0) continue;
      if ((pass == 1) != (wide_last && k == sb::kStream)) continue; // pass 0: everything but a deferred stream kind
      if (pass == 1) {
         for (int o = 0; o < sb::kNumKinds; ++o) {
            if (o == k || p->launches[o].n_classes == 0) continue;
            HIP_TRY(hipEventRecord(c->join[o], stream_of(o)));
            HIP_TRY(hipStreamWaitEvent(stream_of(k), c->join[o], 0));
         }
      }
      hipStream_t s = stream_of(k);
      // Start order.  A tall-tile workgroup needs a whole CU's registers, a block-kind one half of them: once the
      // workgroups of a lighter kind have spread over the chip, a heavier one waits for a CU to drain -- the 9
      // tall workgroups of C3 then finish at 1.48 ms instead of 0.65 ms, behind everything else (they start within a
      // few microseconds of the block kind, in either order).  So every kind is held back a few microseconds behind
      // the next heavier one: tall, block (+SBGPU_BLOCK_DELAY_US), wave (+SBGPU_WAVE_DELAY_US more); 0 = off.
      // C3: 1.49 -> 1.45 ms per step with the wave delay at 5 (no better at 10-40).
      static const int wave_delay_us = sb::exp_env("SBGPU_WAVE_DELAY_US") ? std::atoi(sb::exp_env("SBGPU_WAVE_DELAY_US")) : 5;
      static const int block_delay_us = sb::exp_env("SBGPU_BLOCK_DELAY_US") ? std::atoi(sb::exp_env("SBGPU_BLOCK_DELAY_US")) : 5;
      const bool tall_runs = p->launches[sb::kBlockTall].n_classes > 0;
      int hold_us = 0;
      if (fork && k == sb::kBlock && tall_runs) hold_us = block_delay_us;
      if (fork && kKindStream[k] == 0) hold_us = wave_delay_us + ((tall_runs && p->launches[sb::kBlock].n_classes > 0) ? block_delay_us : 0);
      // (split runs: a kind's kernels follow each other on its stream, run after run, and the kinds drift apart by themselves --
      // the hold would only sit on the path of the kind that ends last)
      if (join_stream && !(p->zero_bytes || p->n_wide_desc)) hold_us = 0;
      if (hold_us > 0) {
         hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, s, (unsigned long long)hold_us * 100ull);
         HIP_TRY(hipGetLastError());
      }
      if (timing) HIP_TRY(hipEventRecord(c->t0[k], s));
      if (k == sb::kStream) {
         // wide loci: cooperative launches, one per round (all workgroups of a launch are resident)
         if (p->n_wide_desc) {
            p->wide_epoch = (p->wide_epoch % 0x1FFFFFu) + 1; // 1 .. 2^21 - 1: epoch * 2048 + round fits 32 bits
            int32_t *d_err = nullptr;
            HIP_TRY(hipHostGetDevicePointer((void **)&d_err, c->wide_error, 0));
            // Rounds run one after the other on this kind's stream: a round fills the chip (one workgroup per CU, all
            // registers), its loci run the same number of exchanges give or take, and a cooperative launch is only
            // promised residency on an otherwise free device -- two rounds in flight could both end up half resident
            // and spin in their exchanges until the timeout.
            for (const sbgpu_plan::WideRound &r : p->wide_rounds) {
               sb::WideArgs wa;
               wa.a = a;
               wa.table = p->d_wide_table + r.first_desc;
               wa.n_desc = r.n_desc;
               wa.bufs = p->d_wide_bufs;
               wa.epoch = p->wide_epoch;
               wa.error = d_err;
               HIP_TRY(sb::launch_wide(wa, r.n_blocks, r.lds_bytes, s));
            }
         }
         const int32_t n_all = (int32_t)p->host.classes[kl.first_class].loci.size();
         if (n_all > p->n_wide_loci) { // the rest: one workgroup per locus, F streamed from L2
            sb::ClassArgs ca = {};
            ca.loci = p->d_loci_all + p->loci_off[kl.first_class] + p->n_wide_loci;
            ca.n = n_all - p->n_wide_loci;
            ca.batch = 0;
            HIP_TRY(sb::launch_stream(a, ca, p->d_row_keep, ca.n, p->stream_lds_bytes, s));
         }
      } else {
         const bool wave_kind = k == sb::kWaveH || k == sb::kWave1 || k == sb::kWave2;
         const bool phased = wave_kind && !p->lat.empty() && !f32;
         sb::FusedLaunch fl;
         fl.a = a;
         fl.ph.table = kl.d_table;
         fl.ph.n_classes = kl.n_classes;
         fl.ph.lists_in = p->d_loci_all;
         fl.ph.n_in = p->d_class_n + kl.first_class;
         fl.ph.n_batches = kl.n_blocks;
         fl.ph.total_blocks = nullptr;
         fl.ph.lists_out = phased ? p->lat[0].d_lists : nullptr;
         fl.ph.n_out = phased ? p->lat[0].d_counts : nullptr;
         fl.ph.route = phased ? p->lat[0].d_route : nullptr;
         fl.ph.next_table = phased ? p->lat[0].d_table : nullptr;
         fl.ph.it_limit = phased ? p->host.first_limit : SBGPU_EM_MAX_ITER;
         fl.ph.resume = 0;
         fl.n_blocks = std::max(1, (int)(kl.n_blocks * p->host.grid_scale + 0.5));
         hipError_t e = hipErrorInvalidValue;
         if (f32) {
            sb::FusedLaunchF32 f;
            f.a = a32, f.ph = fl.ph, f.n_blocks = fl.n_blocks;
            e = sb::launch_fused_f32(k, f, s);
         } else if (k == sb::kWaveH) e = sb::launch_fused_wave_h(fl, s);
         else if (k == sb::kWave1) e = sb::launch_fused_wave_1(fl, s);
         else if (k == sb::kWave2) e = sb::launch_fused_wave_2(fl, s);
         else if (k == sb::kBlock) e = sb::launch_fused_block(fl, s);
         else if (k == sb::kBlockTall) e = sb::launch_fused_block_tall(fl, s);
         HIP_TRY(e);
         // later phases: the loci still running continue in lane-rich layouts; a one-workgroup kernel turns the
         // survivor counts into the next launch's block table in between
         for (size_t i = 0; phased && i < p->lat.size(); ++i) {
            const sbgpu_plan::LatDev &ld = p->lat[i];
            const bool last = i + 1 == p->lat.size();
            if (timing && i < (size_t)kMaxPhaseEvents) HIP_TRY(hipEventRecord(c->tp[i], s));
            HIP_TRY(sb::launch_phase_prepare(ld.d_table, ld.d_counts, ld.n_classes, ld.d_total, s));
            sb::FusedLaunch ll;
            ll.a = a;
            ll.ph.table = ld.d_table;
            ll.ph.n_classes = ld.n_classes;
            ll.ph.lists_in = ld.d_lists;
            ll.ph.n_in = ld.d_counts;
            ll.ph.n_batches = 0;
            ll.ph.total_blocks = ld.d_total;
            ll.ph.lists_out = last ? nullptr : p->lat[i + 1].d_lists;
            ll.ph.n_out = last ? nullptr : p->lat[i + 1].d_counts;
            ll.ph.route = last ? nullptr : p->lat[i + 1].d_route;
            ll.ph.next_table = last ? nullptr : p->lat[i + 1].d_table;
            ll.ph.it_limit = ld.it_limit;
            ll.ph.resume = 1;
            ll.n_blocks = ld.grid;
            hipError_t el = hipErrorInvalidValue;
            if (!ld.repack) el = sb::launch_lat(ll, s);
            else if (k == sb::kWaveH) el = sb::launch_fused_wave_h(ll, s);
            else if (k == sb::kWave1) el = sb::launch_fused_wave_1(ll, s);
            else el = sb::launch_fused_wave_2(ll, s);
            HIP_TRY(el);
            if (timing) c->n_phase_timed = (int)std::min<size_t>(i + 1, kMaxPhaseEvents);
         }
         if (timing && phased && p->lat.size() <= (size_t)kMaxPhaseEvents) HIP_TRY(hipEventRecord(c->tp[p->lat.size()], s));
      }
      if (timing) {
         HIP_TRY(hipEventRecord(c->t1[k], s));
         c->timed[k] = true;
      }
   }
   // the kinds' streams join the caller's -- or, sbgpu_em_run_device_split, ANOTHER stream of the caller's (the one its epilogue
   // runs on), so that the stream the next run forks from does not wait for this one's kernels
   hipStream_t join_to = join_stream ? (hipStream_t)join_stream : main;
   if (fork) {
      for (int k = 0; k < sb::kNumKinds; ++k) {
         if (p->launches[k].n_classes == 0) continue;
         if (stream_of(k) == join_to) continue;
         HIP_TRY(hipEventRecord(c->join[k], stream_of(k)));
         HIP_TRY(hipStreamWaitEvent(join_to, c->join[k], 0));
      }
   } else if (join_to != main) {
      HIP_TRY(hipEventRecord(c->fork, main));
      HIP_TRY(hipStreamWaitEvent(join_to, c->fork, 0));
   }
   return SBGPU_OK;
}

extern "C" {

int sbgpu_em_run_device(sbgpu_ctx_t *c, const sbgpu_plan_t *p, const int32_t *d_count, const double *d_F,
                        double *d_theta, int32_t *d_status, int32_t *d_iters, void *stream)
{
   static const bool use_graph = sb::exp_env("SBGPU_GRAPH") && std::atoi(sb::exp_env("SBGPU_GRAPH")) != 0;
   if (!use_graph || !c || !p || c->timing || p->n_wide_desc) // (cooperative launches and timing events stay outside)
      return em_run_impl(c, p, d_count, d_F, d_theta, d_status, d_iters, stream, false);
   const void *key[6] = {d_count, d_F, d_theta, d_status, d_iters, stream};
   if (!p->graph_exec || std::memcmp(key, p->graph_key, sizeof key) != 0) {
      if (p->graph_exec) (void)hipGraphExecDestroy(p->graph_exec);
      p->graph_exec = nullptr;
      hipGraph_t g = nullptr;
      // captured on the context's own stream (the caller's may be the legacy default stream, which cannot capture);
      // the graph is launched into the caller's
      HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
      const int rc = em_run_impl(c, p, d_count, d_F, d_theta, d_status, d_iters, c->stream, false);
      const hipError_t e = hipStreamEndCapture(c->stream, &g);
      if (rc != SBGPU_OK) {
         if (g) (void)hipGraphDestroy(g);
         return rc;
      }
      HIP_TRY(e);
      const hipError_t ei = hipGraphInstantiate(&p->graph_exec, g, nullptr, nullptr, 0);
      (void)hipGraphDestroy(g);
      HIP_TRY(ei);
      std::memcpy(p->graph_key, key, sizeof key);
   }
   HIP_TRY(hipGraphLaunch(p->graph_exec, (hipStream_t)stream));
   return SBGPU_OK;
}

int sbgpu_em_run_device_split(sbgpu_ctx_t *c, const sbgpu_plan_t *p, const int32_t *d_count, const double *d_F, double *d_theta,
                              int32_t *d_status, int32_t *d_iters, void *fork_stream, void *join_stream)
{
   if (!join_stream) return fail(SBGPU_EINVAL, "sbgpu_em_run_device_split: null join stream");
   return em_run_impl(c, p, d_count, d_F, d_theta, d_status, d_iters, fork_stream, false, nullptr, nullptr, join_stream);
}

int sbgpu_em_run_device_f32(sbgpu_ctx_t *c, const sbgpu_plan_t *p, const int32_t *d_count, const float *d_F,
                            float *d_theta, int32_t *d_status, int32_t *d_iters, void *stream)
{
   return em_run_impl(c, p, d_count, d_F, d_theta, d_status, d_iters, stream, true);
}

int sbgpu_last_stage_ms(sbgpu_ctx_t *c, int cap, float *ms, const char **names)
{
   if (!c || cap < 0 || (cap > 0 && (!ms || !names))) return fail(SBGPU_EINVAL, "sbgpu_last_stage_ms: bad argument");
   int n = 0;
   for (int i = 0; i < c->n_stages && n < cap; ++i) {
      float t = 0.0f;
      if (hipEventSynchronize(c->stage_ev[i][1]) != hipSuccess || hipEventElapsedTime(&t, c->stage_ev[i][0], c->stage_ev[i][1]) != hipSuccess)
         return fail(SBGPU_EHIP, "sbgpu_last_stage_ms: the stage events were not recorded");
      ms[n] = t;
      names[n] = c->stage_name[i];
      ++n;
   }
   return n;
}

int sbgpu_em_run_device_bias(sbgpu_ctx_t *c, const sbgpu_plan_t *p, const int32_t *d_count, const double *d_F, const double *d_row_bias,
                             const double *d_iso_bias, double *d_theta, int32_t *d_status, int32_t *d_iters, void *stream)
{
   return em_run_impl(c, p, d_count, d_F, d_theta, d_status, d_iters, stream, false, d_row_bias, d_iso_bias);
}

int sbgpu_em_run_device_bias_f32(sbgpu_ctx_t *c, const sbgpu_plan_t *p, const int32_t *d_count, const float *d_F, const float *d_row_bias,
                                 const float *d_iso_bias, float *d_theta, int32_t *d_status, int32_t *d_iters, void *stream)
{
   return em_run_impl(c, p, d_count, d_F, d_theta, d_status, d_iters, stream, true, d_row_bias, d_iso_bias);
}

int sbgpu_set_timing(sbgpu_ctx_t *c, int on)
{
   if (!c) return fail(SBGPU_EINVAL, "sbgpu_set_timing: null ctx");
   c->timing = on != 0;
   return SBGPU_OK;
}

int sbgpu_em_last_phase_ms(sbgpu_ctx_t *c, float *ms, int cap)
{
   if (!c || (!ms && cap > 0)) return fail(SBGPU_EINVAL, "sbgpu_em_last_phase_ms: null argument");
   // tp[0] = end of phase 0, tp[i] = end of later phase i; t0[wave kind] = start of phase 0
   int wave = -1;
   for (int k = 0; k <= sb::kWave2; ++k)
      if (c->timed[k]) wave = k;
   if (wave < 0 || c->n_phase_timed == 0) return 0;
   int n = 0;
   for (int i = 0; i <= c->n_phase_timed && n < cap; ++i) {
      HIP_TRY(hipEventSynchronize(c->tp[i]));
      HIP_TRY(hipEventElapsedTime(&ms[n], i == 0 ? c->t0[wave] : c->tp[i - 1], c->tp[i]));
      ++n;
   }
   return n;
}

int sbgpu_em_last_kernel_ms(sbgpu_ctx_t *c, float ms[6])
{
   if (!c || !ms) return fail(SBGPU_EINVAL, "sbgpu_em_last_kernel_ms: null argument");
   for (int k = 0; k < sb::kNumKinds && k < 6; ++k) {
      ms[k] = 0.0f;
      if (!c->timed[k]) continue;
      HIP_TRY(hipEventSynchronize(c->t1[k]));
      HIP_TRY(hipEventElapsedTime(&ms[k], c->t0[k], c->t1[k]));
   }
   return SBGPU_OK;
}

#ifdef SB_STAMPS
// diagnostic build only: one trivial launch per stream, to read the stream -> hardware queue
// mapping off a rocprofv3 kernel trace (grid size = 64 * (index + 1) identifies the stream)
int sbgpu_debug_touch_streams(sbgpu_ctx_t *c, double *d_buf)
{
   for (int i = 0; i < kAuxStreams; ++i) {
      hipLaunchKernelGGL(sum_kernel, dim3(i + 1), dim3(1024), 0, c->aux[i], (int64_t)0, d_buf, d_buf + 1);
      HIP_TRY(hipGetLastError());
   }
   hipLaunchKernelGGL(sum_kernel, dim3(20), dim3(1024), 0, c->stream, (int64_t)0, d_buf, d_buf + 1);
   hipLaunchKernelGGL(sum_kernel, dim3(30), dim3(1024), 0, (hipStream_t)0, (int64_t)0, d_buf, d_buf + 1);
   HIP_TRY(hipDeviceSynchronize());
   return SBGPU_OK;
}

int sbgpu_debug_read_stamps(void *out, size_t bytes)
{
   HIP_TRY(hipDeviceSynchronize());
   // the kernels' translation units each hold their own copy of the buffer: merge them (unwritten slots are 0)
   std::vector<unsigned long long> part(bytes / 8), all(bytes / 8, 0ull);
   hipError_t (*readers[])(void *, size_t) = {sb::read_stamps_wave_h, sb::read_stamps_wave_1, sb::read_stamps_wave_2,
                                              sb::read_stamps_block, sb::read_stamps_block_tall};
   for (auto rd : readers) {
      HIP_TRY(rd(part.data(), part.size() * 8));
      for (size_t i = 0; i < all.size(); ++i) all[i] = std::max(all[i], part[i]);
   }
   std::memcpy(out, all.data(), all.size() * 8);
   return SBGPU_OK;
}
#endif

int sbgpu_plan_locus_kinds(const sbgpu_plan_t *p, int8_t *out)
{
   if (!p || (!out && p->host.n_loci)) return fail(SBGPU_EINVAL, "sbgpu_plan_locus_kinds: null argument");
   for (const sb::SizeClass &sc : p->host.classes)
      for (int32_t l : sc.loci) out[l] = (int8_t)sc.kind;
   return SBGPU_OK;
}

int sbgpu_em_batch(sbgpu_ctx_t *c, const sbgpu_batch_t *b, double *theta_out, int32_t *status_out,
                   int32_t *iters_out)
{
   if (!c || !b) return fail(SBGPU_EINVAL, "sbgpu_em_batch: null argument");
   if (b->n_loci == 0) return SBGPU_OK;
   if (!theta_out || !status_out) return fail(SBGPU_EINVAL, "sbgpu_em_batch: null output");
   if (b->n_loci < 0 || !b->row_off || !b->iso_off || !b->f_off) return fail(SBGPU_EINVAL, "sbgpu_em_batch: bad n_loci or null offset array");
   // The batch's sizes are its last offsets: the uploads (2.5 ms for C3 from pageable memory) start at once on a
   // helper thread while this thread sorts the loci into size classes (2.3 ms) -- sbgpu_plan_create validates the
   // offsets and reports what is wrong with them.
   const int64_t n_rows = b->row_off[b->n_loci], n_iso = b->iso_off[b->n_loci], n_el = b->f_off[b->n_loci];
   if (n_rows < 0 || n_iso < 0 || n_el < 0 || (n_rows && !b->count) || (n_el && !b->F))
      return fail(SBGPU_EINVAL, "sbgpu_em_batch: negative size or null count / F array");
   sbgpu_plan_t *p = nullptr;
   int32_t *d_count = nullptr, *d_status = nullptr, *d_iters = nullptr;
   double *d_F = nullptr, *d_theta = nullptr;
   auto cleanup = [&]() {
      (void)hipFree(d_count);
      (void)hipFree(d_status);
      (void)hipFree(d_iters);
      (void)hipFree(d_F);
      (void)hipFree(d_theta);
      sbgpu_plan_destroy(p);
   };
#define TRY_CLEAN(expr)                                                                       \
   do {                                                                                       \
      hipError_t e_ = (expr);                                                                 \
      if (e_ != hipSuccess) {                                                                 \
         cleanup();                                                                           \
         return fail(e_ == hipErrorOutOfMemory ? SBGPU_ENOMEM : SBGPU_EHIP,                   \
                     std::string(#expr) + ": " + hipGetErrorString(e_));                      \
      }                                                                                       \
   } while (0)
   TRY_CLEAN(hipMalloc(&d_count, (size_t)(n_rows + 1) * sizeof(int32_t)));
   TRY_CLEAN(hipMalloc(&d_F, (size_t)(n_el + 1) * sizeof(double)));
   TRY_CLEAN(hipMalloc(&d_theta, (size_t)(n_iso + 1) * sizeof(double)));
   TRY_CLEAN(hipMalloc(&d_status, (size_t)b->n_loci * sizeof(int32_t)));
   TRY_CLEAN(hipMalloc(&d_iters, (size_t)b->n_loci * sizeof(int32_t)));
   hipError_t up_err = hipSuccess;
   auto upload = [&]() {
      up_err = hipSetDevice(c->device);
      if (up_err == hipSuccess && n_rows)
         up_err = hipMemcpyAsync(d_count, b->count, (size_t)n_rows * sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
      if (up_err == hipSuccess && n_el)
         up_err = hipMemcpyAsync(d_F, b->F, (size_t)n_el * sizeof(double), hipMemcpyHostToDevice, c->stream);
   };
   std::thread uploader;
   try {
      uploader = std::thread(upload);
   } catch (const std::system_error &) { // no thread to be had: upload here, then plan
      upload();
   }
   int rc = sbgpu_plan_create(c, b->n_loci, b->row_off, b->iso_off, b->f_off, &p);
   if (uploader.joinable()) uploader.join();
   if (rc != SBGPU_OK) {
      const std::string why = g_err; // cleanup() must not lose the planner's message
      (void)hipStreamSynchronize(c->stream);
      cleanup();
      return fail(rc, why);
   }
   TRY_CLEAN(up_err);
   rc = sbgpu_em_run_device(c, p, d_count, d_F, d_theta, d_status, d_iters, c->stream);
   if (rc != SBGPU_OK) {
      cleanup();
      return rc;
   }
   TRY_CLEAN(hipMemcpyAsync(theta_out, d_theta, (size_t)n_iso * sizeof(double), hipMemcpyDeviceToHost, c->stream));
   TRY_CLEAN(hipMemcpyAsync(status_out, d_status, (size_t)b->n_loci * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
   if (iters_out)
      TRY_CLEAN(hipMemcpyAsync(iters_out, d_iters, (size_t)b->n_loci * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
   TRY_CLEAN(hipStreamSynchronize(c->stream));
#undef TRY_CLEAN
   cleanup();
   if (c->wide_error && *c->wide_error) {
      *c->wide_error = 0;
      return fail(SBGPU_EHIP, "sbgpu_em_batch: a barrier of the wide-locus EM kernel timed out");
   }
   return SBGPU_OK;
}

int sbgpu_abundance_device(sbgpu_ctx_t *c, const sbgpu_plan_t *p, const double *d_theta,
                           const int32_t *d_status, const int32_t *d_length,
                           const sbgpu_abundance_params_t *params, double *d_fpkm, double *d_frac,
                           int32_t *d_keep, double *d_sum_fpkm, void *stream)
{
   if (!c || !p || !params) return fail(SBGPU_EINVAL, "sbgpu_abundance_device: null argument");
   if (p->host.n_loci == 0) return SBGPU_OK;
   if (!d_theta || !d_status || !d_length || !d_fpkm || !d_frac || !d_keep || !d_sum_fpkm)
      return fail(SBGPU_EINVAL, "sbgpu_abundance_device: null device pointer");
   if (params->total_mapped_reads <= 0) return fail(SBGPU_EINVAL, "sbgpu_abundance_device: total_mapped_reads must be > 0");
   hipStream_t s = (hipStream_t)stream;
   const int64_t n = p->host.n_loci;
   const int threads = 256;
   const int64_t blocks = (n + threads - 1) / threads;
   hipLaunchKernelGGL(abundance_kernel, dim3((unsigned)blocks), dim3(threads), 0, s, n, p->d_iso_off, d_theta, d_status,
                      d_length, *params, d_fpkm, d_frac, d_keep, p->d_locus_sum, p->d_epi_ticket, d_sum_fpkm);
   HIP_TRY(hipGetLastError());
   return SBGPU_OK;
}

int sbgpu_tpm_device(sbgpu_ctx_t *c, int64_t n_iso, const double *d_fpkm, const int32_t *d_keep,
                     const double *d_total_fpkm, double *d_tpm, void *stream)
{
   if (!c) return fail(SBGPU_EINVAL, "sbgpu_tpm_device: null ctx");
   if (n_iso == 0) return SBGPU_OK;
   if (!d_fpkm || !d_keep || !d_total_fpkm || !d_tpm) return fail(SBGPU_EINVAL, "sbgpu_tpm_device: null device pointer");
   hipStream_t s = (hipStream_t)stream;
   const int threads = 256;
   hipLaunchKernelGGL(tpm_kernel, dim3((unsigned)((n_iso + threads - 1) / threads)), dim3(threads), 0, s, n_iso,
                      d_fpkm, d_keep, d_total_fpkm, d_tpm);
   HIP_TRY(hipGetLastError());
   return SBGPU_OK;
}

// ---------------------------------------------------------------- bin-weight model
int sbgpu_insert_pdf_table(const sbgpu_insert_t *ins, int32_t n, double *pdf_out)
{
   if (!ins || (n > 0 && !pdf_out)) return fail(SBGPU_EINVAL, "sbgpu_insert_pdf_table: null argument");
   if (ins->use_emp && (!ins->emp_hist || ins->end_offset < ins->start_offset || ins->total_reads <= 0))
      return fail(SBGPU_EINVAL, "sbgpu_insert_pdf_table: malformed empirical insert-size law");
   // InsertSize::emp_dist_pdf, /root/reference/src/read.cpp:274-297; normal_pdf include/common.h:92-99
   const double inv_sqrt_2pi = 0.3989422804014327;
   for (int32_t fl = 0; fl < n; ++fl) {
      double ret = 0.0;
      if (ins->use_emp && fl >= ins->start_offset && fl <= ins->end_offset)
         ret = ins->emp_hist[fl - ins->start_offset] / ins->total_reads;
      if (ret == 0.0) {
         const double a = ((double)fl - ins->mean) / ins->sd;
         const double p = inv_sqrt_2pi / ins->sd * std::exp(-0.5 * a * a);
         // the reference runs with FTZ/DAZ (-Ofast): a subnormal density is 0 there
         ret = (p >= 2.2250738585072014e-308) ? p : 0.0;
      }
      pdf_out[fl] = ret;
   }
   return SBGPU_OK;
}

int sbgpu_binweight_device(sbgpu_ctx_t *c, int64_t n_pairs, const int64_t *d_seg_off, const uint32_t *d_seg_lens,
                           const uint32_t *d_implicit_mask, const int32_t *d_iso_len, const int64_t *d_out_index,
                           const double *d_pdf, int32_t pdf_len, int32_t read_len, int32_t lmin_base,
                           int32_t long_read, double *d_F, void *stream)
{
   if (!c) return fail(SBGPU_EINVAL, "sbgpu_binweight_device: null ctx");
   if (n_pairs == 0) return SBGPU_OK;
   if (n_pairs < 0 || n_pairs > INT32_MAX) return fail(SBGPU_EINVAL, "sbgpu_binweight_device: bad n_pairs");
   if (!d_seg_off || !d_seg_lens || !d_implicit_mask || !d_iso_len || !d_F || (!long_read && !d_pdf))
      return fail(SBGPU_EINVAL, "sbgpu_binweight_device: null device pointer");
   hipStream_t s = (hipStream_t)stream;
   sb::BinWeightArgs a;
   a.n_pairs = n_pairs;
   a.seg_off = d_seg_off;
   a.seg_lens = d_seg_lens;
   a.implicit_mask = d_implicit_mask;
   a.iso_len = d_iso_len;
   a.out_index = d_out_index;
   a.pdf = d_pdf;
   a.out = d_F;
   a.pdf_len = pdf_len;
   a.pdf_support = c->d_pdf_support;
   if (!long_read) {
      hipLaunchKernelGGL(sb::pdf_support_kernel, dim3(1), dim3(256), 0, s, d_pdf, pdf_len, c->d_pdf_support);
      HIP_TRY(hipGetLastError());
   }
   a.read_len = read_len;
   a.lmin_base = lmin_base;
   a.long_read = long_read;
   // one wave per pair; enough waves to fill the chip several times over
   const int64_t n_batches = (n_pairs + 63) / 64; // a wave serves batches of 64 pairs
   const int64_t want = n_batches < (int64_t)c->n_cu * 64 ? n_batches : (int64_t)c->n_cu * 64;
   hipLaunchKernelGGL(sb::binweight_kernel, dim3((unsigned)want), dim3(64), 0, s, a);
   HIP_TRY(hipGetLastError());
   return SBGPU_OK;
}

int sbgpu_binweight_host(sbgpu_ctx_t *c, int64_t n_pairs, const int64_t *seg_off, const uint32_t *seg_lens,
                         const uint32_t *implicit_mask, const int32_t *iso_len, const sbgpu_insert_t *ins,
                         double *weight_out)
{
   if (!c || !ins) return fail(SBGPU_EINVAL, "sbgpu_binweight_host: null argument");
   if (n_pairs == 0) return SBGPU_OK;
   if (!seg_off || !seg_lens || !implicit_mask || !iso_len || !weight_out)
      return fail(SBGPU_EINVAL, "sbgpu_binweight_host: null argument");
   int64_t max_l = 1;
   for (int64_t p = 0; p < n_pairs; ++p) {
      const int64_t ns = seg_off[p + 1] - seg_off[p];
      if (ins->long_read) continue; // F = 1/L: the segments are not looked at (estimate.cpp:236-247)
      if (ns < 1 || ns > sb::kBinWeightMaxSeg)
         return fail(SBGPU_ESHAPE, "sbgpu_binweight_host: a pair needs 1..32 segments");
      int64_t l = 0;
      for (int64_t k = seg_off[p]; k < seg_off[p + 1]; ++k) l += seg_lens[k];
      if (l > max_l) max_l = l;
   }
   if (max_l > (1 << 26)) return fail(SBGPU_ESHAPE, "sbgpu_binweight_host: segment lengths out of range");
   const int32_t pdf_len = (int32_t)max_l + 1;
   std::vector<double> pdf((size_t)pdf_len);
   int rc = sbgpu_insert_pdf_table(ins, pdf_len, pdf.data());
   if (rc != SBGPU_OK) return rc;
   const int64_t nseg_total = seg_off[n_pairs];
   int64_t *d_off = nullptr;
   uint32_t *d_seg = nullptr, *d_mask = nullptr;
   int32_t *d_len = nullptr;
   double *d_pdf = nullptr, *d_out = nullptr;
   auto cleanup = [&]() {
      (void)hipFree(d_off);
      (void)hipFree(d_seg);
      (void)hipFree(d_mask);
      (void)hipFree(d_len);
      (void)hipFree(d_pdf);
      (void)hipFree(d_out);
   };
#define TRY_CLEAN(expr)                                                                       \
   do {                                                                                       \
      hipError_t e_ = (expr);                                                                 \
      if (e_ != hipSuccess) {                                                                 \
         cleanup();                                                                           \
         return fail(e_ == hipErrorOutOfMemory ? SBGPU_ENOMEM : SBGPU_EHIP,                   \
                     std::string(#expr) + ": " + hipGetErrorString(e_));                      \
      }                                                                                       \
   } while (0)
   TRY_CLEAN(hipMalloc(&d_off, (size_t)(n_pairs + 1) * sizeof(int64_t)));
   TRY_CLEAN(hipMalloc(&d_seg, (size_t)(nseg_total + 1) * sizeof(uint32_t)));
   TRY_CLEAN(hipMalloc(&d_mask, (size_t)n_pairs * sizeof(uint32_t)));
   TRY_CLEAN(hipMalloc(&d_len, (size_t)n_pairs * sizeof(int32_t)));
   TRY_CLEAN(hipMalloc(&d_pdf, (size_t)pdf_len * sizeof(double)));
   TRY_CLEAN(hipMalloc(&d_out, (size_t)n_pairs * sizeof(double)));
   TRY_CLEAN(hipMemcpyAsync(d_off, seg_off, (size_t)(n_pairs + 1) * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
   TRY_CLEAN(hipMemcpyAsync(d_seg, seg_lens, (size_t)nseg_total * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
   TRY_CLEAN(hipMemcpyAsync(d_mask, implicit_mask, (size_t)n_pairs * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
   TRY_CLEAN(hipMemcpyAsync(d_len, iso_len, (size_t)n_pairs * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
   TRY_CLEAN(hipMemcpyAsync(d_pdf, pdf.data(), (size_t)pdf_len * sizeof(double), hipMemcpyHostToDevice, c->stream));
   const int32_t lmin_base = ins->use_emp ? ins->start_offset : ins->read_len;
   rc = sbgpu_binweight_device(c, n_pairs, d_off, d_seg, d_mask, d_len, nullptr, d_pdf, pdf_len, ins->read_len,
                               lmin_base, ins->long_read, d_out, c->stream);
   if (rc != SBGPU_OK) {
      cleanup();
      return rc;
   }
   TRY_CLEAN(hipMemcpyAsync(weight_out, d_out, (size_t)n_pairs * sizeof(double), hipMemcpyDeviceToHost, c->stream));
   TRY_CLEAN(hipStreamSynchronize(c->stream));
#undef TRY_CLEAN
   cleanup();
   return SBGPU_OK;
}

// ---------------------------------------------------------------- output formatting (A9)
int sbgpu_format_value(double v, char out[12])
{
   if (!out) return fail(SBGPU_EINVAL, "sbgpu_format_value: null out");
   char full[400];
   std::snprintf(full, sizeof(full), "%f", v); // std::to_string(double)
   std::strncpy(out, full, 12);                // print2gtf: strncpy into char[12] ...
   out[11] = 0;                                // ... and terminate: first 11 characters
   return SBGPU_OK;
}

int sbgpu_format_gtf_transcript(char *buf, int cap, const char *chrom, char strand, const char *gene_id,
                                const char *transcript_id, const char *ref_gene_id, const char *ref_gene_name,
                                int n_exons, const int32_t *exon_left, const int32_t *exon_right, double fpkm,
                                double frac, double tpm, int32_t keep)
{
   if (!buf || cap < 1 || !chrom || !gene_id || !transcript_id || n_exons < 0 || (n_exons && (!exon_left || !exon_right)))
      return fail(SBGPU_EINVAL, "sbgpu_format_gtf_transcript: bad argument");
   char f[12], r[12], t[12];
   sbgpu_format_value(fpkm, f);
   sbgpu_format_value(frac, r);
   sbgpu_format_value(tpm, t);
   if (keep == 2) { // negative effective length: the strings are "NA" (estimate.cpp:321,339)
      std::strcpy(f, "NA");
      std::strcpy(r, "NA");
   }
   std::string attr = std::string("gene_id \"") + gene_id + "\";transcript_id \"" + transcript_id + "\";";
   if (ref_gene_id && *ref_gene_id) attr += std::string("ref_gene_id \"") + ref_gene_id + "\";";
   if (ref_gene_name && *ref_gene_name) attr += std::string("ref_gene_name \"") + ref_gene_name + "\";";
   attr += std::string("FPKM \"") + f + "\";Frac \"" + r + "\";TPM \"" + t + "\";";
   const int left = n_exons ? exon_left[0] : 0, right = n_exons ? exon_right[n_exons - 1] : 0;
   std::string out;
   char line[256];
   std::snprintf(line, sizeof(line), "%s\tStrawberry\ttranscript\t%d\t%d\t%d\t%c\t%c\t", chrom, left, right, 1000,
                 strand, '.');
   out += line;
   out += attr + "\n";
   for (int k = 0; k < n_exons; ++k) {
      std::snprintf(line, sizeof(line), "%s\tStrawberry\texon\t%d\t%d\t%d\t%c\t%c\t", chrom, exon_left[k],
                    exon_right[k], 1000, strand, '.');
      out += line;
      out += attr + " exon_id \"" + std::to_string(k + 1) + "\";\n";
   }
   std::snprintf(buf, (size_t)cap, "%s", out.c_str());
   return (int)out.size();
}

static int context_row(const char *who, char *buf, int cap, const char *sample, int32_t sample_frag_count, const char *gene_id,
                       uint32_t gene_frag_count, int n_iso, const char *const *transcript_ids, const double *fpkm,
                       const double *cond_prob, const double *frac, int n_seg, const uint32_t *seg_left,
                       const uint32_t *seg_right, uint32_t path_count, const double *seq_stats, uint32_t flags)
{
   if (!buf || cap < 1 || !sample || !gene_id || n_iso < 1 || !transcript_ids || !fpkm || !cond_prob || !frac || n_seg < 0 ||
       (n_seg && (!seg_left || !seg_right)))
      return fail(SBGPU_EINVAL, std::string(who) + ": bad argument");
   std::string out = std::string(sample) + "\t" + std::to_string(sample_frag_count) + "\t" + gene_id + "\t" +
                     std::to_string(gene_frag_count) + "\t";
   char num[64];
   for (int j = 0; j < n_iso; ++j) out += std::string(j ? "," : "") + (transcript_ids[j] ? transcript_ids[j] : "");
   out += "\t";
   for (int j = 0; j < n_iso; ++j) out += std::string(j ? "," : "") + std::to_string(fpkm[j]);
   out += "\t";
   for (int j = 0; j < n_iso; ++j) {
      std::snprintf(num, sizeof(num), "%.12g", cond_prob[j]); // ostream << setprecision(12)
      out += std::string(j ? "," : "") + num;
   }
   out += "\t";
   for (int j = 0; j < n_iso; ++j) out += std::string(j ? "," : "") + std::to_string(frac[j]);
   out += "\t";
   for (int k = 0; k < n_seg; ++k) out += "[" + std::to_string(seg_left[k]) + "-" + std::to_string(seg_right[k]) + "]";
   out += "\t" + std::to_string(path_count);
   if (seq_stats) { // src/alignments.cpp:1630-1635
      out += "\t" + std::to_string(seq_stats[0]) + "\t" + std::to_string(seq_stats[1]);
      for (int q = 0; q < 4; ++q) out += "\t" + std::to_string((bool)((flags >> q) & 1u));
   }
   out += "\n";
   std::snprintf(buf, (size_t)cap, "%s", out.c_str());
   return (int)out.size();
}

int sbgpu_format_context_row(char *buf, int cap, const char *sample, int32_t sample_frag_count, const char *gene_id,
                             uint32_t gene_frag_count, int n_iso, const char *const *transcript_ids, const double *fpkm,
                             const double *cond_prob, const double *frac, int n_seg, const uint32_t *seg_left,
                             const uint32_t *seg_right, uint32_t path_count)
{
   return context_row("sbgpu_format_context_row", buf, cap, sample, sample_frag_count, gene_id, gene_frag_count, n_iso,
                      transcript_ids, fpkm, cond_prob, frac, n_seg, seg_left, seg_right, path_count, nullptr, 0);
}

int sbgpu_format_context_row_seq(char *buf, int cap, const char *sample, int32_t sample_frag_count, const char *gene_id,
                                 uint32_t gene_frag_count, int n_iso, const char *const *transcript_ids, const double *fpkm,
                                 const double *cond_prob, const double *frac, int n_seg, const uint32_t *seg_left,
                                 const uint32_t *seg_right, uint32_t path_count, double gc, double entropy, uint32_t flags)
{
   const double st[2] = {gc, entropy};
   return context_row("sbgpu_format_context_row_seq", buf, cap, sample, sample_frag_count, gene_id, gene_frag_count, n_iso,
                      transcript_ids, fpkm, cond_prob, frac, n_seg, seg_left, seg_right, path_count, st, flags);
}

} // extern "C"
