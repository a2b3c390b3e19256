// strawberry_amd/csrc/comm_api.hip -- the path's one collective behind the C ABI.
//
// The reference sums two scalars over all loci -- the mapped-fragment total before the EM
// (/root/reference/src/alignments.cpp:1372) and the FPKM total after it (:1821-1824).  With the loci sharded over
// one process per GPU these become all-reduce(sum) of a few bytes: RCCL over xGMI.  librccl.so is opened when the
// first communicator of more than one rank is made (dlopen), so a single-GPU user of libsbgpu.so does not need it.
// A world of one rank needs neither RCCL nor an id: its all-reduce is the identity -- unless SBGPU_COMM_FORCE_RCCL=1 is
// set in the environment, which sends a world of ONE through the same RCCL calls as a world of eight (dlopen, id,
// ncclCommInitRank, ncclAllReduce on the caller's stream): the way to exercise the binding on a one-GPU box.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/sbgpu.h"
#include "api_internal.h"

namespace {

// the slice of the RCCL API this file uses (rccl.h of ROCm 7.2); enum values as published there
typedef struct {
   char internal[128];
} RcclUniqueId;
typedef void *RcclComm;
constexpr int kRcclSum = 0, kRcclMax = 2, kRcclInt64 = 4, kRcclFloat64 = 8;
struct Rccl {
   void *lib = nullptr;
   int (*GetUniqueId)(RcclUniqueId *) = nullptr;
   int (*CommInitRank)(RcclComm *, int, RcclUniqueId, int) = nullptr;
   int (*AllReduce)(const void *, void *, size_t, int, int, RcclComm, hipStream_t) = nullptr;
   int (*CommDestroy)(RcclComm) = nullptr;
   int (*CommCount)(RcclComm, int *) = nullptr;
   const char *(*GetErrorString)(int) = nullptr;
   std::string why; // non-empty: RCCL is unusable, and why
};
Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl &rccl()
{
   std::call_once(g_rccl_once, []() {
      for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
         g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
         if (g_rccl.lib) break;
      }
      if (!g_rccl.lib) {
         const char *e = dlerror();
         g_rccl.why = std::string("dlopen(librccl.so): ") + (e ? e : "not found");
         return;
      }
      auto sym = [&](const char *n) {
         void *p = dlsym(g_rccl.lib, n);
         if (!p && g_rccl.why.empty()) g_rccl.why = std::string("librccl.so lacks ") + n;
         return p;
      };
      g_rccl.GetUniqueId = (int (*)(RcclUniqueId *))sym("ncclGetUniqueId");
      g_rccl.CommInitRank = (int (*)(RcclComm *, int, RcclUniqueId, int))sym("ncclCommInitRank");
      g_rccl.AllReduce = (int (*)(const void *, void *, size_t, int, int, RcclComm, hipStream_t))sym("ncclAllReduce");
      g_rccl.CommDestroy = (int (*)(RcclComm))sym("ncclCommDestroy");
      g_rccl.CommCount = (int (*)(RcclComm, int *))dlsym(g_rccl.lib, "ncclCommCount"); // optional: only sbgpu_comm_rccl_ranks asks
      g_rccl.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
   });
   return g_rccl;
}

int rccl_fail(const char *what, int rc)
{
   const Rccl &r = rccl();
   return sb::api_fail(SBGPU_ERCCL, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error"));
}

} // namespace

struct sbgpu_comm {
   int rank = 0, world = 1, device = 0;
   RcclComm comm = nullptr;
   hipStream_t stream = nullptr; // the context's own stream: the host-buffer forms run on it
   void *d_scratch = nullptr;    // kScratchBytes of device memory for the host-buffer forms
   sbgpu_host_allreduce_fn host_fn = nullptr; // sbgpu_comm_init_host: the exchange is the caller's, on host buffers
   void *host_user = nullptr;
};
namespace {
constexpr size_t kScratchBytes = 4096;
}

extern "C" {

int sbgpu_comm_unique_id(uint8_t id_out[SBGPU_COMM_ID_BYTES])
{
   if (!id_out) return sb::api_fail(SBGPU_EINVAL, "sbgpu_comm_unique_id: null output");
   const Rccl &r = rccl();
   if (!r.why.empty()) return sb::api_fail(SBGPU_ERCCL, "sbgpu_comm_unique_id: " + r.why);
   RcclUniqueId id;
   const int rc = r.GetUniqueId(&id);
   if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
   static_assert(sizeof(id) == SBGPU_COMM_ID_BYTES, "the id is 128 bytes");
   std::memcpy(id_out, &id, sizeof(id));
   return SBGPU_OK;
}

int sbgpu_comm_init(sbgpu_ctx_t *ctx, int rank, int world, const uint8_t id[SBGPU_COMM_ID_BYTES], sbgpu_comm_t **comm_out)
{
   if (!ctx || !comm_out) return sb::api_fail(SBGPU_EINVAL, "sbgpu_comm_init: null argument");
   *comm_out = nullptr;
   if (world < 1 || rank < 0 || rank >= world) return sb::api_fail(SBGPU_EINVAL, "sbgpu_comm_init: need 0 <= rank < world");
   sbgpu_comm *c = new (std::nothrow) sbgpu_comm();
   if (!c) return sb::api_fail(SBGPU_ENOMEM, "sbgpu_comm_init: out of host memory");
   c->rank = rank, c->world = world, c->device = sb::ctx_device(ctx);
   c->stream = sb::ctx_stream(ctx);
   const char *force = std::getenv("SBGPU_COMM_FORCE_RCCL");
   const bool forced = world == 1 && force && force[0] == '1';
   if (world > 1 || forced) {
      if (!id && !forced) {
         delete c;
         return sb::api_fail(SBGPU_EINVAL, "sbgpu_comm_init: world > 1 needs the id rank 0 made with sbgpu_comm_unique_id");
      }
      const Rccl &r = rccl();
      if (!r.why.empty()) {
         delete c;
         return sb::api_fail(SBGPU_ERCCL, "sbgpu_comm_init: " + r.why);
      }
      uint8_t own_id[SBGPU_COMM_ID_BYTES];
      if (!id) { // a forced world of one without an id: it is its own rank 0
         const int rc_id = sbgpu_comm_unique_id(own_id);
         if (rc_id != SBGPU_OK) {
            delete c;
            return rc_id;
         }
         id = own_id;
      }
      hipError_t e = hipSetDevice(c->device);
      if (e != hipSuccess) {
         delete c;
         return sb::api_fail(SBGPU_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
      }
      RcclUniqueId uid;
      std::memcpy(&uid, id, sizeof(uid));
      const int rc = r.CommInitRank(&c->comm, world, uid, rank);
      if (rc != 0) {
         delete c;
         return rccl_fail("ncclCommInitRank", rc);
      }
      e = hipMalloc(&c->d_scratch, kScratchBytes);
      if (e != hipSuccess) {
         (void)r.CommDestroy(c->comm);
         delete c;
         return sb::api_fail(SBGPU_ENOMEM, std::string("hipMalloc(comm scratch): ") + hipGetErrorString(e));
      }
   }
   *comm_out = c;
   return SBGPU_OK;
}

int sbgpu_comm_destroy(sbgpu_comm_t *c)
{
   if (!c) return SBGPU_OK;
   if (c->comm) (void)rccl().CommDestroy(c->comm);
   if (c->d_scratch) {
      (void)hipSetDevice(c->device);
      (void)hipFree(c->d_scratch);
   }
   delete c;
   return SBGPU_OK;
}

int sbgpu_comm_rccl_ranks(const sbgpu_comm_t *c, int *ranks)
{
   if (!c || !ranks) return sb::api_fail(SBGPU_EINVAL, "sbgpu_comm_rccl_ranks: null argument");
   *ranks = 0;
   if (!c->comm) return SBGPU_OK; // a world of one without RCCL
   const Rccl &r = rccl();
   if (!r.CommCount) return sb::api_fail(SBGPU_ERCCL, "sbgpu_comm_rccl_ranks: librccl.so lacks ncclCommCount");
   const int rc = r.CommCount(c->comm, ranks);
   if (rc != 0) return rccl_fail("ncclCommCount", rc);
   return SBGPU_OK;
}

int sbgpu_comm_info(const sbgpu_comm_t *c, int *rank, int *world)
{
   if (!c) return sb::api_fail(SBGPU_EINVAL, "sbgpu_comm_info: null comm");
   if (rank) *rank = c->rank;
   if (world) *world = c->world;
   return SBGPU_OK;
}

static int allreduce(sbgpu_comm_t *c, void *d_buf, int64_t n, int dtype, void *stream, const char *who, int op = kRcclSum)
{
   if (!c) return sb::api_fail(SBGPU_EINVAL, std::string(who) + ": null comm");
   if (n < 0 || (n > 0 && !d_buf)) return sb::api_fail(SBGPU_EINVAL, std::string(who) + ": bad buffer");
   if (c->host_fn && n > 0 && c->world > 1) { // the caller's exchange: through host memory, synchronous
      std::vector<char> h((size_t)n * 8);
      hipError_t e = hipSetDevice(c->device);
      if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d_buf, h.size(), hipMemcpyDeviceToHost, (hipStream_t)stream);
      if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
      if (e != hipSuccess) return sb::api_fail(SBGPU_EHIP, std::string(who) + ": " + hipGetErrorString(e));
      if (c->host_fn(c->host_user, h.data(), n, dtype == kRcclFloat64 ? 1 : 0, op == kRcclMax ? 1 : 0) != 0)
         return sb::api_fail(SBGPU_ERCCL, std::string(who) + ": the caller's all-reduce failed");
      e = hipMemcpyAsync(d_buf, h.data(), h.size(), hipMemcpyHostToDevice, (hipStream_t)stream);
      if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
      if (e != hipSuccess) return sb::api_fail(SBGPU_EHIP, std::string(who) + ": " + hipGetErrorString(e));
      return SBGPU_OK;
   }
   if (!c->comm || n == 0) return SBGPU_OK; // the sum over one rank (a forced world of one has a communicator and goes on)
   const int rc = rccl().AllReduce(d_buf, d_buf, (size_t)n, dtype, op, c->comm, (hipStream_t)stream);
   if (rc != 0) return rccl_fail(who, rc);
   return SBGPU_OK;
}

int sbgpu_allreduce_sum_f64(sbgpu_comm_t *c, double *d_buf, int64_t n, void *stream)
{
   return allreduce(c, d_buf, n, kRcclFloat64, stream, "sbgpu_allreduce_sum_f64");
}

int sbgpu_allreduce_sum_i64(sbgpu_comm_t *c, int64_t *d_buf, int64_t n, void *stream)
{
   return allreduce(c, d_buf, n, kRcclInt64, stream, "sbgpu_allreduce_sum_i64");
}

int sbgpu_allreduce_max_i64(sbgpu_comm_t *c, int64_t *d_buf, int64_t n, void *stream)
{
   return allreduce(c, d_buf, n, kRcclInt64, stream, "sbgpu_allreduce_max_i64", kRcclMax);
}

// host buffers: staged through the communicator's scratch on the context's stream, synchronous
static int allreduce_host(sbgpu_comm_t *c, void *buf, int64_t n, int dtype, const char *who, int op = kRcclSum)
{
   if (!c) return sb::api_fail(SBGPU_EINVAL, std::string(who) + ": null comm");
   if (n < 0 || (n > 0 && !buf) || (size_t)n * 8 > kScratchBytes) return sb::api_fail(SBGPU_EINVAL, std::string(who) + ": 0 <= n <= 512 values");
   if (c->host_fn && n > 0 && c->world > 1) {
      if (c->host_fn(c->host_user, buf, n, dtype == kRcclFloat64 ? 1 : 0, op == kRcclMax ? 1 : 0) != 0)
         return sb::api_fail(SBGPU_ERCCL, std::string(who) + ": the caller's all-reduce failed");
      return SBGPU_OK;
   }
   if (!c->comm || n == 0) return SBGPU_OK;
   hipError_t e = hipSetDevice(c->device);
   if (e == hipSuccess) e = hipMemcpyAsync(c->d_scratch, buf, (size_t)n * 8, hipMemcpyHostToDevice, c->stream);
   if (e != hipSuccess) return sb::api_fail(SBGPU_EHIP, std::string(who) + ": " + hipGetErrorString(e));
   const int rc = allreduce(c, c->d_scratch, n, dtype, c->stream, who, op);
   if (rc != SBGPU_OK) return rc;
   e = hipMemcpyAsync(buf, c->d_scratch, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream);
   if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
   if (e != hipSuccess) return sb::api_fail(SBGPU_EHIP, std::string(who) + ": " + hipGetErrorString(e));
   return SBGPU_OK;
}

int sbgpu_allreduce_sum_f64_host(sbgpu_comm_t *c, double *buf, int64_t n)
{
   return allreduce_host(c, buf, n, kRcclFloat64, "sbgpu_allreduce_sum_f64_host");
}

int sbgpu_allreduce_sum_i64_host(sbgpu_comm_t *c, int64_t *buf, int64_t n)
{
   return allreduce_host(c, buf, n, kRcclInt64, "sbgpu_allreduce_sum_i64_host");
}

int sbgpu_allreduce_max_i64_host(sbgpu_comm_t *c, int64_t *buf, int64_t n)
{
   return allreduce_host(c, buf, n, kRcclInt64, "sbgpu_allreduce_max_i64_host", kRcclMax);
}

int sbgpu_comm_init_host(sbgpu_ctx_t *ctx, int rank, int world, sbgpu_host_allreduce_fn fn, void *user, sbgpu_comm_t **comm_out)
{
   if (!ctx || !comm_out || !fn) return sb::api_fail(SBGPU_EINVAL, "sbgpu_comm_init_host: null argument");
   *comm_out = nullptr;
   if (world < 1 || rank < 0 || rank >= world) return sb::api_fail(SBGPU_EINVAL, "sbgpu_comm_init_host: need 0 <= rank < world");
   sbgpu_comm *c = new (std::nothrow) sbgpu_comm();
   if (!c) return sb::api_fail(SBGPU_ENOMEM, "sbgpu_comm_init_host: out of host memory");
   c->rank = rank, c->world = world, c->device = sb::ctx_device(ctx);
   c->stream = sb::ctx_stream(ctx);
   c->host_fn = fn, c->host_user = user;
   *comm_out = c;
   return SBGPU_OK;
}

} // extern "C"
