// strawberry_amd/csrc/api_internal.h -- what the ABI translation units share (not exported
// through include/sbgpu.h): the error slot and the two context fields the launchers need.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "../../include/sbgpu.h"

namespace sb {
// A vector whose resize() leaves new elements uninitialised: the targets of device-to-host copies of tens of MB,
// where the zero fill of std::vector costs as much as the copy.
template <class T>
struct DefaultInitAlloc : std::allocator<T> {
   template <class U>
   struct rebind {
      using other = DefaultInitAlloc<U>;
   };
   using std::allocator<T>::allocator;
   template <class U>
   void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value)
   {
      ::new ((void *)p) U;
   }
   template <class U, class... A>
   void construct(U *p, A &&...a)
   {
      ::new ((void *)p) U(std::forward<A>(a)...);
   }
};
template <class T>
using PodVec = std::vector<T, DefaultInitAlloc<T>>;

// (bin, isoform) pairs made on the device (bins_device.h): sbgpu_binweight_device's inputs in one arena.
// A bins handle that holds them downloads them only when somebody exports the pair arrays.
struct DevicePairs {
   char *arena = nullptr; // sb::dev_take'n; owned by the handle it is attached to
   size_t capacity = 0;
   int64_t n_pairs = 0, n_pair_segs = 0;
   bool any_wide = false; // some pair spans more than 32 segments and carries none (fine for long reads only)
   size_t o_seg_off = 0, o_seg_lens = 0, o_mask = 0, o_iso_len = 0, o_out_index = 0;
   const int64_t *seg_off() const { return (const int64_t *)(arena + o_seg_off); }
   const uint32_t *seg_lens() const { return (const uint32_t *)(arena + o_seg_lens); }
   const uint32_t *mask() const { return (const uint32_t *)(arena + o_mask); }
   const int32_t *iso_len() const { return (const int32_t *)(arena + o_iso_len); }
   const int64_t *out_index() const { return (const int64_t *)(arena + o_out_index); }
};
// exonbin_api.hip: the isoforms' segment lists of an annotation (host data; depends on the annotation only)
struct IsoSegments {
   std::vector<int64_t> seg_off;           // [n_iso + 1]
   std::vector<int32_t> seg_idx, locus, len; // segment indices within the locus; the isoform's locus; its exonic length
};
void iso_segments(const sbgpu_annotation_t *annot, IsoSegments *out);
// What the device grouping knows once its last kernel (the pairs' fill) is in the stream: enough to launch the bin
// weights and the EM behind it, BEFORE the host-side handle is built (which then runs beside those kernels).
struct DeviceGrouping {
   const int64_t *row_off, *f_off; // host, [n_loci + 1]
   int64_t n_bins, n_elem;
   const int32_t *d_count;         // device, [n_bins]: the bins' fragment counts
   const DevicePairs *pairs;       // device arrays of the (bin, isoform) pairs
};
// The isoforms' segment lists (IsoSegments) in device memory
struct DeviceIsoSegments {
   const int64_t *seg_off = nullptr;
   const int32_t *seg_idx = nullptr, *locus = nullptr, *len = nullptr;
};
// The isoforms in the segment basis (exonbin_device.h: iso_masks_kernel): made per call, or once for a resident annotation
struct DeviceSegBasis {
   const uint64_t *member = nullptr, *start = nullptr, *adj = nullptr;
   const uint64_t *member_hi = nullptr, *start_hi = nullptr, *adj_hi = nullptr; // bits 64-127 (loci of 65-128 segments / isoforms)
   const uint32_t *ok = nullptr;
};
size_t seg_basis_bytes(int64_t n_loci, int64_t n_iso);
// launches iso_masks_kernel on `stream` over a device annotation, into `mem` (seg_basis_bytes): -> the pointers
int make_seg_basis(sbgpu_ctx_t *ctx, const sbgpu_annotation_t *d_annot, int64_t n_iso, char *mem, void *stream, DeviceSegBasis *out);
// An annotation kept resident (sbgpu_annotation_pin, chain_api.hip): the caller's arrays remembered by address, their
// device copies, and what the chain makes of an annotation alone.  Lives with the context.
struct ResidentAnnotation {
   sbgpu_annotation_t key;  // the caller's struct as pinned
   char *arena = nullptr;   // device (sb::dev_take)
   size_t capacity = 0;
   sbgpu_annotation_t dev;  // the arrays' device copies
   DeviceIsoSegments d_iso;
   DeviceSegBasis seg_basis; // (same arena)
   IsoSegments iso;         // host
   int64_t max_locus_span = 1, max_iso = 1, max_seg = 1; // longest locus' segments together; widest locus
   uint64_t print = 0;      // fingerprint() of the caller's arrays when they were pinned
   // A cheap content check beside the addresses: the totals, and up to 256 evenly spaced entries of every array (FNV-1a).
   // The caller's arrays may have been freed and their addresses recycled by another annotation of the same shape since the
   // pin; that one does not match (it would otherwise be served the stale device copies and tables -- wrong theta, no error).
   static uint64_t fingerprint(const sbgpu_annotation_t *a)
   {
      uint64_t h = 1469598103934665603ull;
      auto mix = [&h](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
      const int64_t nl = a->n_loci, n_iso = a->iso_off[nl], n_exon = a->exon_off[n_iso], n_seg = a->seg_off[nl];
      mix((uint64_t)nl), mix((uint64_t)n_iso), mix((uint64_t)n_exon), mix((uint64_t)n_seg);
      auto sample = [&mix](auto *p, int64_t n) {
         if (!p || n <= 0) return;
         const int64_t step = n > 256 ? n / 256 : 1;
         for (int64_t k = 0; k < n; k += step) mix((uint64_t)p[k]);
         mix((uint64_t)p[n - 1]);
      };
      sample(a->iso_off, nl + 1), sample(a->exon_off, n_iso + 1), sample(a->seg_off, nl + 1);
      sample(a->exon_left, n_exon), sample(a->exon_right, n_exon), sample(a->seg_left, n_seg), sample(a->seg_right, n_seg);
      return h;
   }
   // the pin was made for these arrays (addresses and counts only: the arrays may be gone already -- an owner releasing
   // ITS pin must not read them, and must not release a pin that another annotation has replaced it with)
   bool same_arrays(const sbgpu_annotation_t *a) const
   {
      return a && a->n_loci == key.n_loci && a->iso_off == key.iso_off && a->exon_off == key.exon_off && a->exon_left == key.exon_left &&
             a->exon_right == key.exon_right && a->seg_off == key.seg_off && a->seg_left == key.seg_left && a->seg_right == key.seg_right;
   }
   bool matches(const sbgpu_annotation_t *a) const { return same_arrays(a) && fingerprint(a) == print; }
};
const ResidentAnnotation *ctx_resident_annotation(const sbgpu_ctx_t *ctx);
void ctx_set_resident_annotation(sbgpu_ctx_t *ctx, ResidentAnnotation *r); // takes ownership; frees the one before (nullptr: just that)
// What a caller that runs the grouping as one stage of a longer stream (chain_api.hip) hands in: all optional.
struct GroupingHooks {
   const sbgpu_annotation_t *d_annot = nullptr; // iso_off, seg_off, seg_left, seg_right of `annot` in device memory already
   const DeviceIsoSegments *d_iso = nullptr;    // the segment lists too (then `iso_pre` must be given as well)
   // called as soon as the loci's bin counts are on the host (row_off, f_off: [n_loci + 1]), while the middle kernels run
   std::function<void(const int64_t *row_off, const int64_t *f_off)> rows_known;
   // called once the pairs' fill kernel is launched, before the host-side handle is built
   std::function<int(const DeviceGrouping &)> after_pairs;
};
// sbgpu_bins_create_device with the segment lists made beforehand (nullptr: made inside)
int bins_create_device_impl(sbgpu_ctx_t *ctx, const sbgpu_annotation_t *annot, const sbgpu_hits_t *d_hits, const float *d_mass,
                            const int64_t *locus_hit_off, int32_t compat_words, int32_t key_words, const uint32_t *d_compat,
                            const uint32_t *d_key, int64_t *d_hit_bin, void *stream, const IsoSegments *iso_pre, sbgpu_bins_t **out,
                            const uint64_t *d_span, const uint32_t *d_fhash, // spans / hashes of the hits (exonbin_device_impl), or null
                            const GroupingHooks *hooks = nullptr);
// n_iso: the annotation's isoform count where the caller knows it on the host (-1: read from the device)
int exonbin_device_impl(sbgpu_ctx_t *ctx, const sbgpu_annotation_t *annot, const sbgpu_hits_t *hits, int32_t compat_words,
                        int32_t key_words, uint32_t *d_compat, uint32_t *d_key, uint64_t *d_span, uint32_t *d_fhash, void *stream,
                        int64_t n_iso = -1, const DeviceSegBasis *seg_basis = nullptr);
// bamdecode_api.hip: the device array `record` of a device handle (accepted read -> its record's index)
const int64_t *bamreads_device_record(const sbgpu_bamreads_t *reads);
int api_fail(int code, const std::string &msg); // records sbgpu_last_error(), returns code
// Experiment switches: SBGPU_* variables that select paths DESIGN.md / profiles/EXPERIMENTS_*.md record as measured and NOT adopted
// (phased execution, launch graphs, tile and schedule variants, A/B forms of the grouping ...).  They exist only in a library built
// with -DSB_EXPERIMENTS (`make experiments` -> libsbgpu_exp.so, never shipped, never timed by bench.py); the shipped library does
// not read them.  What stays a plain getenv: resources (SBGPU_POOL_GB, SBGPU_HOST_THREADS), diagnostics (SBGPU_HOST_TIMING) and the
// tests' hooks that force a fallback route the default inputs do not take (SBGPU_PAIR_FORCE_SORT, SBGPU_COLLAPSE_FORCE_SEQ,
// SBGPU_COLLAPSE_TWO_SORTS, SBGPU_BAM_TWO_PASS, SBGPU_NO_WIDE, SBGPU_COMM_FORCE_RCCL).
inline const char *exp_env(const char *name)
{
#ifdef SB_EXPERIMENTS
   return std::getenv(name);
#else
   (void)name;
   return nullptr;
#endif
}
hipStream_t ctx_stream(const sbgpu_ctx_t *ctx); // the context's own stream
hipStream_t ctx_aux_stream(const sbgpu_ctx_t *ctx, int i); // one of the context's side streams (0..7; the EM's kinds use 0, 1, 2, 6)
int ctx_cu_count(const sbgpu_ctx_t *ctx);
int ctx_device(const sbgpu_ctx_t *ctx);        // the HIP device the context was made on
// device scratch that lives with the context (slot 0..7, grows on demand, never shrinks): valid until the next
// request for the same slot; one host thread per context
hipError_t ctx_scratch(sbgpu_ctx_t *ctx, int slot, size_t bytes, char **out);
// Device allocations that change hands (a handle's arenas, a plan's arena): a hipMalloc / hipFree pair per call costs a
// few hundred microseconds -- seconds for the tens of GB of a sample-sized call -- and the free waits for the device, so
// blocks handed back are kept -- process-wide, per device, at most 24 blocks and half of the device's memory
// (SBGPU_POOL_GB) -- and a request is served from them when one fits without wasting more than half;
// dev_release_idle (sbgpu_release_idle_memory) lets the idle blocks go.
// dev_give waits for the device first, as hipFree does (a handle may be destroyed while a kernel on the caller's stream
// still reads its arena).
hipError_t dev_take(size_t bytes, char **out, size_t *capacity);
size_t dev_release_idle();
void dev_give(char *block, size_t capacity);
// While one of these lives in a thread, its dev_give calls do NOT wait for the device: for a caller whose blocks were only ever
// used on streams it has synchronised itself, and that runs other work -- an upload on another stream -- it must not wait for
// (sbgpu_front_stream_*: every stage's handle goes back to the pool while the next chunk is on its way).
struct DevGiveStreamSynced {
   DevGiveStreamSynced();
   ~DevGiveStreamSynced();
   DevGiveStreamSynced(const DevGiveStreamSynced &) = delete;
   DevGiveStreamSynced &operator=(const DevGiveStreamSynced &) = delete;
};
// pinned host scratch of the same kind (slot 0..3): the targets of small device-to-host copies that must not block the host
hipError_t ctx_pinned(sbgpu_ctx_t *ctx, int slot, size_t bytes, char **out);
// a second stream for copies that run beside the context's kernels, and events (0..5) to order the two (made on first use)
hipError_t ctx_copy_stream(sbgpu_ctx_t *ctx, hipStream_t *out);
hipError_t ctx_event(sbgpu_ctx_t *ctx, int which, hipEvent_t *out);
// sizes of the last device grouping's pair arena (bytes): a hint that lets the next one allocate before it knows its own
size_t ctx_pairs_hint(const sbgpu_ctx_t *ctx);
void ctx_set_pairs_hint(sbgpu_ctx_t *ctx, size_t bytes);
// kernel stages of the chain entry points: while sbgpu_set_timing is on, the launches between a begin and its end
// are bracketed by HIP events on their stream (sbgpu_last_stage_ms reads them); no-ops otherwise
void ctx_stage_reset(sbgpu_ctx_t *ctx);
void ctx_stage_begin(sbgpu_ctx_t *ctx, const char *name, hipStream_t s);
void ctx_stage_end(sbgpu_ctx_t *ctx, hipStream_t s);
bool ctx_take_wide_error(sbgpu_ctx_t *ctx);    // true once if a wide-locus barrier timed out since the last call (clears the flag)
// locus_bins.cpp: finish bins that were grouped on the device (host copies of the per-bin arrays)
// `pairs`: made on the device already (the handle takes the arena over); nullptr: make them here, on the host
// The per-bin arrays of a device grouping -- count, key words, compat words, 12+ bytes per bin -- stay where the pack
// kernel wrote them: one device arena the handle takes over, downloaded on the first export that asks for them (like
// the pairs).  A caller that only wants abundances never pays the 17 MB (1.4 M bins) over PCIe.
struct DeviceBinArrays {
   char *arena = nullptr; // sb::dev_take'n; owned by the handle it is attached to
   size_t capacity = 0;
   size_t o_count = 0, o_key = 0, o_compat = 0;
};
// `iso_len`: the isoforms' exonic lengths where the caller has them (IsoSegments::len)
int bins_from_groups(const sbgpu_annotation_t *annot, int32_t compat_words, int32_t key_words, const int64_t *row_off,
                     const DeviceBinArrays &arrays, int64_t n_hits_used, const DevicePairs *pairs, const std::vector<int32_t> *iso_len,
                     sbgpu_bins_t **out);
const DevicePairs *bins_device_pairs(const sbgpu_bins_t *bins); // nullptr when the pairs live on the host
void bins_set_weights(sbgpu_bins_t *bins, std::vector<double> &&F);
void bins_set_grouping(sbgpu_bins_t *bins, bool on_device, const std::string &why_host); // what sbgpu_bins_grouping reports
void bins_set_hit_bin(sbgpu_bins_t *bins, std::vector<int64_t> &&hit_bin);
void bins_set_device_hit_bin(sbgpu_bins_t *bins, char *arena, size_t capacity, int64_t n); // hit -> bin left in HBM (the handle owns the arena)
const double *bins_weights_tail(const sbgpu_bins_t *bins, size_t at); // F.data() + at (the empirical histogram lives there)
} // namespace sb
