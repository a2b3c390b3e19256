// strawberry_amd/csrc/api_internal.h -- what the ABI translation units share (not exported
// through include/sbgpu.h): the error slot and the two context fields the launchers need.
#pragma once

#include <hip/hip_runtime.h>

#include <functional>
#include <string>
#include <vector>

#include "../../include/sbgpu.h"

namespace sb {
// (bin, isoform) pairs made on the device (bins_device.h): sbgpu_binweight_device's inputs in one arena.
// A bins handle that holds them downloads them only when somebody exports the pair arrays.
struct DevicePairs {
   char *arena = nullptr; // hipMalloc'ed; owned by the handle it is attached to
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
// sbgpu_bins_create_device with the segment lists made beforehand (nullptr: made inside)
int bins_create_device_impl(sbgpu_ctx_t *ctx, const sbgpu_annotation_t *annot, const sbgpu_hits_t *d_hits, const float *d_mass,
                            const int64_t *locus_hit_off, int32_t compat_words, int32_t key_words, const uint32_t *d_compat,
                            const uint32_t *d_key, int64_t *d_hit_bin, void *stream, const IsoSegments *iso_pre, sbgpu_bins_t **out,
                            const uint64_t *d_span, const uint32_t *d_fhash, // spans / hashes of the hits (exonbin_device_impl), or null
                            const std::function<int(const DeviceGrouping &)> *after_pairs = nullptr); // called once the pairs' fill is launched
int exonbin_device_impl(sbgpu_ctx_t *ctx, const sbgpu_annotation_t *annot, const sbgpu_hits_t *hits, int32_t compat_words,
                        int32_t key_words, uint32_t *d_compat, uint32_t *d_key, uint64_t *d_span, uint32_t *d_fhash, void *stream);
int api_fail(int code, const std::string &msg); // records sbgpu_last_error(), returns code
hipStream_t ctx_stream(const sbgpu_ctx_t *ctx); // the context's own stream
int ctx_cu_count(const sbgpu_ctx_t *ctx);
int ctx_device(const sbgpu_ctx_t *ctx);        // the HIP device the context was made on
// device scratch that lives with the context (slot 0..7, grows on demand, never shrinks): valid until the next
// request for the same slot; one host thread per context
hipError_t ctx_scratch(sbgpu_ctx_t *ctx, int slot, size_t bytes, char **out);
// kernel stages of the chain entry points: while sbgpu_set_timing is on, the launches between a begin and its end
// are bracketed by HIP events on their stream (sbgpu_last_stage_ms reads them); no-ops otherwise
void ctx_stage_reset(sbgpu_ctx_t *ctx);
void ctx_stage_begin(sbgpu_ctx_t *ctx, const char *name, hipStream_t s);
void ctx_stage_end(sbgpu_ctx_t *ctx, hipStream_t s);
bool ctx_take_wide_error(sbgpu_ctx_t *ctx);    // true once if a wide-locus barrier timed out since the last call (clears the flag)
// locus_bins.cpp: finish bins that were grouped on the device (host copies of the per-bin arrays)
// `pairs`: made on the device already (the handle takes the arena over); nullptr: make them here, on the host
int bins_from_groups(const sbgpu_annotation_t *annot, int32_t compat_words, int32_t key_words, const int64_t *row_off,
                     const int32_t *count, const uint32_t *key, const uint32_t *compat, int64_t n_hits_used,
                     const DevicePairs *pairs, sbgpu_bins_t **out);
const DevicePairs *bins_device_pairs(const sbgpu_bins_t *bins); // nullptr when the pairs live on the host
void bins_set_weights(sbgpu_bins_t *bins, std::vector<double> &&F);
void bins_set_hit_bin(sbgpu_bins_t *bins, std::vector<int64_t> &&hit_bin);
const double *bins_weights_tail(const sbgpu_bins_t *bins, size_t at); // F.data() + at (the empirical histogram lives there)
} // namespace sb
