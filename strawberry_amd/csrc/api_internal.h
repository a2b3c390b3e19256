// strawberry_amd/csrc/api_internal.h -- what the ABI translation units share (not exported
// through include/sbgpu.h): the error slot and the two context fields the launchers need.
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/sbgpu.h"

namespace sb {
int api_fail(int code, const std::string &msg); // records sbgpu_last_error(), returns code
hipStream_t ctx_stream(const sbgpu_ctx_t *ctx); // the context's own stream
int ctx_cu_count(const sbgpu_ctx_t *ctx);
// locus_bins.cpp: finish bins that were grouped on the device (host copies of the per-bin arrays)
int bins_from_groups(const sbgpu_annotation_t *annot, int32_t compat_words, int32_t key_words, const int64_t *row_off,
                     const int32_t *count, const uint32_t *key, const uint32_t *compat, int64_t n_hits_used,
                     sbgpu_bins_t **out);
void bins_set_weights(sbgpu_bins_t *bins, std::vector<double> &&F);
void bins_set_hit_bin(sbgpu_bins_t *bins, std::vector<int64_t> &&hit_bin);
const double *bins_weights_tail(const sbgpu_bins_t *bins, size_t at); // F.data() + at (the empirical histogram lives there)
} // namespace sb
