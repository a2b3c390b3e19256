// strawberry_amd/csrc/api_internal.h -- what the ABI translation units share (not exported
// through include/sbgpu.h): the error slot and the two context fields the launchers need.
#pragma once

#include <hip/hip_runtime.h>

#include <string>

#include "../../include/sbgpu.h"

namespace sb {
int api_fail(int code, const std::string &msg); // records sbgpu_last_error(), returns code
hipStream_t ctx_stream(const sbgpu_ctx_t *ctx); // the context's own stream
int ctx_cu_count(const sbgpu_ctx_t *ctx);
} // namespace sb
