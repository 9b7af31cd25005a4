// Launch interface of the SUPERSEDED split-bf16 kernels kept as bit-identity references for the tests (libnerf_hip_ref.so; not part
// of the product library or of include/nerf_hip.h): the double-buffered 16-point forward the weight-ring forward replaced, the
// 32-point forward of rounds 1-2, and the double-buffered delta chain the ring dgrad replaced.
#pragma once
#include <hip/hip_runtime.h>

namespace nerf {

hipError_t launch_field_dgrad3(const float* packed3, const float* act, const float* d_raw, int n_rays, int S,
                               float* delta, int mode /* 0 x3 chain + fp32 deltas, 1 single-product chain, 2 x3 chain + bf16 deltas */, hipStream_t stream);
hipError_t launch_field_fwd3(const float* packed3, const float* rays, int ray_stride, const float* z_vals,
                             int n_rays, int S, float* raw, float* act, int bf16_save, hipStream_t stream);
hipError_t launch_field_fwd16(const float* packed3, const float* rays, int ray_stride, const float* z_vals,
                              int n_rays, int S, float* raw, float* act, int bf16_save, hipStream_t stream);

}  // namespace nerf
