// extern "C" entry points of libnerf_hip_ref.so: the superseded split-bf16 kernels as TEST references (tests/test_gpu_round3.py:
// the weight-ring forward and dgrad are bit-identical to them).  Plain launches, no buffer records; packed3 = the product
// library's bf16 repack (nerf_pack_params_split(streams = 15, split = 0)).
#include <hip/hip_runtime.h>
#include "ref_launchers.h"

extern "C" {

// field_fwd16_kernel<SAVE>: the double-buffered 16-point forward; act NULL = inference, bf16_save = 1: bf16 rows (the ring kernel's layout)
int nerf_ref_field_fwd16(const float* packed3, const float* rays, int ray_stride, const float* z_vals, int n_rays, int n_samples,
                         float* raw, float* act, int bf16_save, void* stream) {
    if (!packed3 || !rays || !z_vals || !raw || n_rays < 0 || n_samples < 1) return -1;
    return (int)nerf::launch_field_fwd16(packed3, rays, ray_stride, z_vals, n_rays, n_samples, raw, act, bf16_save, (hipStream_t)stream);
}
// field_dgrad3_kernel<MODE>: the double-buffered delta chain; mode 2 = three-term chain writing bf16 deltas (the ring kernel's layout)
int nerf_ref_field_dgrad3(const float* packed3, const float* act, const float* d_raw, int n_rays, int n_samples, float* delta, int mode,
                          void* stream) {
    if (!packed3 || !act || !d_raw || !delta || n_rays < 0 || n_samples < 1 || mode < 0 || mode > 2) return -1;
    return (int)nerf::launch_field_dgrad3(packed3, act, d_raw, n_rays, n_samples, delta, mode, (hipStream_t)stream);
}

}  // extern "C"
