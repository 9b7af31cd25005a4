// field_fwd16r_kernel: the 16-point-per-wave split-bf16 forward (field_fwd16_kernel of field_fwd_bf16.hip: encode + 8x256
// trunk + density head + folded view branch -> raw[P,4]; run_nerf.py:37-51, run_nerf_helpers.py:15-45, :96-119) on the
// weight RING of field_ring.h instead of the double-buffered weight stream.  Same fragment stream (P16F), same MFMA order
// per accumulator, same encodings and heads: `raw`, the saved rows, encodings and ReLU bitmasks are BIT-IDENTICAL to
// field_fwd16_kernel<SAVE> (tests/test_gpu_parity.py::test_ring_forward_bit_identical); what changes is when the weights
// arrive and when the fragments are requested.  SAVE: 0 = inference, 2 = bf16 rows (the default operand storage of the
// weight-gradient GEMM); fp32 rows (SAVE 1) stay on field_fwd16_kernel<1>.
#include "field_fwd_ring_body.h"

namespace nerf {

template <int SAVE>
__global__ __launch_bounds__(FIELD_WAVES * 64) void field_fwd16r_kernel(FieldFwdRingArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    field_fwd16r_tile<SAVE>(a, lds, (long)blockIdx.x);
}

hipError_t launch_field_fwd16r(const float* packed3, const float* rays, int ray_stride, const float* z_vals,
                               int n_rays, int S, float* raw, float* act, hipStream_t stream) {
    FieldFwdRingArgs a{packed3, rays, z_vals, raw, act, ray_stride, n_rays, S};
    const long P = (long)n_rays * S;
    if (P <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((P + PTS_PER_WG - 1) / PTS_PER_WG);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e0 = hipFuncSetAttribute((const void*)field_fwd16r_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS_FLOATS * 4);
        hipError_t e2 = hipFuncSetAttribute((const void*)field_fwd16r_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS_FLOATS * 4);
        if (e0 != hipSuccess) return e0;
        if (e2 != hipSuccess) return e2;
        attr_set = true;
    }
    if (act) hipLaunchKernelGGL(field_fwd16r_kernel<2>, dim3(blocks), dim3(FIELD_WAVES * 64), RING_LDS_FLOATS * 4, stream, a);
    else hipLaunchKernelGGL(field_fwd16r_kernel<0>, dim3(blocks), dim3(FIELD_WAVES * 64), RING_LDS_FLOATS * 4, stream, a);
    return hipGetLastError();
}

}  // namespace nerf
