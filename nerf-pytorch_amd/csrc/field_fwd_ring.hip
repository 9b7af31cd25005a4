// field_fwd16r_kernel: the 16-point-per-wave three-term-split forward (encode + 8x256 trunk + density head + folded view
// branch -> raw[P,4]; run_nerf.py:37-51, run_nerf_helpers.py:15-45, :96-119) on the weight RING of field_ring.h.  Same fragment
// stream (P16F), same MFMA order per accumulator, same encodings and heads as round 2's double-buffered kernel (deleted in round 5):
// with SP = SplitBF16 `raw`, the saved rows, encodings and ReLU bitmasks were BIT-IDENTICAL to it while both existed -- the digests
// of tests/golden/kernel_digests.json were recorded in that state and pin every later change; what changed against that kernel
// is when the weights arrive and when the fragments are requested.  SAVE: 0 = inference, 2 = 16-bit rows of the split's type (operands of the weight-gradient
// GEMM).  SP (split_types.h): bf16 split or fp16 split (fp32-class products, 11-bit saved rows); RED: the reduced inference class
// (field_ring8.h).
#include "field_fwd_ring_body.h"

namespace nerf {

template <int SAVE, typename SP, bool RED = false>
__global__ __launch_bounds__(FIELD_WAVES * 64) void field_fwd16r_kernel(FieldFwdRingArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    field_fwd16r_tile<SAVE, SP, RED>(a, lds, (long)blockIdx.x);
}

template <int SAVE, typename SP, bool RED = false>
static hipError_t launch_one(const FieldFwdRingArgs& a, unsigned blocks, hipStream_t stream) {
    static bool attr_set = false;       // (one flag per instantiation)
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)field_fwd16r_kernel<SAVE, SP, RED>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS_FLOATS * 4);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((field_fwd16r_kernel<SAVE, SP, RED>), dim3(blocks), dim3(FIELD_WAVES * 64), RING_LDS_FLOATS * 4, stream, a);
    return hipGetLastError();
}

// split: 0 = bf16 (packed3 from the bf16 repack), 1 = fp16 (packed3 from the fp16 repack); split_types.h
// 2 = fp16 main term + fp8 correction terms (field_ring8.h; inference only: act must be null; packed3 from the reduced repack);
// 3 = the same, every ray's last sample left unwritten (launch_field_fwd16r_last evaluates it);
// 5 = fp16 with TWO-WORD saves (hi and lo rows: act laid out by act_layout3(P, N, true)); without act it is split 1
hipError_t launch_field_fwd16r(const float* packed3, const float* rays, int ray_stride, const float* z_vals,
                               int n_rays, int S, float* raw, float* act, int split, hipStream_t stream) {
    FieldFwdRingArgs a{packed3, rays, z_vals, raw, act, ray_stride, n_rays, S, S, 0, S, 0, split == 3 ? 1 : 0};
    const long P = (long)n_rays * S;
    if (P <= 0) return hipSuccess;
    // the kernel's point arithmetic is 32-bit, and a saving launch addresses its rows with 32-bit lane offsets (field_fwd_ring_body.h)
    if (P > FWD16R_MAX_POINTS || (act && P > FWD16R_MAX_SAVED_POINTS)) return hipErrorInvalidValue;
    const unsigned blocks = (unsigned)((P + PTS_PER_WG - 1) / PTS_PER_WG);
    if (split == 5) return act ? launch_one<3, SplitF16>(a, blocks, stream) : launch_one<0, SplitF16>(a, blocks, stream);
    if (split >= 2) return act ? hipErrorInvalidValue : launch_one<0, SplitF16, true>(a, blocks, stream);
    if (split) return act ? launch_one<2, SplitF16>(a, blocks, stream) : launch_one<0, SplitF16>(a, blocks, stream);
    return act ? launch_one<2, SplitBF16>(a, blocks, stream) : launch_one<0, SplitBF16>(a, blocks, stream);
}

// every ray's LAST sample of a pass of S samples, three-term fp16 products, written into the pass's raw[n_rays][S][4]: the guard of
// the reduced inference class.  The reference appends dists[-1] = 1e10 (run_nerf.py:277-278), so alpha of the last sample is a STEP
// function of the sign of its density (:293): a density within the product class's error of zero flips a ray's opacity
// (tools/analysis_accuracy_classes.py).  Evaluating those n_rays points (1/64 and 1/192 of the passes) in the fp32-class products
// makes flips as rare as on the fp16x3 datapath.  packed3: the fp16 three-term repack of the same parameters.
//   packed3_next (nullable): in the SAME launch, the last sample of the hierarchical pass that refines this one, evaluated by that
// pass's network into raw_next[n_rays][S_next][4].  Its depth is known before its samples are drawn: sample_pdf draws inside
// [z_mid[0], z_mid[-1]] (run_nerf.py:392-396, helpers:196-239), so the sorted union keeps this pass's last depth as its last depth.
// One launch of 2 x n_rays / 128 workgroups costs one pass of a workgroup through the network (~70 us) whether it evaluates one
// network or two: a 4096-ray batch pays that once instead of twice.
template <typename SP>
__global__ __launch_bounds__(FIELD_WAVES * 64) void field_fwd16r_last2_kernel(FieldFwdRingArgs a, FieldFwdRingArgs b, unsigned blocks_a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (blockIdx.x < blocks_a) field_fwd16r_tile<0, SP>(a, lds, (long)blockIdx.x);
    else field_fwd16r_tile<0, SP>(b, lds, (long)(blockIdx.x - blocks_a));
}

hipError_t launch_field_fwd16r_last(const float* packed3, const float* rays, int ray_stride, const float* z_vals,
                                    int n_rays, int S, float* raw, const float* packed3_next, float* raw_next, int S_next,
                                    hipStream_t stream) {
    if (n_rays <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n_rays + PTS_PER_WG - 1) / PTS_PER_WG);
    FieldFwdRingArgs a{packed3, rays, z_vals, raw, nullptr, ray_stride, n_rays, 1, S, S - 1, S, S - 1, 0};
    if (!packed3_next) return launch_one<0, SplitF16>(a, blocks, stream);
    FieldFwdRingArgs b{packed3_next, rays, z_vals, raw_next, nullptr, ray_stride, n_rays, 1, S, S - 1, S_next, S_next - 1, 0};
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)field_fwd16r_last2_kernel<SplitF16>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS_FLOATS * 4);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((field_fwd16r_last2_kernel<SplitF16>), dim3(2 * blocks), dim3(FIELD_WAVES * 64), RING_LDS_FLOATS * 4, stream, a, b, blocks);
    return hipGetLastError();
}

}  // namespace nerf
