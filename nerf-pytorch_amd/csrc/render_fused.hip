// render_infer_kernel: render_rays (run_nerf.py:351-412) without gradients as ONE launch on the split-bf16 datapath.
//
// A workgroup (8 wavefronts) owns 16 consecutive rays from the coarse depths to the final colours:
//   1. coarse depths z_vals (run_nerf.py:357-379), all 512 threads;
//   2. the coarse network over its 16 x N_samples points, 128 at a time: field_fwd16r_tile, the body of field_fwd16r_kernel
//      (encode + 8x256 trunk + heads, weights through the LDS ring);
//   3. per ray, one wavefront each (two rays per wavefront): raw2outputs (:262-305) -> rgb0 / disp0 / acc0 / weights, then
//      sample_pdf + sort (:392-396, helpers:196-239) -> the 16 x (N_samples + N_importance) depths of the fine pass, z_std;
//   4. the fine network over those points, 128 at a time;
//   5. per ray: raw2outputs -> rgb / disp / acc.
// The per-ray code is the code of the stand-alone kernels (ray_device.h), the network tile is the code of the stand-alone
// forward: every output is BIT-IDENTICAL to the chain sample_coarse -> field forward -> composite -> sample_fine -> field
// forward -> composite (tests/test_gpu_round3.py).  Between the phases the rays' small arrays (depths, raw, weights: 13 MB per
// 4096 rays) go through global memory, i.e. L2, exactly as between the separate launches; `raw` is an output anyway.
//
// What it buys and what it costs: six launches and their gaps become one; the per-ray phases, which the
// separate launches spread over the whole chip (one wavefront per ray, thousands in flight), run here two rays per wavefront
// between the network tiles of a workgroup that owns its CU alone (148 KiB of LDS) -- the matrix pipes idle meanwhile.
#include "field_fwd_ring_body.h"
#include <cstdlib>
#include "ray_device.h"

namespace nerf {

// rays per workgroup (template parameter): 4, 8 or 16 -- the rays must fill whole 128-point tiles in both passes; ray k of the
// workgroup is wavefront k mod 8's in the per-ray phases
constexpr int FUSED_SCRATCH_FLOATS = 2048;  // LDS floats per wavefront in the per-ray phases (the weight ring is idle then)

template <int FUSED_RAYS, typename SP>
__global__ __launch_bounds__(FIELD_WAVES * 64) void render_infer_kernel(RenderInferArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ray0 = blockIdx.x * FUSED_RAYS;
    const int Sc = a.n_c, S2 = a.n_c + a.n_f;
    const bool fine = a.n_f > 0;
    float* sm = lds + wave * FUSED_SCRATCH_FLOATS;

    // ---- 1. coarse depths of the 16 rays
    for (int idx = threadIdx.x; idx < FUSED_RAYS * Sc; idx += FIELD_WAVES * 64) {
        const int r = ray0 + idx / Sc, j = idx % Sc;
        if (r < a.n_rays) {
            const float near = a.rays[(long)r * a.ray_stride + 6], far = a.rays[(long)r * a.ray_stride + 7];
            a.z_c[(long)r * Sc + j] = coarse_depth(near, far, [&](int k) { return linspace01_at(k, Sc); }, j, Sc, a.lindisp,
                                                   a.t_rand ? a.t_rand + (long)r * Sc : nullptr);
        }
    }

    const int tiles_c = FUSED_RAYS * Sc / PTS_PER_WG, tiles_f = fine ? FUSED_RAYS * S2 / PTS_PER_WG : 0;
#pragma unroll 1
    for (int t = 0; t < tiles_c + tiles_f; ++t) {
        if (fine && t == tiles_c) {
            // ---- 3. raw of all coarse tiles is visible to the workgroup; the weight ring is idle: its LDS is scratch
            __syncthreads();
            // (an opaque copy of the lane number: everything the per-ray phase derives from it is computed HERE -- hoisted out of
            // the tile loop it would be carried through the network tiles, whose register file is full: 256 VGPRs + scratch)
            int lane_p = lane;
            asm volatile("" : "+v"(lane_p));
#pragma unroll 1
            for (int k = wave; k < FUSED_RAYS; k += FIELD_WAVES) {     // ray k of the workgroup: wavefront k mod 8
                const int r = ray0 + k;
                if (r >= a.n_rays) break;
                CompositeArgs ca{a.raw_c, a.z_c, a.rays + 3, a.noise_c, a.noise_std, a.ray_stride, a.n_rays, Sc, a.white_bkgd,
                                 a.rgb_c, a.disp_c, a.acc_c, a.w_c, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
                composite_ray<false>(ca, r, lane_p, sm);
                // the weights this wavefront just stored are read back (other lanes) by the sampling below
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                FineArgs fa{a.z_c, a.w_c, a.u, nullptr, a.z_f, nullptr, a.z_std, a.n_rays, Sc, a.n_f, 0};
                sample_fine_ray<WaveSync>(fa, r, lane_p, sm);
                WaveSync::sync();
            }
        }
        __syncthreads();        // depths of this pass visible; LDS free for the ring
        const bool second = t >= tiles_c;
        const FieldFwdRingArgs fa{second ? a.packed_f : a.packed_c, a.rays, second ? a.z_f : a.z_c, second ? a.raw_f : a.raw_c,
                                  nullptr, a.ray_stride, a.n_rays, second ? S2 : Sc, second ? S2 : Sc, 0, second ? S2 : Sc, 0, 0};
        const long wg = second ? (long)blockIdx.x * tiles_f + (t - tiles_c) : (long)blockIdx.x * tiles_c + t;
        field_fwd16r_tile<0, SP>(fa, lds, wg);
    }
    // ---- 5. colours of the last pass
    __syncthreads();
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
#pragma unroll 1
    for (int k = wave; k < FUSED_RAYS; k += FIELD_WAVES) {
        const int r = ray0 + k;
        if (r >= a.n_rays) break;
        if (fine) {
            CompositeArgs ca{a.raw_f, a.z_f, a.rays + 3, a.noise_f, a.noise_std, a.ray_stride, a.n_rays, S2, a.white_bkgd,
                             a.rgb_f, a.disp_f, a.acc_f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
            composite_ray<false>(ca, r, lane_e, sm);
        } else {
            CompositeArgs ca{a.raw_c, a.z_c, a.rays + 3, a.noise_c, a.noise_std, a.ray_stride, a.n_rays, Sc, a.white_bkgd,
                             a.rgb_c, a.disp_c, a.acc_c, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
            composite_ray<false>(ca, r, lane_e, sm);
        }
        WaveSync::sync();
    }
}

// sizes this kernel takes: whole 128-point tiles per 16 rays in both passes, per-ray scratch within one wavefront's share
bool render_infer_fused_ok(int n_c, int n_f) {
    const int S2 = n_c + n_f;
    int np2 = 1;
    while (np2 < n_f) np2 <<= 1;
    return n_c >= 3 && n_f >= 0 && (16 * n_c) % PTS_PER_WG == 0 && (16 * S2) % PTS_PER_WG == 0 &&
           2 * S2 <= FUSED_SCRATCH_FLOATS && 3 * n_c + np2 <= FUSED_SCRATCH_FLOATS;
}

template <int R, typename SP>
static hipError_t launch_infer_one(const RenderInferArgs& a, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)render_infer_kernel<R, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS_FLOATS * 4);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const dim3 grid((unsigned)((a.n_rays + R - 1) / R)), block(FIELD_WAVES * 64);
    hipLaunchKernelGGL((render_infer_kernel<R, SP>), grid, block, RING_LDS_FLOATS * 4, stream, a);
    return hipGetLastError();
}

hipError_t launch_render_infer(const RenderInferArgs& a, hipStream_t stream) {
    if (a.n_rays <= 0) return hipSuccess;
    // rays per workgroup: the fewest (4, 8, 16) that fill whole 128-point tiles in both passes -- more, shorter workgroups
    // fill the last round of 256 better and drift apart, so that one workgroup's per-ray phase meets another's network tiles
    const int S2 = a.n_c + a.n_f;
    auto tiles = [&](int R) { return (R == 4 || R == 8 || R == 16) && (R * a.n_c) % PTS_PER_WG == 0 && (R * S2) % PTS_PER_WG == 0; };
    static const char* force = getenv("NERF_FUSED_RAYS");
    int R = tiles(4) ? 4 : tiles(8) ? 8 : 16;
    if (R == 4 && tiles(8)) {       // 8 per workgroup when that fills its last round of 256 workgroups to 95 % (measured: 4096 and
        const long wg8 = (a.n_rays + 7) / 8;        // 32768 rays 0.3-0.7 % faster with 8; 1024 and 5000 rays 13-60 % faster with 4)
        if ((double)wg8 >= 0.95 * (double)(((wg8 + 255) / 256) * 256)) R = 8;
    }
    if (force && tiles(atoi(force))) R = atoi(force);       // (only 4, 8, 16; anything else is ignored)
    if (a.split != 0 && a.split != 1) return hipErrorInvalidValue;      // (the reduced class needs its guard launch between the passes)
    if (a.split) {
        if (R == 4) return launch_infer_one<4, SplitF16>(a, stream);
        if (R == 8) return launch_infer_one<8, SplitF16>(a, stream);
        return launch_infer_one<16, SplitF16>(a, stream);
    }
    if (R == 4) return launch_infer_one<4, SplitBF16>(a, stream);
    if (R == 8) return launch_infer_one<8, SplitBF16>(a, stream);
    return launch_infer_one<16, SplitBF16>(a, stream);
}

}  // namespace nerf
