// Fused field evaluation: ray point -> positional encoding -> 8x256 trunk with
// skip -> density head + view-dependent colour head -> raw[P,4].
// Replaces run_nerf.py:37-51 (run_network + batchify) and
// run_nerf_helpers.py:15-45 (Embedder.embed), :96-119 (NeRF.forward): the
// [P,90] encoded tensor, the per-sample view-direction broadcast and every
// intermediate activation round trip disappear; activations stay in VGPRs
// (transposed MFMA formulation, see nerf_common.h), weights stream L2 -> LDS.
//
// One wave = 16 points; one 512-thread workgroup = 128 points, 2 waves / SIMD.
// Exact fp32: v_mfma_f32_16x16x4_f32 is bitwise an fmaf chain.
#include "field_device.h"

#include "launchers.h"

namespace nerf {

struct FieldFwdArgs {
    const float* packed;    // PACKED_FLOATS
    const float* rays;      // [N][ray_stride]: o3 d3 near far viewdir3
    const float* z_vals;    // [N][S]
    float* raw;             // [N][S][4]
    float* act;             // nullable: saved activations (act_layout)
    int ray_stride;
    int n_rays;
    int S;
};

template <bool SAVE>
__global__ __launch_bounds__(FIELD_WAVES * 64) void field_fwd_kernel(FieldFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4;
    const long P = (long)a.n_rays * a.S;
    const long p_raw = ((long)blockIdx.x * FIELD_WAVES + wave) * PTS_PER_WAVE + (lane & 15);
    const bool valid = p_raw < P;
    const long p = valid ? p_raw : P - 1;
    const int ray = (int)(p / a.S);

    WeightStream<true> ws;
    ws.start(a.packed, lds, wave, lane);
    stage_small(a.packed, lds);          // visible after the first acquire() barrier

    // ---- sample point (run_nerf.py:381: o + d * z, separate multiply and add)
    const float* rp = a.rays + (long)ray * a.ray_stride;
    const float z = a.z_vals[p];
    const float x0 = rp[0] + rp[3] * z;
    const float x1 = rp[1] + rp[4] * z;
    const float x2 = rp[2] + rp[5] * z;
    const float vd0 = rp[8], vd1 = rp[9], vd2 = rp[10];

    float e[16];
    encode_xyz(e, x0, x1, x2, q);

    ActLayout al{};
    if (SAVE) {
        al = act_layout((size_t)P, (size_t)a.n_rays);
        if (valid) {
            float* eo = a.act + al.enc + (size_t)p * 64;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int col = encslot(s, q);
                if (col >= 0) nt_store(eo + col, e[s]);
            }
        }
    }

    const float* bias = small_ptr(lds, SM_BIAS);
    f32x4 acc[16];
    float h[64];

    // ---- layer 0: 63 -> 256
    load_bias<16>(acc, bias, q);
    mma_chunk<16, 16, 0, 16>(acc, e, ws.acquire(), lane);
#pragma unroll
    for (int nb = 0; nb < 16; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[4 * nb + r] = fmaxf(acc[nb][r], 0.0f);
    // Saved activations are stored one chunk late: right AFTER the next acquire() has issued its DMA, so the
    // stores have a whole chunk of MFMA work to drain before the following vmcnt(0) + barrier.
    auto save_trunk = [&](int layer) {
        if (SAVE && valid) {
            float* ho = a.act + (size_t)layer * (size_t)P * W + (size_t)p * W + 4 * q;      // == al.h[layer]
#pragma unroll
            for (int nb = 0; nb < 16; ++nb)
                nt_store(reinterpret_cast<f32x4*>(ho + 16 * nb), f32x4{h[4 * nb], h[4 * nb + 1], h[4 * nb + 2], h[4 * nb + 3]});
            save_mask<64>(a.act + al.mask, layer, (size_t)P, (size_t)p, q, h);
        }
    };

    // ---- layers 1..7 (layer 5 also contracts the xyz encoding: skip connection)
#pragma unroll 1
    for (int l = 1; l < D; ++l) {
        load_bias<16>(acc, bias + l * W, q);
        const float* first = ws.acquire();
        save_trunk(l - 1);
        if (l == SKIP + 1) { mma_chunk<16, 16, 0, 16>(acc, e, first, lane); first = ws.acquire(); }
        mma_chunk<16, 16, 0, 64>(acc, h, first, lane);
        mma_chunk<16, 16, 16, 64>(acc, h, ws.acquire(), lane);
        mma_chunk<16, 16, 32, 64>(acc, h, ws.acquire(), lane);
        mma_chunk<16, 16, 48, 64>(acc, h, ws.acquire(), lane);
#pragma unroll
        for (int nb = 0; nb < 16; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) h[4 * nb + r] = fmaxf(acc[nb][r], 0.0f);
    }

    // ---- density head: alpha_linear 256 -> 1 (VALU dot + quarter reduction)
    float sigma = 0.0f;
    {
        const float* wa = small_ptr(lds, SM_WALPHA) + 4 * q;
#pragma unroll
        for (int nb = 0; nb < 16; ++nb) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(wa + 16 * nb);
#pragma unroll
            for (int r = 0; r < 4; ++r) sigma = fmaf(h[4 * nb + r], w[r], sigma);
        }
        sigma = quarter_sum(sigma) + small_ptr(lds, SM_BALPHA)[0];
    }

    // ---- feature_linear 256 -> 256 (no activation)
    load_bias<16>(acc, small_ptr(lds, SM_BFEAT), q);
    {
        const float* first = ws.acquire();
        save_trunk(D - 1);
        mma_chunk<16, 16, 0, 64>(acc, h, first, lane);
    }
    mma_chunk<16, 16, 16, 64>(acc, h, ws.acquire(), lane);
    mma_chunk<16, 16, 32, 64>(acc, h, ws.acquire(), lane);
    mma_chunk<16, 16, 48, 64>(acc, h, ws.acquire(), lane);
#pragma unroll
    for (int nb = 0; nb < 16; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[4 * nb + r] = acc[nb][r];

    // ---- view branch: [feature, enc(dir)] 283 -> 128, ReLU
    float v[7];
    encode_dir(v, vd0, vd1, vd2, q);
    if (SAVE && valid && (p - (long)ray * a.S) == 0) {
        float* dout = a.act + al.dir + (size_t)ray * 32;
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            const int col = dirslot(s, q);
            if (col >= 0) nt_store(dout + col, v[s]);
        }
    }
    f32x4 av[8];
    load_bias<8>(av, small_ptr(lds, SM_BVIEWS), q);
    {
        const float* first = ws.acquire();
        if (SAVE && valid) {
            float* ho = a.act + al.feat + (size_t)p * W + 4 * q;
#pragma unroll
            for (int nb = 0; nb < 16; ++nb)
                nt_store(reinterpret_cast<f32x4*>(ho + 16 * nb), f32x4{h[4 * nb], h[4 * nb + 1], h[4 * nb + 2], h[4 * nb + 3]});
        }
        mma_chunk<8, 16, 0, 64>(av, h, first, lane);
    }
    mma_chunk<8, 16, 16, 64>(av, h, ws.acquire(), lane);
    mma_chunk<8, 16, 32, 64>(av, h, ws.acquire(), lane);
    mma_chunk<8, 16, 48, 64>(av, h, ws.acquire(), lane);
    mma_chunk<8, 7, 0, 7>(av, v, ws.acquire(), lane);
    float hv[32];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[4 * nb + r] = fmaxf(av[nb][r], 0.0f);
    if (SAVE && valid) {
        float* ho = a.act + al.hv + (size_t)p * WV + 4 * q;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
            nt_store(reinterpret_cast<f32x4*>(ho + 16 * nb), f32x4{hv[4 * nb], hv[4 * nb + 1], hv[4 * nb + 2], hv[4 * nb + 3]});
        save_mask<32>(a.act + al.mask, D, (size_t)P, (size_t)p, q, hv);
    }

    // ---- rgb_linear 128 -> 3
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    {
        const float* wr = small_ptr(lds, SM_WRGB) + 4 * q;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr + 16 * nb);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(wr + WV + 16 * nb);
            const f32x4 w2 = *reinterpret_cast<const f32x4*>(wr + 2 * WV + 16 * nb);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                c0 = fmaf(hv[4 * nb + r], w0[r], c0);
                c1 = fmaf(hv[4 * nb + r], w1[r], c1);
                c2 = fmaf(hv[4 * nb + r], w2[r], c2);
            }
        }
        c0 = quarter_sum(c0) + small_ptr(lds, SM_BRGB)[0];
        c1 = quarter_sum(c1) + small_ptr(lds, SM_BRGB)[1];
        c2 = quarter_sum(c2) + small_ptr(lds, SM_BRGB)[2];
    }
    if (valid && q == 0) *reinterpret_cast<f32x4*>(a.raw + (size_t)p * 4) = f32x4{c0, c1, c2, sigma};
}

hipError_t launch_field_fwd(const float* packed, const float* rays, int ray_stride, const float* z_vals,
                            int n_rays, int S, float* raw, float* act, hipStream_t stream) {
    FieldFwdArgs a{packed, rays, z_vals, raw, act, ray_stride, n_rays, S};
    const long P = (long)n_rays * S;
    if (P <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((P + PTS_PER_WG - 1) / PTS_PER_WG);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e1 = hipFuncSetAttribute((const void*)field_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FIELD_LDS_FLOATS * 4);
        hipError_t e2 = hipFuncSetAttribute((const void*)field_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, FIELD_LDS_FLOATS * 4);
        if (e1 != hipSuccess) return e1;
        if (e2 != hipSuccess) return e2;
        attr_set = true;
    }
    if (act)
        hipLaunchKernelGGL(field_fwd_kernel<true>, dim3(blocks), dim3(FIELD_WAVES * 64), FIELD_LDS_FLOATS * 4, stream, a);
    else
        hipLaunchKernelGGL(field_fwd_kernel<false>, dim3(blocks), dim3(FIELD_WAVES * 64), FIELD_LDS_FLOATS * 4, stream, a);
    return hipGetLastError();
}

}  // namespace nerf
