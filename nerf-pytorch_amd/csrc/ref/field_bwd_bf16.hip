// Split-bf16 ("bf16x3") backward of the field evaluation: dgrad chain (this file, same register-resident
// structure as field_fwd_bf16.hip with the transposed (hi, lo) weight stream) and the weight-gradient GEMM
// wgrad3_kernel: dW = delta^T * X with both operands split on the fly into (hi, lo) bf16 while they are staged
// into LDS (3 bf16 MFMAs per product, fp32 accumulate), contraction over points.
#include <type_traits>
#include "field_device_bf16.h"
#include "launchers.h"
#include "ref_launchers.h"

namespace nerf {

struct FieldBwd3Args {
    const float* packed3;
    const float* act;       // saved by field_fwd3_kernel<true> (bitmasks in the bf16x3 lane order)
    const float* d_raw;     // [P][4]
    float* delta;           // delta_layout3(P): 32-point feature-major tiles
    int n_rays, S;
};

template <int NV>
__device__ inline void apply_mask3(float (&d)[NV], const f32x16* acc, u32x4 m) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const unsigned bit = (m[i >> 5] >> (i & 31)) & 1u;
        d[i] = bit ? acc[i >> 4][i & 15] : 0.0f;
    }
}

// MODE 0: the split-bf16 (3-term) chain, fp32 deltas in HBM (operands of wgrad3_256_kernel).
// MODE 2: the same chain — every delta is computed exactly as in mode 0 — but what is WRITTEN for the weight-gradient
//         GEMM is rounded to bf16 (RNE; same tiles, 2-byte elements: operands of wgrad1_kernel).  The rounding touches
//         only the GEMM operands, never the chain: zero-mean, 2^-9 relative per element, averaged over the ~10^6 points
//         of the contraction (1e-4 of |dW| at 131 k points, falling with 1/sqrt(points)).
// MODE 1: mixed-precision backward: single bf16 product W_hi^T * delta_hi per term (the lo fragments of the same weight
//         stream are skipped) and bf16 deltas in HBM — here the rounding IS in the chain.
template <int MODE>
__global__ __launch_bounds__(FIELD3_WAVES * 64) void field_dgrad3_kernel(FieldBwd3Args a) {
    constexpr bool MIXED = MODE == 1;
    constexpr bool OUT16 = MODE != 0;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;
    const size_t P = (size_t)a.n_rays * a.S;
    const size_t p_raw = ((size_t)blockIdx.x * FIELD3_WAVES + wave) * PTS_PER_WAVE3 + (lane & 31);
    const bool valid = p_raw < P;
    const size_t p = valid ? p_raw : P - 1;

    WeightStreamT<MIXED ? 4 : 3, FIELD3_WAVES> ws;       // MIXED: the hi-only stream, 8 k-steps per 64 KiB chunk
    ws.start(a.packed3 + (MIXED ? P1B : P3B_VIEWS), lds, wave, lane, valid);
    stage_small_from(a.packed3 + P3_SMALL, lds, FIELD3_WAVES * 64);

    const ActLayout3 al = act_layout3(P, (size_t)a.n_rays);
    const DeltaLayout3 dl = delta_layout3(P);
    const size_t tile = (size_t)blockIdx.x * FIELD3_WAVES + wave;          // this wave's tile of every delta region
    const int lslot = half * 128 + (lane & 31);
    const f32x4 g = *reinterpret_cast<const f32x4*>(a.d_raw + p * 4);       // (d_rgb3, d_sigma)
    if (valid) {        // tile-major copy of d_raw: the A operand of the rgb_linear / alpha_linear weight gradients
        const size_t goff = tile * (4 * 32) + half * 64 + (lane & 31);
        if (OUT16) {
            __bf16* gt = reinterpret_cast<__bf16*>(a.delta + dl.graw) + goff;
            nt_store(gt, (__bf16)(half ? g[2] : g[0]));
            nt_store(gt + 32, (__bf16)(half ? g[3] : g[1]));
        } else {
            float* gt = a.delta + dl.graw + goff;
            nt_store(gt, half ? g[2] : g[0]);
            nt_store(gt + 32, half ? g[3] : g[1]);
        }
    }
    u32x4 msk[D + 1];
    {
        const u32x4* mp = reinterpret_cast<const u32x4*>(a.act + al.mask) + p * 2 + half;
#pragma unroll
        for (int l = 0; l <= D; ++l) msk[l] = mp[(size_t)l * P * 2];
    }

    // ---- rgb_linear^T (VALU) + ReLU mask of the view branch: lane value i = feature 32*(i>>4) + d32row(i&15, half)
    float dhv[64];
    {
        const float* wr = small_ptr(lds, SM_WRGB);
        __syncthreads();                 // stage_small visible
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int col = 32 * ob + 8 * q4 + 4 * half;
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr + col);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(wr + WV + col);
                const f32x4 w2 = *reinterpret_cast<const f32x4*>(wr + 2 * WV + col);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * ob + 4 * q4 + r;
                    const float v = g[0] * w0[r] + g[1] * w1[r] + g[2] * w2[r];
                    dhv[i] = ((msk[D][i >> 5] >> (i & 31)) & 1u) ? v : 0.0f;
                }
            }
    }

    f32x16 acc[8];
    float d[128];
    auto zero_acc = [&]() {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;
    };
    // every delta is written while the NEXT contraction runs, a quarter (32 stores) after each chunk acquire;
    // the following acquire<NQ> keeps those stores in flight (counted vmcnt, see WeightStreamT::acquire)
    constexpr int NQ = OUT16 ? STORES_PER_QUARTER3 / 2 : STORES_PER_QUARTER3;   // bf16 deltas leave as paired stores
    // bf16 deltas: unconditional paired stores (field_device_bf16.h).  A wave whose tile lies beyond the padded point range
    // (possible in the last workgroup only) writes to a dump tile instead: the `feat` region, which the folded datapaths
    // never use (wave-uniform select of the base, no branch)
    const bool tile_ok = __builtin_amdgcn_readfirstlane((int)(tile * 32 < pad32(P))) != 0;
    auto out16 = [&](size_t region_off, int F) {
        return reinterpret_cast<__bf16*>(a.delta + (tile_ok ? region_off : dl.feat)) + (tile_ok ? tile * (size_t)(F * 32) : (size_t)0);
    };
    using Q0 = std::integral_constant<int, 0>; using Q1 = std::integral_constant<int, 1>;
    using Q2 = std::integral_constant<int, 2>; using Q3 = std::integral_constant<int, 3>;
    auto store_q = [&](auto part, size_t off) {
        if (OUT16) store_tile3h_pair<2 * decltype(part)::value, 2>(out16(off, W), lane, d);
        else if (valid) store_tile3<2 * decltype(part)::value, 2>(a.delta + off + tile * (W * 32) + lslot, d);
    };
    using V0 = std::integral_constant<int, 0>; using V32 = std::integral_constant<int, 32>;
    using V64 = std::integral_constant<int, 64>; using V96 = std::integral_constant<int, 96>;
    auto load_alpha = [&]() {           // acc = alpha_linear^T * d_sigma (the feature_linear^T contraction adds to it)
        const float* wa = small_ptr(lds, SM_WALPHA);
#pragma unroll
        for (int ob = 0; ob < 8; ++ob)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(wa + 32 * ob + 8 * q4 + 4 * half);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[ob][4 * q4 + r] = g[3] * w[r];
            }
    };

    if constexpr (MIXED) {
        // one 64 KiB chunk = 8 k-steps of the hi-only stream: W'^T (the folded view branch, nerf_common.h) is one chunk,
        // every 256-wide contraction two; the feature_linear^T chunks behind it are skipped
        load_alpha();           // acc = alpha_linear^T d_sigma; the view branch adds W'^T d_hv: delta of the trunk output
        {
            const float* cur = ws.acquire(FOLD_SKIP_CHUNKS_BWD_HI);
            store_tile3h_pair<0, 4>(out16(dl.hv, WV), lane, dhv);
            mma1_chunk<8, 8, 0, 0, 64>(acc, dhv, cur, lane);
        }
        apply_mask3<128>(d, acc, msk[D - 1]);
#pragma unroll 1
        for (int l = D - 1; l >= 1; --l) {
            zero_acc();
            const size_t off = (size_t)l * pad32(P) * W;                       // dl.h[l]
            const float* cur = ws.template acquire<2 * NQ>();
            store_q(Q0{}, off); store_q(Q1{}, off);
            mma1_chunk<8, 8, 0, 0, 128>(acc, d, cur, lane);
            cur = ws.template acquire<2 * NQ>();
            store_q(Q2{}, off); store_q(Q3{}, off);
            mma1_chunk<8, 8, 0, 64, 128>(acc, d, cur, lane);
            u32x4 m = msk[0];                   // ReLU bitmask of h_{l-1} (static indices only: msk stays in registers)
#pragma unroll
            for (int t = 1; t < D; ++t) if (t == l - 1) m = msk[t];
            apply_mask3<128>(d, acc, m);
        }
    } else {
        auto mma_h = [&](auto voff, const float* cur) { mma3_chunk<8, 4, decltype(voff)::value, 128>(acc, d, cur, lane); };

        // ---- view branch folded with feature_linear (nerf_common.h): delta of the trunk output =
        //      (alpha_linear^T d_sigma + W'^T d_hv) * relu'(h7); the feature_linear^T chunks of the stream are skipped
        load_alpha();
        {
            const float* cur = ws.acquire();
            // 64 stores (paired bf16: 32)
            if (OUT16) store_tile3h_pair<0, 4>(out16(dl.hv, WV), lane, dhv);
            else if (valid) store_tile3<0, 4>(a.delta + dl.hv + tile * (WV * 32) + lslot, dhv);
            mma3_chunk<8, 4, 0, 64>(acc, dhv, cur, lane);
        }
        mma3_chunk<8, 4, 32, 64>(acc, dhv, ws.template acquire<OUT16 ? 32 : 63>(FOLD_SKIP_CHUNKS_BWD), lane);
        apply_mask3<128>(d, acc, msk[D - 1]);

        // ---- trunk: delta_{l-1} = (W_l^T delta_l) * relu'(h_{l-1}),  l = 7 .. 1
#pragma unroll 1
        for (int l = D - 1; l >= 1; --l) {
            zero_acc();
            const size_t off = (size_t)l * pad32(P) * W;                   // dl.h[l]: delta of layer l = input of this step
            const float* cur = ws.acquire();
            store_q(Q0{}, off);
            mma_h(V0{}, cur);
            cur = ws.template acquire<NQ>();
            store_q(Q1{}, off);
            mma_h(V32{}, cur);
            cur = ws.template acquire<NQ>();
            store_q(Q2{}, off);
            mma_h(V64{}, cur);
            cur = ws.template acquire<NQ>();
            store_q(Q3{}, off);
            mma_h(V96{}, cur);
            u32x4 m = msk[0];                   // ReLU bitmask of h_{l-1} (static indices only: msk stays in registers)
#pragma unroll
            for (int t = 1; t < D; ++t) if (t == l - 1) m = msk[t];
            apply_mask3<128>(d, acc, m);
        }
    }
    // dl.h[0]
    if (OUT16) store_tile3h_pair<0, 8>(out16(0, W), lane, d);
    else if (valid) store_tile3<0, 8>(a.delta + tile * (W * 32) + lslot, d);
}

hipError_t launch_field_dgrad3(const float* packed3, const float* act, const float* d_raw, int n_rays, int S,
                               float* delta, int mode, hipStream_t stream) {
    const long P = (long)n_rays * S;
    if (P <= 0) return hipSuccess;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)field_dgrad3_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, FIELD_LDS_FLOATS * 4);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)field_dgrad3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, FIELD_LDS_FLOATS * 4);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)field_dgrad3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, FIELD_LDS_FLOATS * 4);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    FieldBwd3Args ba{packed3, act, d_raw, delta, n_rays, S};
    const unsigned blocks = (unsigned)((P + PTS_PER_WG3 - 1) / PTS_PER_WG3);
    if (mode == 1) hipLaunchKernelGGL(field_dgrad3_kernel<1>, dim3(blocks), dim3(FIELD3_WAVES * 64), FIELD_LDS_FLOATS * 4, stream, ba);
    else if (mode == 2) hipLaunchKernelGGL(field_dgrad3_kernel<2>, dim3(blocks), dim3(FIELD3_WAVES * 64), FIELD_LDS_FLOATS * 4, stream, ba);
    else hipLaunchKernelGGL(field_dgrad3_kernel<0>, dim3(blocks), dim3(FIELD3_WAVES * 64), FIELD_LDS_FLOATS * 4, stream, ba);
    return hipGetLastError();
}

}  // namespace nerf
