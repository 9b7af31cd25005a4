// Probe: what does a bare MFMA stream of the field kernels' shape reach on this chip, with realistic (random) operand data?
//   MODE 0: 32x32x16 bf16, 1 wave / SIMD (256-thread workgroups, 148 KiB LDS to force one workgroup per CU), 8 accumulators of
//           16 registers, the dgrad's order (4 blocks x {hi*bhi, hi*blo, lo*bhi});  MODE 1: 16x16x32 bf16, 2 waves / SIMD
//           (512 threads), 16 accumulators of 4 registers, the forward's order.
//   FILL 0: MFMAs only.  FILL 1: + one ds_read_b128 (lane-linear, conflict-free) per MFMA for 8 of every 12 MFMAs, fragments
//           used 8 MFMAs later (the ring kernels' request pattern).  FILL 2: + 3 VALU operations per MFMA on top.
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int FILL>
__global__ __launch_bounds__(MODE == 0 ? 256 : 512) void mfma_k(const u32x4* __restrict__ data, float* out, int units) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) reinterpret_cast<u32x4*>(lds)[i] = data[i & 4095];
    __syncthreads();
    const u32x4* lp = reinterpret_cast<const u32x4*>(lds) + lane;
    u32x4 hi[4], lo[4], nh[4];
    for (int i = 0; i < 4; ++i) { hi[i] = data[lane + 64 * i]; lo[i] = data[lane + 64 * (i + 4)]; nh[i] = hi[i]; }
    u32x4 bhi = data[lane + 512], blo = data[lane + 576];
    float junk = 0.f;
    if constexpr (MODE == 0) {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int u0 = 0; u0 < units; u0 += 2)
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
            const int u = u0 + uu;
            const int g4 = uu * 4;
            const u32x4* p = lp + (u & 7) * 512;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[g4 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, hi[i]), __builtin_bit_cast(bf16x8, bhi), acc[g4 + i], 0, 0, 0);
                if (FILL >= 1) lo[i] = p[(2 * i + 1) * 64];
                if (FILL >= 2) { junk = junk * 1.0001f + 0.5f; junk = junk * 0.9999f - 0.25f; bhi[0] ^= (unsigned)i; }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[g4 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, hi[i]), __builtin_bit_cast(bf16x8, blo), acc[g4 + i], 0, 0, 0);
                if (FILL >= 1) nh[i] = p[(2 * i) * 64 + 512];
                if (FILL >= 2) { junk = junk * 1.0001f + 0.5f; junk = junk * 0.9999f - 0.25f; blo[1] ^= (unsigned)i; }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[g4 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, lo[i]), __builtin_bit_cast(bf16x8, bhi), acc[g4 + i], 0, 0, 0);
                if (FILL >= 2) { junk = junk * 1.0001f + 0.5f; junk = junk * 0.9999f - 0.25f; bhi[2] ^= (unsigned)i; }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (FILL >= 1) for (int i = 0; i < 4; ++i) hi[i] = nh[i];
        }
        float s = junk;
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][7];
        if (s == 12345.678f) out[0] = s;
    } else {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int u0 = 0; u0 < units; u0 += 4)
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
            const int u = u0 + uu;
            const int g4 = uu * 4;
            const u32x4* p = lp + (u & 7) * 512;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[g4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, hi[i]), __builtin_bit_cast(bf16x8, bhi), acc[g4 + i], 0, 0, 0);
                if (FILL >= 1) lo[i] = p[(2 * i + 1) * 64];
                if (FILL >= 2) { junk = junk * 1.0001f + 0.5f; bhi[0] ^= (unsigned)i; }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[g4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, hi[i]), __builtin_bit_cast(bf16x8, blo), acc[g4 + i], 0, 0, 0);
                if (FILL >= 1) nh[i] = p[(2 * i) * 64 + 512];
                if (FILL >= 2) { junk = junk * 1.0001f + 0.5f; blo[1] ^= (unsigned)i; }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[g4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, lo[i]), __builtin_bit_cast(bf16x8, bhi), acc[g4 + i], 0, 0, 0);
                if (FILL >= 2) { junk = junk * 1.0001f + 0.5f; bhi[2] ^= (unsigned)i; }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (FILL >= 1) for (int i = 0; i < 4; ++i) hi[i] = nh[i];
        }
        float s = junk;
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
        if (s == 12345.678f) out[0] = s;
    }
}

// MODE 2 (what a tf32-class datapath would issue instead of MODE 1): per 16 output blocks and K = 128, the main term as four
// 16x16x32 bf16 MFMAs per block and BOTH correction terms as one 16x16x128 fp8 MFMA each (block-scaled f8f6f4, scale 1.0) --
// 64 bf16 + 32 fp8 MFMAs where MODE 1 issues 192 bf16 MFMAs.  FILL as above (fragment reads for 2 of every 3 MFMAs).
typedef int i32x8 __attribute__((ext_vector_type(8)));
template <int FILL>
__global__ __launch_bounds__(512) void mfma8_k(const u32x4* __restrict__ data, float* out, int groups) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) reinterpret_cast<u32x4*>(lds)[i] = data[i & 4095];
    __syncthreads();
    const u32x4* lp = reinterpret_cast<const u32x4*>(lds) + lane;
    u32x4 hi[4], f8a[4];
    for (int i = 0; i < 4; ++i) { hi[i] = data[lane + 64 * i]; f8a[i] = data[lane + 64 * (i + 4)]; }
    u32x4 bhi = data[lane + 512], f8b0 = data[lane + 576], f8b1 = data[lane + 640];
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float junk = 0.f;
    for (int g = 0; g < groups; ++g) {          // one group = K 128 x 16 blocks
        const u32x4* p = lp + (g & 7) * 512;
#pragma unroll
        for (int blk = 0; blk < 16; ++blk) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                acc[blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, hi[ks]), __builtin_bit_cast(bf16x8, bhi), acc[blk], 0, 0, 0);
                if (FILL >= 1) hi[ks] = p[((blk * 4 + ks) & 7) * 64];
                if (FILL >= 2) { junk = junk * 1.0001f + 0.5f; bhi[0] ^= (unsigned)ks; }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                i32x8 a, b;
                for (int w = 0; w < 4; ++w) { a[w] = (int)f8a[t][w]; a[4 + w] = (int)f8a[2 + t][w]; b[w] = (int)f8b0[w]; b[4 + w] = (int)f8b1[w]; }
                acc[blk] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[blk], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                if (FILL >= 1) f8a[t] = p[(8 + ((blk + t) & 3)) * 64];
                if (FILL >= 2) { junk = junk * 1.0001f + 0.5f; f8b0[1] ^= (unsigned)t; }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = junk;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}

extern "C" int probe_mfma8(int fill, const void* data, float* out, int blocks, int groups, void* stream) {
    const size_t lds = 148 * 1024;
#define GO8(F) { hipFuncSetAttribute((const void*)mfma8_k<F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                 hipLaunchKernelGGL((mfma8_k<F>), dim3(blocks), dim3(512), lds, (hipStream_t)stream, (const u32x4*)data, out, groups); }
    if (fill == 0) GO8(0) else if (fill == 1) GO8(1) else GO8(2)
    return (int)hipGetLastError();
}

extern "C" int probe_mfma(int mode, int fill, const void* data, float* out, int blocks, int units, void* stream) {
    const size_t lds = 148 * 1024;
#define GO(M, F) { hipFuncSetAttribute((const void*)mfma_k<M, F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                   hipLaunchKernelGGL((mfma_k<M, F>), dim3(blocks), dim3(M == 0 ? 256 : 512), lds, (hipStream_t)stream, (const u32x4*)data, out, units); }
    if (mode == 0) { if (fill == 0) GO(0, 0) else if (fill == 1) GO(0, 1) else GO(0, 2) }
    else { if (fill == 0) GO(1, 0) else if (fill == 1) GO(1, 1) else GO(1, 2) }
    return (int)hipGetLastError();
}
