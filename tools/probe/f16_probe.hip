// Probe for the fp16 three-term split (round 4): what do gfx950's f16 MFMAs do with SUBNORMAL operands, what do the
// conversions around them do, and does an f16 stream run at the bf16 stream's rate / power?
//   probe_f16_values : one wave; A = a (uniform), B = b (uniform) through v_mfma_f32_16x16x32_f16 and 32x32x16_f16 -> every
//                      output = K * a * b if nothing is flushed.  Plus v_cvt_pk_f16_f32 and v_fma_mix_f32 on a value table.
//   probe_f16_stream : the forward's (MODE 1) / dgrad's (MODE 0) MFMA order of mfma_probe.hip with f16 or bf16 MFMAs
//                      (F16 = 1 / 0), FILL as there.
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// in: pairs (a_bits, b_bits) of f16 patterns; out: [case][4] = {16x16x32 result, 32x32x16 result, bf16 16x16x32 with the same
// 16-bit patterns, 0}
__global__ __launch_bounds__(64) void values_k(const unsigned* __restrict__ cases, int n_cases, float* __restrict__ out) {
    for (int c = 0; c < n_cases; ++c) {
        const unsigned a = cases[2 * c] & 0xffffu, b = cases[2 * c + 1] & 0xffffu;
        const u32x4 A = {a * 65537u, a * 65537u, a * 65537u, a * 65537u};
        const u32x4 B = {b * 65537u, b * 65537u, b * 65537u, b * 65537u};
        f32x4 c4 = {0, 0, 0, 0};
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), c4, 0, 0, 0);
        f32x16 c16;
        for (int r = 0; r < 16; ++r) c16[r] = 0.f;
        c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), c16, 0, 0, 0);
        f32x4 cb = {0, 0, 0, 0};
        cb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), cb, 0, 0, 0);
        if (threadIdx.x == 0) {
            out[4 * c + 0] = c4[0];
            out[4 * c + 1] = c16[0];
            out[4 * c + 2] = cb[0];
            out[4 * c + 3] = 0.f;
        }
    }
}

// conversions: in[i] (fp32) -> out[4 i ..] = {cvt_pk_f16 bits (low half), fma_mix(a - f32(hi)) as float bits, cvt_pk of that, 0}
__global__ __launch_bounds__(64) void cvt_k(const float* __restrict__ in, int n, unsigned* __restrict__ out) {
    const int i = threadIdx.x + blockIdx.x * 64;
    if (i >= n) return;
    const float a = in[i];
    unsigned r, r2;
    float lo;
    asm("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(r) : "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(r), "v"(a));
    asm("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(r2) : "v"(lo));
    out[4 * i + 0] = r & 0xffffu;
    out[4 * i + 1] = __float_as_uint(lo);
    out[4 * i + 2] = r2 & 0xffffu;
    out[4 * i + 3] = 0u;
}

template <int F16> __device__ __forceinline__ f32x16 mm32(u32x4 a, u32x4 b, f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <int F16> __device__ __forceinline__ f32x4 mm16(u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int MODE, int FILL, int F16>
__global__ __launch_bounds__(MODE == 0 ? 256 : 512) void stream_k(const u32x4* __restrict__ data, float* out, int units) {
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
                acc[g4 + i] = mm32<F16>(hi[i], bhi, acc[g4 + i]);
                if (FILL >= 1) lo[i] = p[(2 * i + 1) * 64];
                if (FILL >= 2) { junk = junk * 1.0001f + 0.5f; junk = junk * 0.9999f - 0.25f; bhi[0] ^= (unsigned)i; }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[g4 + i] = mm32<F16>(hi[i], blo, acc[g4 + i]);
                if (FILL >= 1) nh[i] = p[(2 * i) * 64 + 512];
                if (FILL >= 2) { junk = junk * 1.0001f + 0.5f; junk = junk * 0.9999f - 0.25f; blo[1] ^= (unsigned)i; }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[g4 + i] = mm32<F16>(lo[i], bhi, acc[g4 + i]);
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
                acc[g4 + i] = mm16<F16>(hi[i], bhi, acc[g4 + i]);
                if (FILL >= 1) lo[i] = p[(2 * i + 1) * 64];
                if (FILL >= 2) { junk = junk * 1.0001f + 0.5f; bhi[0] ^= (unsigned)i; }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[g4 + i] = mm16<F16>(hi[i], blo, acc[g4 + i]);
                if (FILL >= 1) nh[i] = p[(2 * i) * 64 + 512];
                if (FILL >= 2) { junk = junk * 1.0001f + 0.5f; blo[1] ^= (unsigned)i; }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[g4 + i] = mm16<F16>(lo[i], bhi, acc[g4 + i]);
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

extern "C" int probe_f16_values(const unsigned* cases, int n_cases, float* out, void* stream) {
    hipLaunchKernelGGL(values_k, dim3(1), dim3(64), 0, (hipStream_t)stream, cases, n_cases, out);
    return (int)hipGetLastError();
}
extern "C" int probe_f16_cvt(const float* in, int n, unsigned* out, void* stream) {
    hipLaunchKernelGGL(cvt_k, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, in, n, out);
    return (int)hipGetLastError();
}
extern "C" int probe_f16_stream(int mode, int fill, int f16, const void* data, float* out, int blocks, int units, void* stream) {
    const size_t lds = 148 * 1024;
#define GO(M, F, H) { hipFuncSetAttribute((const void*)stream_k<M, F, H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                      hipLaunchKernelGGL((stream_k<M, F, H>), dim3(blocks), dim3(M == 0 ? 256 : 512), lds, (hipStream_t)stream, (const u32x4*)data, out, units); }
#define GOF(M, H) { if (fill == 0) GO(M, 0, H) else if (fill == 1) GO(M, 1, H) else GO(M, 2, H) }
    if (mode == 0) { if (f16) GOF(0, 1) else GOF(0, 0) }
    else { if (f16) GOF(1, 1) else GOF(1, 0) }
    return (int)hipGetLastError();
}
