// Probe: does a wave's weight-DMA wait (s_waitcnt vmcnt(N), in-order counter) inherit the completion latency of the
// stores it issued one phase earlier?  Skeleton of the saving forward: per phase and wave 8 x 1 KiB LDS-DMA pieces from
// an L2-resident 2.4 MB buffer, 96 MFMAs of dummy work, NST dwordx4 stores (policy POL) to a streaming 8 GiB buffer,
// then "wait for the DMA, let the NST newest stores drain" + workgroup barrier.
//   POL 0: no stores   1: plain   2: nt   3: sc1   4: sc0 sc1   5: nt, waited with vmcnt(0) (no counted wait)
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ inline void dma_1k(const float* gsrc_lane, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc_lane), "s"(lds_dst_uniform) : "memory");
}
template <int POL> __device__ inline void st16(f32x4* p, f32x4 v) {
    if (POL == 1) *p = v;
    else if (POL == 2 || POL == 5) __builtin_nontemporal_store(v, p);
    else if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}
template <int POL, int NST>
__global__ __launch_bounds__(512) void ack_k(const float* __restrict__ wsrc, float* __restrict__ dst, size_t dst_floats, int phases, float* out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lbase = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)lds);
    typedef float f32x4a __attribute__((ext_vector_type(4)));
    f32x4a acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4a{0, 0, 0, 0};
    u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
    // this workgroup's slice of the streaming buffer
    const size_t per_wg = dst_floats / gridDim.x;
    f32x4* my = reinterpret_cast<f32x4*>(dst + (size_t)blockIdx.x * per_wg) + wave * 64 + lane;
    size_t cursor = 0;
    const size_t per_wg16 = per_wg / 4;
    for (int ph = 0; ph < phases; ++ph) {
        // wait: DMA of the previous phase landed; NST newest stores may still drain (POL 5: everything)
        if (POL == 0 || POL == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NST) : "memory");
        __syncthreads();
        const float* src = wsrc + ((size_t)(ph * 7 + blockIdx.x) % 36) * 16384 + wave * 2048 + lane * 4;     // 64 KiB chunks of a 2.4 MB buffer
        const unsigned dstl = __builtin_amdgcn_readfirstlane(lbase + (unsigned)((ph & 1) * 65536 + wave * 8192));
#pragma unroll
        for (int i = 0; i < 8; ++i) dma_1k(src + i * 256, dstl + i * 1024);
        // 96 MFMAs of work (16x16x32 bf16: ~16 cycles each)
#pragma unroll
        for (int r = 0; r < 12; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
        if (POL != 0) {
#pragma unroll
            for (int s = 0; s < NST; ++s) {
                st16<POL>(my + (cursor % per_wg16), f32x4{acc[s & 7][0], 1.f, 2.f, 3.f});
                cursor += 512;       // the workgroup advances by 8 KiB per store round
            }
        }
    }
    float sum = 0;
    for (int i = 0; i < 8; ++i) sum += acc[i][0];
    if (sum == 12345.678f) out[0] = lds[lane];
}
extern "C" int probe_ack(int pol, int nst, const void* wsrc, void* dst, size_t dst_bytes, int blocks, int phases, void* out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const size_t nf = dst_bytes / 4;
#define GO(P, N) do { hipFuncSetAttribute((const void*)ack_k<P, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); \
    hipLaunchKernelGGL((ack_k<P, N>), dim3(blocks), dim3(512), 131072, s, (const float*)wsrc, (float*)dst, nf, phases, (float*)out); } while (0)
    if (nst == 4) { switch (pol) { case 0: GO(0, 4); break; case 1: GO(1, 4); break; case 2: GO(2, 4); break; case 3: GO(3, 4); break; case 4: GO(4, 4); break; case 5: GO(5, 4); break; } }
    else { switch (pol) { case 0: GO(0, 16); break; case 1: GO(1, 16); break; case 2: GO(2, 16); break; case 3: GO(3, 16); break; case 4: GO(4, 16); break; case 5: GO(5, 16); break; } }
    return (int)hipGetLastError();
}
