// Device-side building blocks shared by the forward and backward field kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "nerf_common.h"

namespace nerf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Streaming store (`nt`): everything the backward reads back (activations, deltas, bitmasks) is written once and not
// touched again by the writing kernel.  A plain store allocates its line in the XCD's L2, and gigabytes of them evict
// the 2.4 MB weight stream every workgroup keeps re-reading through L2 -> LDS DMA: the forward with saves ran 1.8x
// slower than without until its stores went non-temporal (measured: 4.2 -> 3.0 ms on 786k points).
template <typename T>
__device__ inline void nt_store(T* ptr, T v) { __builtin_nontemporal_store(v, ptr); }

// The same store with the address given as (wave-uniform 64-bit base in scalar registers) + (32-bit byte offset of the lane):
// written as asm because the compiler, given the sum, hoists `base + lane offset` into a 64-bit VGPR pair per base.
__device__ __forceinline__ void nt_store_saddr(const void* uniform_base, unsigned lane_bytes, unsigned v) {
    asm volatile("global_store_dword %0, %1, %2 nt" ::"v"(lane_bytes), "v"(v), "s"(uniform_base) : "memory");
}

typedef unsigned u32x4_st __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store_saddr(const void* uniform_base, unsigned lane_bytes, u32x4_st v) {
    asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(lane_bytes), "v"(v), "s"(uniform_base) : "memory");
}

#define NERF_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define NERF_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int CHUNK_FLOATS = CHUNK_KS * KSTEP_F16;      // 16384 floats = 64 KiB per LDS buffer

// forward weight-stream chunk sizes (floats), in consumption order
__device__ constexpr int fwd_chunk_floats(int c) {
    return c < 34 ? CHUNK_FLOATS : (c < 38 ? CHUNK_KS * KSTEP_F8 : (c == 38 ? KS_DIR * KSTEP_F8 : 0));
}
// backward stream: VIEWS^T (2) | FEAT^T (4) | L7^T..L1^T (28)
__device__ constexpr int bwd_chunk_floats(int c) { return c < 34 ? CHUNK_FLOATS : 0; }

// LDS byte address (group-segment offset) of a generic pointer into __shared__ memory
__device__ inline unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

// L2 -> LDS DMA of one chunk: every wave-instruction moves 64 lanes x 16 B = 1 KiB
// into a lane-linear LDS image (wave-uniform base in M0 + lane*16).
// Issued through inline asm on purpose: hipcc treats the builtin as a possible
// alias of every later ds_read and drains it with vmcnt(0) before the first MFMA
// of the *current* chunk, which serialises the prefetch.  The asm form is not
// tracked, so completion is waited for explicitly in WeightStream::acquire().
__device__ inline void dma_1k(const float* gsrc_lane, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc_lane), "s"(lds_dst_uniform)
        : "memory");
}
// four consecutive 1 KiB pieces under ONE M0 set-up: the immediate offset advances the global and the LDS address alike
__device__ inline void dma_4k(const float* gsrc_lane, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "global_load_lds_dwordx4 %1, off offset:1024\n\t"
        "global_load_lds_dwordx4 %1, off offset:2048\n\t"
        "global_load_lds_dwordx4 %1, off offset:3072\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc_lane), "s"(lds_dst_uniform)
        : "memory");
}
// two consecutive 1 KiB pieces under one M0 set-up (weight ring, field_ring.h)
__device__ inline void dma_2k(const float* gsrc_lane, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "global_load_lds_dwordx4 %1, off offset:1024\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc_lane), "s"(lds_dst_uniform)
        : "memory");
}
// every wave copies one contiguous span of the chunk (nfloats / NWAVES, a multiple of 256 floats) in 4 KiB and 1 KiB steps
template <int NWAVES>
__device__ inline void dma_chunk(const float* gsrc, float* lbuf, int nfloats, int wave, int lane) {
    const unsigned base = __builtin_amdgcn_readfirstlane(lds_addr(lbuf));
    const int span = nfloats / NWAVES;              // chunk sizes are multiples of NWAVES * 256 floats, except the tails below
    if ((nfloats % (NWAVES * 256)) == 0) {
        int i = wave * span;
        const int end = i + span;
        for (; i + 1024 <= end; i += 1024) dma_4k(gsrc + i + lane * 4, base + (unsigned)i * 4u);
        for (; i < end; i += 256) dma_1k(gsrc + i + lane * 4, base + (unsigned)i * 4u);
    } else {
        for (int i = wave * 256; i < nfloats; i += NWAVES * 256)
            dma_1k(gsrc + i + lane * 4, base + (unsigned)i * 4u);
    }
}

// Small parameters (biases, density / colour head weights) live in LDS behind the
// two weight buffers, so the main loop issues no compiler-tracked global loads.
constexpr int SMALL_FLOATS = PACKED_FLOATS - SM_BIAS;
constexpr int FIELD_LDS_FLOATS = 2 * CHUNK_FLOATS + SMALL_FLOATS;
__device__ inline void stage_small_from(const float* small_src, float* lds, int nthreads) {
    const f32x4* src = reinterpret_cast<const f32x4*>(small_src);
    f32x4* dst = reinterpret_cast<f32x4*>(lds + 2 * CHUNK_FLOATS);
    for (int i = threadIdx.x; i < SMALL_FLOATS / 4; i += nthreads) dst[i] = src[i];
}
__device__ inline void stage_small(const float* packed, float* lds) { stage_small_from(packed + SM_BIAS, lds, FIELD_WAVES * 64); }
// LDS address of packed[SM_x] after stage_small()
__device__ inline const float* small_ptr(const float* lds, int sm_offset) {
    return lds + 2 * CHUNK_FLOATS + (sm_offset - SM_BIAS);
}

// chunk sizes (32-bit words) of the two weight streams of the exact-fp32 kernels, in consumption order (MODE 0 forward, 1 backward)
template <int MODE>
__device__ constexpr int stream_chunk_words(int c) { return MODE == 0 ? fwd_chunk_floats(c) : bwd_chunk_floats(c); }

// Double-buffered weight stream.  acquire() = "chunk c has landed for every wave,
// nobody still reads the other buffer" -> start DMA of chunk c+1 -> hand out chunk c.
template <int MODE, int NWAVES>
struct WeightStreamT {
    const float* next_src;
    float* lds;
    int wave, lane;
    int c_next;         // stream index of the chunk whose DMA is issued next
    int buf;            // LDS buffer the chunk currently in flight lands in
    bool counted;       // wave-uniform: every lane of this wave issues the stores acquire<N> accounts for
    __device__ inline void start(const float* src, float* lds_, int wave_, int lane_, bool all_lanes_store = false) {
        lds = lds_; wave = wave_; lane = lane_; buf = 0;
        counted = __builtin_amdgcn_readfirstlane((int)__all(all_lanes_store)) != 0;
        const int nf = stream_chunk_words<MODE>(0);
        dma_chunk<NWAVES>(src, lds, nf, wave, lane);
        next_src = src + nf;
        c_next = 1;
    }
    // NPEND = number of vector-memory instructions (activation stores) this wave issued AFTER the DMA of the
    // chunk being acquired: vmcnt retires in order, so waiting for "at most NPEND outstanding" guarantees the DMA
    // has landed while those stores keep draining under the next chunk's MFMAs.  Must never exceed the real count.
    // skip_chunks (wave-uniform): full 64 KiB chunks of the stream to jump over before the prefetch issued here, i.e.
    // the chunk AFTER the one returned is c + 1 + skip_chunks (the folded feature layer's weights stay in the stream
    // for the benefit of one shared layout; the split-bf16 kernels never read them).
    template <int NPEND = 0>
    __device__ inline const float* acquire(int skip_chunks = 0) {
        // a wave with no (or predicated-off) stores has fewer entries in flight than NPEND: it must drain fully
        if (NPEND > 0 && counted) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPEND) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const float* cur = lds + buf * CHUNK_FLOATS;
        next_src += (size_t)skip_chunks * CHUNK_FLOATS;
        c_next += skip_chunks;
        const int nf = stream_chunk_words<MODE>(c_next);
        if (nf > 0) dma_chunk<NWAVES>(next_src, lds + (buf ^ 1) * CHUNK_FLOATS, nf, wave, lane);
        next_src += nf;
        ++c_next;
        buf ^= 1;
        return cur;
    }
};
template <bool FWD>
using WeightStream = WeightStreamT<FWD ? 0 : 1, FIELD_WAVES>;

// acc[nb] (16x16 block nb of the transposed output) += A(lds) * b over KS k-steps.
// b[BOFF + s] is this lane's B operand of k-step s (static indices -> registers).
template <int NB, int KS, int BOFF, int NREG>
__device__ inline void mma_chunk(f32x4 (&acc)[NB], const float (&b)[NREG], const float* lbuf, int lane) {
    constexpr int G = NB / 4;
    const f32x4* a = reinterpret_cast<const f32x4*>(lbuf) + lane;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const f32x4 w = a[(s * G + g) * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[4 * g + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j], b[BOFF + s], acc[4 * g + j], 0, 0, 0);
        }
    }
}

template <int NB>
__device__ inline void load_bias(f32x4 (&acc)[NB], const float* bias, int q) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = *reinterpret_cast<const f32x4*>(bias + 16 * nb + 4 * q);
}

// ReLU sign bits of this lane's NV post-activation values -> act.mask[layer][p][q]
template <int NV>
__device__ inline void save_mask(float* mask_base, int layer, size_t P, size_t p, int q, const float (&h)[NV]) {
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int i = 0; i < NV && i < 32; ++i) lo |= (h[i] > 0.0f ? 1u : 0u) << i;
#pragma unroll
    for (int i = 32; i < NV; ++i) hi |= (h[i] > 0.0f ? 1u : 0u) << (i - 32);
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    nt_store(reinterpret_cast<u32x2_t*>(mask_base) + ((size_t)layer * P + p) * 4 + q, u32x2_t{lo, hi});
}

// sum over the 4 lane quarters holding the same point
__device__ inline float quarter_sum(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// exact 2^k as float
__device__ inline float pow2f(int k) { return __int_as_float((127 + k) << 23); }

// xyz encoding registers of lane quarter q (slot map: encslot)
__device__ inline void encode_xyz(float (&e)[16], float x0, float x1, float x2, int q) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int i = q + 4 * m;
        if (m < 7 || q < 2) {
            const int dim = i % 3, fr = i / 3;
            const float xv = dim == 0 ? x0 : (dim == 1 ? x1 : x2);
            float sn, cs;
            sincosf(xv * pow2f(fr), &sn, &cs);
            e[2 * m] = sn;
            e[2 * m + 1] = cs;
        } else {
            e[2 * m] = q == 2 ? x0 : x2;
            e[2 * m + 1] = q == 2 ? x1 : 0.0f;
        }
    }
}
// dir encoding registers (slot map: dirslot)
__device__ inline void encode_dir(float (&v)[7], float d0, float d1, float d2, int q) {
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const int i = q + 4 * m;
        const int dim = i % 3, fr = i / 3;
        const float xv = dim == 0 ? d0 : (dim == 1 ? d1 : d2);
        float sn, cs;
        sincosf(xv * pow2f(fr), &sn, &cs);
        v[2 * m] = sn;
        v[2 * m + 1] = cs;
    }
    v[6] = q == 0 ? d0 : (q == 1 ? d1 : (q == 2 ? d2 : 0.0f));
}

}  // namespace nerf
