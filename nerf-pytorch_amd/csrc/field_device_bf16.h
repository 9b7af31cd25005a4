// Device helpers of the split-bf16 ("bf16x3") field kernels (forward and dgrad).
#pragma once
#include <type_traits>
#include "field_device.h"

namespace nerf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ inline unsigned pack_bf16x2(float a, float b) {       // RNE, low half = a
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
// 8 fp32 values -> bf16 (hi, lo) B fragments
__device__ inline void split8(const float* v, u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned h = pack_bf16x2(v[2 * i], v[2 * i + 1]);
        hi[i] = h;
        lo[i] = pack_bf16x2(v[2 * i] - __uint_as_float(h << 16), v[2 * i + 1] - __uint_as_float(h & 0xffff0000u));
    }
}
__device__ inline f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// one k-step for NB output blocks: acc[nb] += A_hi*B_hi + A_hi*B_lo + A_lo*B_hi
// kbase = LDS address of this k-step's fragments (u32x4 units), lane-linear: ((nb*2 + hl)*64 + lane)
template <int NB>
__device__ inline void mma3_kstep(f32x16 (&acc)[NB], const u32x4 bhi, const u32x4 blo, const u32x4* kbase) {
#pragma unroll
    for (int g = 0; g < NB; g += 4) {
        u32x4 ahi[4], alo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { ahi[i] = kbase[((g + i) * 2) * 64]; alo[i] = kbase[((g + i) * 2 + 1) * 64]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[g + i] = mfma_bf16(ahi[i], bhi, acc[g + i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[g + i] = mfma_bf16(ahi[i], blo, acc[g + i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[g + i] = mfma_bf16(alo[i], bhi, acc[g + i]);
    }
}

// KS k-steps whose B operands are v[VOFF + 8*s .. +7] (fp32, already activated)
template <int NB, int KS, int VOFF, int NV>
__device__ inline void mma3_chunk(f32x16 (&acc)[NB], const float (&v)[NV], const float* lbuf, int lane) {
    const u32x4* a = reinterpret_cast<const u32x4*>(lbuf) + lane;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        u32x4 bhi, blo;
        split8(&v[VOFF + 8 * s], bhi, blo);
        mma3_kstep<NB>(acc, bhi, blo, a + s * (NB * 2 * 64));
    }
}

// ---- single-term variants for the mixed-precision backward ("bf16x3 forward + bf16 backward"): W_hi * x_hi only
__device__ inline u32x4 pack8(const float* v) {
    return u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
}
// hi-only fragment stream (nerf_common.h, P1B): k-step = NB blocks x 64 lanes x 16 B
template <int NB>
__device__ inline void mma1_kstep(f32x16 (&acc)[NB], const u32x4 bhi, const u32x4* kbase) {
#pragma unroll
    for (int g = 0; g < NB; g += 4) {
        u32x4 ahi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ahi[i] = kbase[(g + i) * 64];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[g + i] = mfma_bf16(ahi[i], bhi, acc[g + i]);
    }
}
// KS k-steps starting at k-step K0 of the chunk in lbuf, B operands v[VOFF + 8*s ..]
template <int NB, int KS, int K0, int VOFF, int NV>
__device__ inline void mma1_chunk(f32x16 (&acc)[NB], const float (&v)[NV], const float* lbuf, int lane) {
    const u32x4* a = reinterpret_cast<const u32x4*>(lbuf) + lane + K0 * (NB * 64);
#pragma unroll
    for (int s = 0; s < KS; ++s) mma1_kstep<NB>(acc, pack8(&v[VOFF + 8 * s]), a + s * (NB * 64));
}

// ---- 16-point-per-wave variant (v_mfma_f32_16x16x32_bf16): the fp32 kernels' register shape (64 accumulators + 64
// activations per lane, 2 waves / SIMD) on the bf16 pipe.  One k-step = 32 contraction slots = the lane's values
// v[8*s .. 8*s+7]; LDS fragments: ((nb*2 + hl)*64 + lane) x 16 B per k-step (nerf_common.h, P16F)
__device__ inline f32x4 mfma16_bf16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// fragments of G blocks are requested together (2 x G x 4 registers)
template <int NB, int G = 4>
__device__ inline void mma16_kstep(f32x4 (&acc)[NB], const u32x4 bhi, const u32x4 blo, const u32x4* kbase) {
#pragma unroll
    for (int g = 0; g < NB; g += G) {
        u32x4 ahi[G], alo[G];
#pragma unroll
        for (int i = 0; i < G; ++i) { ahi[i] = kbase[((g + i) * 2) * 64]; alo[i] = kbase[((g + i) * 2 + 1) * 64]; }
#pragma unroll
        for (int i = 0; i < G; ++i) acc[g + i] = mfma16_bf16(ahi[i], bhi, acc[g + i]);
#pragma unroll
        for (int i = 0; i < G; ++i) acc[g + i] = mfma16_bf16(ahi[i], blo, acc[g + i]);
#pragma unroll
        for (int i = 0; i < G; ++i) acc[g + i] = mfma16_bf16(alo[i], bhi, acc[g + i]);
    }
}
// one k-step whose four MFMA groups (4 blocks = 12 MFMAs each) are separated by `between(integral_constant<int, group>)`:
// the saving forward issues its row stores there, so that they leave as a steady stream (2 stores per 12 MFMAs)
// instead of a burst.  Everything is indexed at compile time (register arrays must never be indexed dynamically).
template <int GI, int NB>
__device__ __forceinline__ void mma16_group(f32x4 (&acc)[NB], const u32x4 bhi, const u32x4 blo, const u32x4* kbase) {
    constexpr int g = 4 * GI;
    u32x4 ahi[4], alo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { ahi[i] = kbase[((g + i) * 2) * 64]; alo[i] = kbase[((g + i) * 2 + 1) * 64]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[g + i] = mfma16_bf16(ahi[i], bhi, acc[g + i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[g + i] = mfma16_bf16(ahi[i], blo, acc[g + i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[g + i] = mfma16_bf16(alo[i], bhi, acc[g + i]);
}
template <int VOFF, int NV, typename F>
__device__ __forceinline__ void mma16_kstep_with(f32x4 (&acc)[16], const float (&v)[NV], const float* kstep_base, int lane, F between) {
    const u32x4* kbase = reinterpret_cast<const u32x4*>(kstep_base) + lane;
    u32x4 bhi, blo;
    split8(&v[VOFF], bhi, blo);
    mma16_group<0>(acc, bhi, blo, kbase); between(std::integral_constant<int, 0>{});
    mma16_group<1>(acc, bhi, blo, kbase); between(std::integral_constant<int, 1>{});
    mma16_group<2>(acc, bhi, blo, kbase); between(std::integral_constant<int, 2>{});
    mma16_group<3>(acc, bhi, blo, kbase); between(std::integral_constant<int, 3>{});
}

template <int NB, int KS, int VOFF, int NV>
__device__ inline void mma16_chunk(f32x4 (&acc)[NB], const float (&v)[NV], const float* lbuf, int lane) {
    const u32x4* a = reinterpret_cast<const u32x4*>(lbuf) + lane;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        u32x4 bhi, blo;
        split8(&v[VOFF + 8 * s], bhi, blo);
        mma16_kstep<NB>(acc, bhi, blo, a + s * (NB * 2 * 64));
    }
}

template <int NB>
__device__ inline void load_bias3(f32x16 (&acc)[NB], const float* bias, int half) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(bias + 32 * nb + 8 * g + 4 * half);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[nb][4 * g + r] = b[r];
        }
}

// Saved rows go to 32-point tiles, feature-major (nerf_common.h, ActLayout3).  tp = the lane's slot of feature 4*half
// in its wave's tile: region + tile * F * 32 + half * 128 + (lane & 31).  Lane value i (0..16*NB-1) is feature
// 32*(i>>4) + d32row(i&15, half); one store instruction writes one feature of the wave's 32 points per lane half
// = two full 128-byte lines.
template <int OB0, int NOB, int NV>
__device__ inline void store_tile3(float* tp, const float (&v)[NV]) {
#pragma unroll
    for (int ob = OB0; ob < OB0 + NOB; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) nt_store(tp + (32 * ob + (r & 3) + 8 * (r >> 2)) * 32, v[16 * ob + r]);
}
// the same tile layout with bf16 (RNE) elements: what the mixed-precision backward reads.  tp = (bf16*)region + the
// same element offset as store_tile3
template <int OB0, int NOB, int NV>
__device__ inline void store_tile3h(__bf16* tp, const float (&v)[NV]) {
#pragma unroll
    for (int ob = OB0; ob < OB0 + NOB; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) nt_store(tp + (32 * ob + (r & 3) + 8 * (r >> 2)) * 32, (__bf16)v[16 * ob + r]);
}
// The same bf16 tiles written by lane PAIRS (adjacent points): both lanes pack their values (r, r+1) — adjacent rows of the
// tile —, swap the word with the neighbour (DPP quad_perm [1,0,3,2]) and select with one v_perm_b32: the even lane holds
// row R of both points, the odd lane row R+1, so one dword store carries two values, the 16 even lanes of a half write
// the whole 64-byte row R, the odd lanes row R+1, and an instruction writes two full 128-byte lines.  Half the store
// instructions of store_tile3h.  Every lane of the wave takes part (DPP) and the stores are UNCONDITIONAL: a per-lane
// predicate is an exec-mask branch — and a basic-block boundary for the scheduler — per store (measured: 6 % of the
// dgrad kernel, also when only one branch per 16 stores is left).  The caller passes a tile that exists: padding
// points of a ragged last tile are written (the weight-gradient GEMM masks points >= P), and a wave whose whole tile
// lies beyond the padded point range is pointed at a dump tile (an unused region of the same buffer).
// tile_base = region + tile * F * 32 (no lane offset).
template <int OB0, int NOB, int NV>
__device__ __forceinline__ void store_tile3h_pair(__bf16* tile_base, int lane, const float (&v)[NV]) {
    const unsigned odd = (unsigned)lane & 1u;
    const unsigned sel = odd ? 0x03020706u : 0x05040100u;
    unsigned* base = reinterpret_cast<unsigned*>(tile_base) + ((lane >> 5) * 4 + (int)odd) * 16 + ((lane & 31) >> 1);
#pragma unroll
    for (int ob = OB0; ob < OB0 + NOB; ++ob)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const unsigned own = pack_bf16x2(v[16 * ob + r], v[16 * ob + r + 1]);
            const unsigned nbr = (unsigned)__builtin_amdgcn_mov_dpp((int)own, 0xB1, 0xF, 0xF, true);
            nt_store(base + (32 * ob + (r & 3) + 8 * (r >> 2)) * 16, __builtin_amdgcn_perm(nbr, own, sel));
        }
}
constexpr int STORES_PER_QUARTER3 = 32;      // store_tile3<2*PART, 2>: a quarter of a 256-feature row set (paired bf16: 16)
// ReLU sign bits of the lane's NV values -> act.mask[layer][p][half] (4 words; NV <= 128)
template <int NV>
__device__ inline void save_mask3(float* mask_base, int layer, size_t P, size_t p, int half, const float (&v)[NV]) {
    unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < NV; ++i) w[i >> 5] |= (v[i] > 0.0f ? 1u : 0u) << (i & 31);
    nt_store(reinterpret_cast<u32x4*>(mask_base) + ((size_t)layer * P + p) * 2 + half, u32x4{w[0], w[1], w[2], w[3]});
}

__device__ inline float half_sum(float v) { return v + __shfl_xor(v, 32); }

}  // namespace nerf
