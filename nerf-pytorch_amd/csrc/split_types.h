// The 16-bit element type of the three-term split  W x ~= W_hi x_hi + W_hi x_lo + W_lo x_hi  (hi = T(v), lo = T(v - hi)),
// as a compile-time policy of the ring kernels (field_ring.h), the streaming weight-gradient GEMM (wgrad1_kernel) and the
// fragment repack:
//
//   SplitBF16 ("bf16x3", rounds 1-3): T = bfloat16.  8 + 8 significant bits per operand, ~2^-17 per product; fp32's exponent
//       range, so no operand can overflow or underflow.  The saved hi words (operands of the weight-gradient GEMM) carry 8 bits.
//   SplitF16  ("fp16x3", round 4):    T = IEEE half.  11 + 11 bits, ~2^-22 per product -- fp32-class -- at the same MFMA
//       rate (v_mfma_f32_16x16x32_f16 / 32x32x16_f16); the saved hi words carry 11 bits (weight-gradient operand rounding 2^-12
//       instead of 2^-9).  Measured on MI355X (tools/probe/f16_probe.py -> profiles/r04_f16_probe.txt): the f16 MFMAs take
//       SUBNORMAL operands exactly (no flush), v_cvt_pk_f16_f32 rounds to nearest even incl. subnormal results, and
//       v_fma_mix_f32 forms v - f32(hi) exactly, so hi + lo represents v to max(2^-23 |v|, 2^-25) absolute.  Range: |v| must
//       stay below 65520 (activations and weights of a NeRF are O(1..100)); an overflow becomes inf -> NaN in `raw`, loudly.
//       Deltas are scaled by an exact power of two per launch (delta_amax_kernel) because upstream gradients are ~1e-6.
//       The split itself is 4 VALU operations per value pair (cvt_pk, 2 x fma_mix, cvt_pk) against 6 for bf16.
#pragma once
#include "field_device.h"

// TIMING-ONLY ablations of the paired 16-bit stores (tools/exp_save_ablation.sh, round 6; results are wrong for every value but 0):
//   0 = product; 1 = pack + DPP swap + v_perm, no store; 2 = store of the lane's own word, no DPP / v_perm; 3 = neither
#ifndef NERF_ABL_SAVE
#define NERF_ABL_SAVE 0
#endif

namespace nerf {

// one paired store under the ablation switch: `own` = this lane's packed word, sel = its v_perm selector
template <typename Store>
__device__ __forceinline__ void paired_store(unsigned own, unsigned sel, Store store) {
    if constexpr (NERF_ABL_SAVE == 3) { asm volatile("" ::"v"(own)); return; }
    unsigned word = own;
    if constexpr (NERF_ABL_SAVE != 2) {
        const unsigned nbr = (unsigned)__builtin_amdgcn_mov_dpp((int)own, 0xB1, 0xF, 0xF, true);
        word = __builtin_amdgcn_perm(nbr, own, sel);
    }
    if constexpr (NERF_ABL_SAVE == 1) { asm volatile("" ::"v"(word)); return; }
    store(word);
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct SplitBF16 {
    static constexpr int F16 = 0;
    // (the packed conversion is written as one v_cvt_pk_bf16_f32: from the C++ form hipcc derives the low half's float with a
    // second conversion of v0 alone -- one VALU operation more per pair)
    __device__ static __forceinline__ unsigned cvt_pk(float a, float b) {
        unsigned r;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    // one (hi, lo) word pair of a B operand: values (v0, v1) -> 16-bit x 2 hi word, word of the remainders
    __device__ static __forceinline__ void split_pair(float v0, float v1, unsigned& hi, unsigned& lo) {
        const unsigned h = cvt_pk(v0, v1);
        hi = h;
        lo = cvt_pk(v0 - __uint_as_float(h << 16), v1 - __uint_as_float(h & 0xffff0000u));
    }
    __device__ static __forceinline__ unsigned short cvt1(float a) { return __builtin_bit_cast(unsigned short, (__bf16)a); }
    __device__ static __forceinline__ float relu(float x) { return fmaxf(x, 0.0f); }
    __device__ static __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {       // 16 points / wave
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {     // 32 points / wave
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ float sum8(u32x4 w) {         // fp32 sum of a fragment's eight values
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += __uint_as_float(w[i] << 16) + __uint_as_float(w[i] & 0xffff0000u);
        return s;
    }
};

struct SplitF16 {
    static constexpr int F16 = 1;
    __device__ static __forceinline__ unsigned cvt_pk(float a, float b) {          // round to nearest even, low half = a
        unsigned r;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    __device__ static __forceinline__ void split_pair(float v0, float v1, unsigned& hi, unsigned& lo) {
        const unsigned h = cvt_pk(v0, v1);
        float l0, l1;       // v - f32(h.half): one fused multiply-add reading the half directly (exact)
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(h), "v"(v0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(h), "v"(v1));
        hi = h;
        lo = cvt_pk(l0, l1);
    }
    __device__ static __forceinline__ unsigned short cvt1(float a) { return __builtin_bit_cast(unsigned short, (_Float16)a); }
    // ReLU that PROPAGATES NaN (IEEE-754-2019 maximum; v_max_f32 returns its non-NaN operand): an activation beyond fp16's range
    // becomes inf in the next operand split and inf - inf = NaN in the contraction -- that NaN must reach `raw`, not turn into a
    // zero activation.  Identical to fmaxf(x, 0) for every non-NaN x; one VALU operation.
    __device__ static __forceinline__ float relu(float x) {
        float r;
        asm("v_maximum3_f32 %0, %1, 0, 0" : "=v"(r) : "v"(x));
        return r;
    }
    __device__ static __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ float sum8(u32x4 w) {
        // (each word is taken apart as an integer first: __builtin_bit_cast of the vector ELEMENT w[i] to a half2 vector inside
        // the unrolled loop made hipcc 7.2 read w[0] four times -- bias gradients off by O(1), caught by the fp64 comparison)
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned u = w[i];
            s += (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu)) + (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16));
        }
        return s;
    }
};

// 32-point tiles of 16-bit elements written by lane PAIRS (adjacent points): both lanes pack their values (r, r+1) -- adjacent rows
// of the tile --, swap the word with the neighbour (DPP quad_perm [1,0,3,2]) and select with one v_perm_b32: the even lane holds row R
// of both points, the odd lane row R+1, so one dword store carries two values and an instruction writes two full 128-byte lines.
// Every lane of the wave takes part (DPP) and the stores are UNCONDITIONAL (a per-lane predicate is an exec-mask branch -- and a
// basic-block boundary for the scheduler -- per store: measured 6 % of the dgrad kernel): the caller passes a tile that exists.
template <typename SP, int OB0, int NOB, int NV, bool LO = false>
__device__ __forceinline__ void store_tile16_pair(unsigned short* tile_base, int lane, const float (&v)[NV]) {
    const unsigned odd = (unsigned)lane & 1u;
    const unsigned sel = odd ? 0x03020706u : 0x05040100u;
    unsigned* base = reinterpret_cast<unsigned*>(tile_base) + ((lane >> 5) * 4 + (int)odd) * 16 + ((lane & 31) >> 1);
#pragma unroll
    for (int ob = OB0; ob < OB0 + NOB; ++ob)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            unsigned own;
            if constexpr (LO) {         // the remainders lo = T(v - hi) of the same pair (two-word saves)
                unsigned hi;
                SP::split_pair(v[16 * ob + r], v[16 * ob + r + 1], hi, own);
            } else own = SP::cvt_pk(v[16 * ob + r], v[16 * ob + r + 1]);
            paired_store(own, sel, [&](unsigned word) __attribute__((always_inline)) { nt_store(base + (32 * ob + (r & 3) + 8 * (r >> 2)) * 16, word); });
        }
}

// host-side value of the same split (the repack kernel and the CPU tests' emulation agree on it)
template <typename SP> __device__ inline unsigned short split_hi(float x) { return SP::cvt1(x); }
template <typename SP> __device__ inline unsigned short split_lo(float x) {
    const unsigned short h = SP::cvt1(x);
    float hf;
    if constexpr (SP::F16) hf = (float)__builtin_bit_cast(_Float16, h);
    else hf = __uint_as_float((unsigned)h << 16);
    return SP::cvt1(x - hf);
}

}  // namespace nerf
