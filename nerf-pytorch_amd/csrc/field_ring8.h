// Reduced-precision product class for INFERENCE ("fp16 main term + fp8 correction terms"; set_precision("fp16_fp8c")):
//
//     W x  ~=  W_hi16 x_hi16   (v_mfma_f32_16x16x32_f16, K = 32 per instruction)
//            + W_hi8  x_lo8    (v_mfma_scale_f32_16x16x128_f8f6f4, fp8 e4m3 x fp8 e4m3, K = 128 per instruction)
//            + W_lo8  x_hi8
//
// hi16 = fp16(v), lo = v - hi16 (fp32, exact); the 8-bit parts are e4m3 roundings of power-of-two multiples of hi16 / lo: the
// correction terms are 2^-11 of the product and need 4 significant bits, which is what e4m3 has.  ~2^-15 per product (three-term
// fp16 split: 2^-22) at 2 instead of 3 MFMA-equivalents (a K = 128 fp8 MFMA takes the time of two K = 32 fp16 MFMAs).  Admitted
// for inference by the north-star gate on both fixtures with >= 18x margin and >= 93 dB against the reference's image
// (tools/analysis_accuracy_classes.py, class fp16+f8c-kernel; profiles/r04_accuracy_classes.md); never used for training.
//   Scales (all exact powers of two): activations x_hi8 = e4m3(2 x_hi16), x_lo8 = e4m3(2^12 x_lo) -- fixed, defined for |x| < 224
//   (v_cvt_scalef32_pk_fp8_f32 returns NaN beyond 448, so a larger activation is LOUD); weights per matrix and part from their
//   largest magnitude at pack time (weight_scale_kernel: max -> [128, 256)).  The MFMA's E8M0 scale operands undo them.
//
// Same weight RING, same unit count and positions as the three-term stream (field_ring.h): only the 256-wide contractions
// (layers 1..7 and the trunk part of the view branch) change their units' CONTENT and consumption; the narrow ones (xyz / direction
// encodings: 7 % of the units) stay three-term fp16 units.  Per T = 128 contraction slots and NG groups of 4 output blocks the
// 4 NG units come KIND-major:  [main k-steps 4T, 4T+1] x NG | [main k-steps 4T+2, 4T+3] x NG | [W_hi8] x NG | [W_lo8] x NG,
// so that every B operand (8 registers) is needed by NG consecutive units only and the next one is built in their shadow.
// A unit is two HALF sets of four 16-byte fragments (frag f = 2 j + half, the (hi, lo) addressing of the three-term units):
//   main unit: half = k-step of the pair, j = output block;     fp8 unit: half h = blocks (2h, 2h+1), j = 2 (block & 1) + 16-byte part.
// Three fragment register sets rotate with an 8-MFMA lead: phase 1 of unit u (half 0) requests half 0 of unit u+1, phase 2 requests
// its half 1.
#pragma once
#include "field_ring.h"

namespace nerf {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

constexpr float X_HI8_INV_SCALE = 0.5f;             // x_hi8 = e4m3(x_hi16 / 0.5)
constexpr float X_LO8_INV_SCALE = 1.0f / 4096.0f;   // x_lo8 = e4m3(x_lo / 2^-12)
constexpr int X_HI8_SCALE_BYTE = 127 - 1;           // E8M0 operand of the MFMA: multiplies by 2^(byte - 127)
constexpr int X_LO8_SCALE_BYTE = 127 - 12;
constexpr int N_RED_MATRICES = 8;                   // layers 1..7 (their 256 hidden inputs) and the folded view matrix W'

// two e4m3 bytes of (a, b) / scale into the low (HI = false) or high half of `old`
template <bool HI>
__device__ __forceinline__ unsigned pk2_fp8(unsigned old, float a, float b, float inv_scale) {
    return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(__builtin_bit_cast(i16x2, old), a, b, inv_scale, HI));
}
__device__ __forceinline__ f32x4 mfma8(const u32x4& a0, const u32x4& a1, const unsigned (&b)[8], f32x4 c, int scale_a, int scale_b) {
    const i32x8 A = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
    const i32x8 B = {(int)b[0], (int)b[1], (int)b[2], (int)b[3], (int)b[4], (int)b[5], (int)b[6], (int)b[7]};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, c, 0 /* A: fp8 e4m3 */, 0 /* B: fp8 e4m3 */, 0, scale_a, 0, scale_b);
}
// fragment register set I of the three (compile-time selection: no pointer tables, the sets must stay in registers)
template <int I>
__device__ __forceinline__ Frag& pick_set(Frag& a, Frag& b, Frag& c) {
    if constexpr (I == 0) return a;
    else if constexpr (I == 1) return b;
    else return c;
}
__device__ __forceinline__ f32x4 mfma16f(const u32x4& a, const unsigned* b, f32x4 c) {
    const u32x4 B = {b[0], b[1], b[2], b[3]};
    return SplitF16::mfma(a, B, c);
}

// NU units of one 256-slot contraction (NG = NU / 8 groups of four output blocks; T = 0, 1), B operand = the lane's 64 values v[].
// sa_hi / sa_lo: E8M0 scale bytes of this matrix's W_hi8 / W_lo8 (wave-uniform).  On entry `fa` holds half 0 of the first unit
// (requested by the predecessor, as ring_units leaves it); on exit `fa` holds half 0 of the unit after the last one.
template <int NU, int NW, int NB>
__device__ __forceinline__ void ring_units8(WeightRingT<NW>& ring, Frag& fa, Frag& fb, Frag& fl, f32x4 (&acc)[NB], const float (&v)[64], int sa_hi, int sa_lo) {
    constexpr int NG = NU / 8;
    static_assert(NU == 32 || NU == 16, "16 or 8 output blocks, K = 256");
    static_assert(NB == 4 * NG, "accumulators");
    unsigned op[2][8];                  // B operands: kind k uses op[k & 1]; the next kind's is built meanwhile
    // ---- operand builders (pair = two consecutive values; compile-time indices only)
    auto main_pair = [&](auto kc, auto tc, auto pc) __attribute__((always_inline)) {        // pair pc (0..7) of main kind kc of T tc
        constexpr int k = decltype(kc)::value, t = decltype(tc)::value, pr = decltype(pc)::value;
        unsigned w = SplitF16::cvt_pk(v[32 * t + 16 * k + 2 * pr], v[32 * t + 16 * k + 2 * pr + 1]);
        asm volatile("" : "+v"(w));         // pins the conversion HERE (pure code is otherwise sunk to its use, units later)
        op[k & 1][pr] = w;
    };
    auto lo_pair = [&](auto tc, auto pc) __attribute__((always_inline)) {                   // pair pc (0..15) of x_lo8 of T tc -> op[0]
        constexpr int t = decltype(tc)::value, pr = decltype(pc)::value;
        const float v0 = v[32 * t + 2 * pr], v1 = v[32 * t + 2 * pr + 1];
        // the fp16 pair of these two values is word pr of the main operands of this T (op[0] = values 0..15, dead as an MFMA operand
        // by now; op[1] = values 16..31, in use, read only).  Pair pr overwrites word pr >> 1 of op[0], which pair pr >> 1 has read.
        unsigned h;
        if constexpr (pr < 8) h = op[0][pr];
        else h = op[1][pr - 8];
        float l0, l1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(h), "v"(v0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(h), "v"(v1));
        unsigned w;
        if constexpr (pr & 1) w = pk2_fp8<true>(op[0][pr >> 1], l0, l1, X_LO8_INV_SCALE);
        else w = pk2_fp8<false>(0u, l0, l1, X_LO8_INV_SCALE);
        asm volatile("" : "+v"(w));
        op[0][pr >> 1] = w;
    };
    auto hi_pair = [&](auto tc, auto pc) __attribute__((always_inline)) {                   // pair pc (0..15) of x_hi8 of T tc -> op[1]
        constexpr int t = decltype(tc)::value, pr = decltype(pc)::value;
        const float v0 = v[32 * t + 2 * pr], v1 = v[32 * t + 2 * pr + 1];
        unsigned w;
        if constexpr (pr & 1) w = pk2_fp8<true>(op[1][pr >> 1], v0, v1, X_HI8_INV_SCALE);
        else w = pk2_fp8<false>(0u, v0, v1, X_HI8_INV_SCALE);
        asm volatile("" : "+v"(w));
        op[1][pr >> 1] = w;
    };
    static_for<0, 8>([&](auto pc) __attribute__((always_inline)) { main_pair(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, pc); });

    // half 1 of the first unit (its half 0 came with the predecessor): the one request with a 4-MFMA lead
    {
        const u32x4* p0 = ring.lane_ptr + ring.slot * (UNIT_WORDS / 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) fb.w[i] = p0[(2 * i + 1) * 64];
    }
    static_for<0, NU>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        constexpr int t = u / (4 * NG), k = (u % (4 * NG)) / NG, g = u % NG, pos = u % CHUNK_UNITS;
        constexpr bool last = u == NU - 1;
        Frag& h0 = pick_set<(2 * u) % 3>(fa, fb, fl);
        Frag& h1 = pick_set<(2 * u + 1) % 3>(fa, fb, fl);
        Frag& n0 = pick_set<(2 * u + 2) % 3>(fa, fb, fl);       // half 0 of the next unit: the third set
        Frag& n1 = pick_set<(2 * u) % 3>(fa, fb, fl);           // half 1 of the next unit: h0's set, free once phase 1 has issued
        if constexpr (pos == CHUNK_UNITS - 1) ring.template barrier<0>();
        const u32x4 *p, *pn;
        ring.unit_ptrs(p, pn);
        __builtin_amdgcn_sched_barrier(0);
        // what this unit builds in its shadow: `work(slot)` after MFMA number `slot` of the unit
        auto work = [&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            constexpr int tn = t + 1;                                       // (next T; exists iff t == 0)
            if constexpr (k == 0) {             // 8 MFMAs: the other main operand (8 pairs over NG units)
                if constexpr (s % (NG == 4 ? 4 : 2) == 0 && (NG == 4 ? s / 4 : s / 2) < 8 / NG)
                    main_pair(std::integral_constant<int, 1>{}, std::integral_constant<int, t>{}, std::integral_constant<int, (8 / NG) * g + (NG == 4 ? s / 4 : s / 2)>{});
            } else if constexpr (k == 1) {      // 8 MFMAs: x_lo8 (16 pairs over NG units)
                constexpr int per_unit = 16 / NG, every = 8 / per_unit;
                if constexpr (s % every == 0) lo_pair(std::integral_constant<int, t>{}, std::integral_constant<int, per_unit * g + s / every>{});
            } else if constexpr (k == 2) {      // 4 long MFMAs: x_hi8 (16 pairs over NG units)
                constexpr int per_slot = 16 / NG / 4;
                static_for<0, per_slot>([&](auto jc) __attribute__((always_inline)) {
                    hi_pair(std::integral_constant<int, t>{}, std::integral_constant<int, (16 / NG) * g + per_slot * s + decltype(jc)::value>{});
                });
            } else if constexpr (tn < 2) {      // 4 long MFMAs: main operand 0 of the next T (8 pairs over NG units)
                constexpr int per_unit = 8 / NG;
                if constexpr (s < per_unit) main_pair(std::integral_constant<int, 0>{}, std::integral_constant<int, tn>{}, std::integral_constant<int, per_unit * g + s>{});
            }
        };
        const unsigned (&b)[8] = op[k & 1];
        if constexpr (k < 2) {
            // ---- main unit: 4 blocks x 2 k-steps on the fp16 pipe
            __builtin_amdgcn_s_waitcnt(lgkmcnt_only(4));
            static_for<0, 4>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                acc[4 * g + i] = mfma16f(h0.w[i], &b[0], acc[4 * g + i]);
                asm volatile("" : "+v"(acc[4 * g + i]));        // (an accumulator is touched twice per unit and next 4 NG units later:
                n0.w[i] = pn[(2 * i) * 64];                     //  unpinned, the pure MFMA is sunk across the DMA branches to that use)
                work(ic);
                __builtin_amdgcn_sched_barrier(0);
            });
            __builtin_amdgcn_s_waitcnt(lgkmcnt_only(4));
            static_for<0, 4>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                acc[4 * g + i] = mfma16f(h1.w[i], &b[4], acc[4 * g + i]);
                asm volatile("" : "+v"(acc[4 * g + i]));
                if constexpr (!last) n1.w[i] = pn[(2 * i + 1) * 64];
                work(std::integral_constant<int, 4 + i>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        } else {
            // ---- fp8 unit: half h = blocks (2h, 2h + 1), fragments (block & 1) * 2 + 16-byte part
            const int sa = k == 2 ? sa_hi : sa_lo;
            constexpr int sb = (k == 2 ? X_LO8_SCALE_BYTE : X_HI8_SCALE_BYTE) * 0x01010101;
            __builtin_amdgcn_s_waitcnt(lgkmcnt_only(4));
            static_for<0, 2>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                acc[4 * g + i] = mfma8(h0.w[2 * i], h0.w[2 * i + 1], b, acc[4 * g + i], sa, sb);
                asm volatile("" : "+v"(acc[4 * g + i]));
                n0.w[2 * i] = pn[(4 * i) * 64];
                n0.w[2 * i + 1] = pn[(4 * i + 2) * 64];
                work(ic);
                __builtin_amdgcn_sched_barrier(0);
            });
            __builtin_amdgcn_s_waitcnt(lgkmcnt_only(4));
            static_for<0, 2>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                acc[4 * g + 2 + i] = mfma8(h1.w[2 * i], h1.w[2 * i + 1], b, acc[4 * g + 2 + i], sa, sb);
                asm volatile("" : "+v"(acc[4 * g + 2 + i]));
                if constexpr (!last) {
                    n1.w[2 * i] = pn[(4 * i + 1) * 64];
                    n1.w[2 * i + 1] = pn[(4 * i + 3) * 64];
                }
                work(std::integral_constant<int, 2 + i>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        ring.template fetch_after_unit<pos>();
    });
    // the next unit's half 0 sits in set (2 NU) % 3; the contract with ring_units / the next call is `fa`
    constexpr int end_set = (2 * NU) % 3;
    if constexpr (end_set != 0) {
        const Frag& src = pick_set<end_set>(fa, fb, fl);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa.w[i] = src.w[i];
    }
}

}  // namespace nerf
