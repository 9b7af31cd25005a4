// Body of the 16-point-per-wave three-term-split forward (bf16 or fp16 split: split_types.h) on the weight ring (see field_fwd_ring.hip): one workgroup tile of 128
// sample points through encode + 8x256 trunk + density head + folded view branch -> raw[.,4].  Shared by
// field_fwd16r_kernel (one tile per workgroup) and render_infer_kernel (render_fused.hip: the tiles of 16 rays, coarse and
// fine pass, inside one launch): same code, bit-identical results.
#pragma once
#include <type_traits>
#include "field_ring8.h"
#include "launchers.h"

namespace nerf {

struct FieldFwdRingArgs {
    const float* packed3;   // PACKED3_WORDS
    const float* rays;
    const float* z_vals;
    float* raw;
    float* act;             // nullable: act_layout3, rows in 16-point bf16 tiles (row16h order)
    int ray_stride, n_rays, S;
    // sample s of ray r reads z_vals[r * z_stride + s_off + s] and writes raw[(r * z_stride + s_off + s) * 4]: z_stride = S, s_off = 0
    // for a whole pass; S = 1, z_stride = the pass's sample count, s_off = z_stride - 1 evaluates ONLY every ray's last sample into
    // the pass's raw (the guard of the reduced inference class: launch_field_fwd16r_last).  Saving needs the whole-pass form.
    int z_stride, s_off;
    // raw row of that sample: (r * raw_stride + raw_off + s) * 4 -- the same numbers, unless a guard launch writes the last sample of
    // ANOTHER pass (the hierarchical pass keeps the last depth of the pass it refines: launch_field_fwd16r_last)
    int raw_stride, raw_off;
    int skip_last;          // 1: sample S - 1 of every ray is left unwritten (the guard launch evaluates it)
};

// units of the P16F stream in consumption order: L0 8 | L1..L4 4 x 32 | L5 40 | L6 L7 2 x 32 | (feature_linear 32: skipped)
// | folded view branch 9 k-steps x 2
constexpr int FWD16_UNITS_TRUNK = 8 + 4 * 32 + 40 + 2 * 32;
constexpr int FWD16_UNITS_SKIP = 32;
constexpr int FWD16_UNITS = FWD16_UNITS_TRUNK + 18;
static_assert(FWD16_UNITS_TRUNK % CHUNK_UNITS == 0, "the view branch starts on a chunk boundary");
static_assert((FWD16_UNITS_TRUNK + FWD16_UNITS_SKIP) * UNIT_WORDS == P16F_VIEWS, "unit arithmetic vs nerf_common.h");

// One workgroup tile (8 waves x 16 points = 128 consecutive sample points, tile index `wg`) through the whole network.
// The workgroup must have passed a barrier since its last use of `lds` (kernel start, or the caller's own __syncthreads()).
// SP: the 16-bit type of the three-term split (split_types.h) -- of the products AND of the rows saved for the backward.
// RED (inference only, SP = SplitF16): the 256-wide contractions run as fp16 main term + fp8 correction terms (field_ring8.h) on a
// packed buffer made by the reduced repack (nerf_pack_params_split(split = 2)).
template <int SAVE, typename SP, bool RED = false>
__device__ __forceinline__ void field_fwd16r_tile(const FieldFwdRingArgs& a, float* lds, long wg) {
    // SAVE: 0 = inference; 2 = 16-bit rows (the hi words: 11 / 8 significant bits); 3 = hi AND lo words ("fp16x3w", round 6: the
    // weight-gradient GEMM then contracts two-word operands, X_hi d_hi + X_lo d_hi + X_hi d_lo) -- the lo words go to the mirror
    // of the layout at ActLayout3::lo
    static_assert(SAVE == 0 || SAVE == 2 || SAVE == 3, "inference, 16-bit rows, or (hi, lo) rows");
    constexpr bool TWO = SAVE == 3;
    static_assert(!RED || (SAVE == 0 && SP::F16), "the reduced products are an inference form of the fp16 split");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4;
    // point indices are 32-bit (the launchers refuse n_rays * S >= 2^31): the wave's 16-point tile number is wave-uniform (SGPR),
    // a lane adds its point, and ray / sample follow from ONE 32-bit division
    const unsigned P = (unsigned)a.n_rays * (unsigned)a.S;
    const unsigned tile16 = (unsigned)__builtin_amdgcn_readfirstlane((int)(wg * FIELD_WAVES + wave));
    const unsigned p_raw = tile16 * PTS_PER_WAVE + (unsigned)(lane & 15);
    const bool valid = p_raw < P;
    const unsigned p = valid ? p_raw : P - 1u;
    const unsigned ray = p / (unsigned)a.S;
    const unsigned si = p - ray * (unsigned)a.S;

    WeightRing ring;
    ring.start(a.packed3 + P16F, lds, wave, lane, FWD16_UNITS_TRUNK, FWD16_UNITS_SKIP, FWD16_UNITS);
    stage_small_ring(a.packed3 + P3_SMALL, lds, FIELD_WAVES * 64);

    const float* rp = a.rays + (size_t)ray * (size_t)a.ray_stride;
    const size_t zi = (size_t)ray * (size_t)a.z_stride + (size_t)(a.s_off + (int)si);
    const float z = a.z_vals[zi];
    const float x0 = rp[0] + rp[3] * z;
    const float x1 = rp[1] + rp[4] * z;
    const float x2 = rp[2] + rp[5] * z;
    float e[16];
    encode_xyz(e, x0, x1, x2, q);

    // ---- saving (layouts: nerf_common.h).  Every store address is a WAVE-UNIFORM 64-bit base (region, tile: scalar registers)
    // plus a 32-bit byte offset of the lane that does not change along the trunk -- no per-lane 64-bit pointers are carried
    // (the launchers refuse to save more than 2^26 points per launch, so the lane offsets fit 32 bits).
    ActLayout3 al{};
    const size_t layer_floats = region_words3(pad32((size_t)P), W);       // words of one 256-wide region of 16-bit rows (== al.h[1] - al.h[0])
    const unsigned odd = (unsigned)lane & 1u;
    const unsigned pair_sel = odd ? 0x03020706u : 0x05040100u;      // v_perm_b32 bytes of {neighbour word, own word}
    const unsigned lane_pair_bytes = 4u * (unsigned)((2 * q + (int)odd) * 8 + ((lane & 15) >> 1));
    const bool tile_ok = (size_t)tile16 * 16 < pad32((size_t)P);
    char* const act_bytes = reinterpret_cast<char*>(a.act);
    // rows (r0, r0 + 1) of block nb of this lane's point, paired with the neighbour point: one dword store (unconditional;
    // a wave whose tile lies beyond the padded range writes to the unused `feat` region)
    auto store_word = [&](size_t region, int F, int nb, int r0, unsigned own, bool lo_part = false) __attribute__((always_inline)) {
        char* tile_base = act_bytes + 4 * ((tile_ok ? region : al.feat) + (lo_part ? al.lo : (size_t)0))
                          + (tile_ok ? (size_t)tile16 * (size_t)(F * 32) : (size_t)0) + (size_t)((16 * nb + 4 * r0) * 32);
        paired_store(own, pair_sel, [&](unsigned word) __attribute__((always_inline)) { nt_store_saddr(tile_base, lane_pair_bytes, word); });
    };
    if (SAVE) {
        al = act_layout3((size_t)P, (size_t)a.n_rays, TWO);
        if (valid) {
            char* enc_tile = act_bytes + 4 * al.enc + (size_t)(tile16 >> 1) * (size_t)(64 * 32 * 2);     // the wave's 32-point tile (uniform)
            const unsigned pp2 = 2u * (p_raw & 31u);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int col = encslot(s, q);
                if (col >= 0) {
                    nt_store(reinterpret_cast<unsigned short*>(enc_tile + ((unsigned)col * 64u + pp2)), SP::cvt1(e[s]));
                    if (TWO) nt_store(reinterpret_cast<unsigned short*>(enc_tile + 4 * al.lo + ((unsigned)col * 64u + pp2)), split_lo<SP>(e[s]));
                }
            }
        }
    }

    const float* bias = ring_small_ptr(lds, SM_BIAS);
    f32x4 acc[16];
    float h[64];
    auto relu = [](float x) __attribute__((always_inline)) { return SP::relu(x); };
    auto take = [&]() {
#pragma unroll
        for (int nb = 0; nb < 16; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) h[4 * nb + r] = relu(acc[nb][r]);
    };
    // row stores guaranteed behind the last fetch part (one per unit, positions 3..6); none under the store-less timing ablations
    constexpr int NP = (SAVE && NERF_ABL_SAVE != 1 && NERF_ABL_SAVE != 3) ? (TWO ? 8 : 4) : 0;
    auto store_pair = [&](size_t region, int F, int nb, int r0, float v0, float v1) __attribute__((always_inline)) {
        if constexpr (TWO) {
            unsigned hi, lo;
            SP::split_pair(v0, v1, hi, lo);
            store_word(region, F, nb, r0, hi);
            store_word(region, F, nb, r0, lo, true);
        } else store_word(region, F, nb, r0, SP::cvt_pk(v0, v1));
    };
    auto no_store = [](auto, auto, const u32x4&, const u32x4&) __attribute__((always_inline)) {};
    // rows of the layer in h[] leave while the next contraction consumes them: unit (k-step kk, group gg) covers
    // block 2 kk + (gg >> 1), rows 2 (gg & 1), 2 (gg & 1) + 1 (row16h order)
    size_t row_region = 0;
    // (the bf16 values of rows (r0, r0 + 1) of block nb ARE word g of the B operand's hi fragment of k-step kk)
    // ... and so do the bits of their ReLU mask: two compares per unit in the shadow of its MFMAs instead of 64 at the layer's
    // end, where neither wave of the SIMD has an MFMA in flight (mw: the four mask words of the layer in h[], without the
    // lane-dependent shift)
    // Two registers hold the four words: word i sits in mw[i & 1] shifted left by 4 (i >> 1) -- a word uses only the low nibble of
    // each byte before the lane-dependent shift.
    unsigned mw[2] = {0u, 0u};
    auto mask_bits = [&](auto nbc, auto rc) __attribute__((always_inline)) {
        constexpr int nb = decltype(nbc)::value, r = decltype(rc)::value;
        unsigned b = h[4 * nb + r] > 0.0f ? 1u << (8 * (nb & 3) + r + 4 * (nb >> 3)) : 0u;
        asm volatile("" : "+v"(b));
        mw[(nb >> 2) & 1] |= b;
    };
    auto store_rows = [&](auto kk, auto gg, const u32x4& bhi, const u32x4& blo) __attribute__((always_inline)) {
        if (!SAVE) return;
        constexpr int nb = 2 * decltype(kk)::value + (decltype(gg)::value >> 1);
        constexpr int r0 = 2 * (decltype(gg)::value & 1);
        store_word(row_region, W, nb, r0, bhi[decltype(gg)::value]);
        if constexpr (TWO) store_word(row_region, W, nb, r0, blo[decltype(gg)::value], true);
        mask_bits(std::integral_constant<int, nb>{}, std::integral_constant<int, r0>{});
        mask_bits(std::integral_constant<int, nb>{}, std::integral_constant<int, r0 + 1>{});
    };
    auto finish_mask = [&](int layer) __attribute__((always_inline)) {     // the words save_mask16 builds, bit for bit
        if (!SAVE) return;
        unsigned w4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            w4[i] = ((mw[i & 1] >> (4 * (i >> 1))) & 0x0f0f0f0fu) << (4 * (q >> 1));
            w4[i] |= __shfl_xor(w4[i], 32);
        }
        if (valid && q < 2) {
            char* layer_base = act_bytes + 4 * al.mask + (size_t)layer * (size_t)P * 32;        // uniform
            nt_store_saddr(layer_base, p * 32u + (unsigned)q * 16u, u32x4{w4[0], w4[1], w4[2], w4[3]});
        }
        mw[0] = mw[1] = 0u;
    };

    ring.ready();
    Frag fa, fb, fl;
    ring.request_first(fa);
    // E8M0 scale bytes (W_hi8, W_lo8) of reduced matrix m, from the pads of the small parameters (pack8_kernel)
    auto red_scale = [&](int m, int part) __attribute__((always_inline)) {
        const unsigned char* sb = reinterpret_cast<const unsigned char*>(ring_small_ptr(lds, SM_BALPHA));
        return __builtin_amdgcn_readfirstlane((int)sb[(m < 6 ? 4 + 2 * m : 28 + 2 * (m - 6)) + part]);
    };

    // ---- layer 0: 63 -> 256 (2 k-steps of the xyz encoding)
    load_bias<16>(acc, bias, q);
    ring_units<SP, 8, 4, 0, true, 0>(ring, fa, fb, fl, acc, e, no_store);
    take();
    // ---- layers 1..7 (layer 5 contracts the xyz encoding first: skip connection).  Layer l writes the rows of layer l-1.
#pragma unroll 1
    for (int l = 1; l < D; ++l) {
        load_bias<16>(acc, bias + l * W, q);
        if (l == SKIP + 1) ring_units<SP, 8, 4, 0, false, 0>(ring, fa, fb, fl, acc, e, no_store);
        row_region = (size_t)(l - 1) * layer_floats;
        if constexpr (RED) ring_units8<32>(ring, fa, fb, fl, acc, h, red_scale(l - 1, 0), red_scale(l - 1, 1));
        else ring_units<SP, 32, 4, 0, false, NP>(ring, fa, fb, fl, acc, h, store_rows);
        finish_mask(l - 1);
        take();
    }
    // What is needed from here on is re-derived from the point and ray NUMBERS (two registers carried through the trunk) and from the
    // hardware's lane number, behind an opaque copy -- otherwise the view direction, the sample index, the output address and a
    // dozen lane-derived selections of the prologue stay live (or spill) along the trunk.
    unsigned ray_l = ray, p_l = p;
    int q_l = (int)(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) >> 4);
    asm volatile("" : "+v"(ray_l), "+v"(p_l), "+v"(q_l));
    const unsigned si_l = p_l - ray_l * (unsigned)a.S;
    // ---- density head: alpha_linear 256 -> 1 (VALU dot + quarter reduction)
    float sigma = 0.0f;
    {
        const float* wa = ring_small_ptr(lds, SM_WALPHA) + 4 * q_l;
#pragma unroll
        for (int nb = 0; nb < 16; ++nb) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(wa + 16 * nb);
#pragma unroll
            for (int r = 0; r < 4; ++r) sigma = fmaf(h[4 * nb + r], w[r], sigma);
        }
        sigma = quarter_sum(sigma) + ring_small_ptr(lds, SM_BALPHA)[0];
    }
    // ---- view branch on the trunk output (feature_linear folded: W', b'): [h7, enc(dir)] -> 128, ReLU
    float dv[8];
    {
        const float* rp_l = a.rays + (size_t)ray_l * (size_t)a.ray_stride;
        float v7[7];
        encode_dir(v7, rp_l[8], rp_l[9], rp_l[10], q_l);
#pragma unroll
        for (int i = 0; i < 7; ++i) dv[i] = v7[i];
        dv[7] = 0.0f;
    }
    if (SAVE && valid && si_l == 0) {
        float* dout = a.act + al.dir + (size_t)ray_l * 32;
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            const int col = dirslot(s, q_l);
            if (col >= 0) nt_store(dout + col, dv[s]);
        }
    }
    f32x4 av[8];
    load_bias<8>(av, ring_small_ptr(lds, SM_BVIEWS), q_l);
    {   // layer 7's rows leave under the 16 units of the trunk part: k-step kk = blocks 2 kk, 2 kk + 1, half of them per unit
        row_region = (size_t)(D - 1) * layer_floats;
        auto store_rows_v = [&](auto kk, auto gg, const u32x4& bhi, const u32x4& blo) __attribute__((always_inline)) {
            if (!SAVE) return;
            constexpr int nb = 2 * decltype(kk)::value + decltype(gg)::value;
            store_word(row_region, W, nb, 0, bhi[2 * decltype(gg)::value]);
            store_word(row_region, W, nb, 2, bhi[2 * decltype(gg)::value + 1]);
            if constexpr (TWO) {
                store_word(row_region, W, nb, 0, blo[2 * decltype(gg)::value], true);
                store_word(row_region, W, nb, 2, blo[2 * decltype(gg)::value + 1], true);
            }
            static_for<0, 4>([&](auto rc) __attribute__((always_inline)) { mask_bits(std::integral_constant<int, nb>{}, rc); });
        };
        if constexpr (RED) ring_units8<16>(ring, fa, fb, fl, av, h, red_scale(7, 0), red_scale(7, 1));
        else ring_units<SP, 16, 2, 0, false, NP>(ring, fa, fb, fl, av, h, store_rows_v);
        finish_mask(D - 1);
        ring_units<SP, 2, 2, 0, false, 0>(ring, fa, fb, fl, av, dv, no_store);
    }
    float hv[32];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[4 * nb + r] = relu(av[nb][r]);
    if (SAVE) {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
#pragma unroll
            for (int r = 0; r < 4; r += 2) store_pair(al.hv, WV, nb, r, hv[4 * nb + r], hv[4 * nb + r + 1]);
        unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) w[nb >> 2] |= (hv[4 * nb + r] > 0.0f ? 1u : 0u) << (8 * (nb & 3) + 4 * (q_l >> 1) + r);
#pragma unroll
        for (int i = 0; i < 2; ++i) w[i] |= __shfl_xor(w[i], 32);
        if (valid && q_l < 2) {
            char* layer_base = act_bytes + 4 * al.mask + (size_t)D * (size_t)P * 32;
            nt_store_saddr(layer_base, p_l * 32u + (unsigned)q_l * 16u, u32x4{w[0], w[1], w[2], w[3]});
        }
    }
    // ---- rgb_linear 128 -> 3
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    {
        const float* wr = ring_small_ptr(lds, SM_WRGB) + 4 * q_l;
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
        c0 = quarter_sum(c0) + ring_small_ptr(lds, SM_BRGB)[0];
        c1 = quarter_sum(c1) + ring_small_ptr(lds, SM_BRGB)[1];
        c2 = quarter_sum(c2) + ring_small_ptr(lds, SM_BRGB)[2];
    }
    if (valid && q_l == 0 && !(a.skip_last && (int)si_l == a.S - 1)) {
        const size_t ri = (size_t)ray_l * (size_t)a.raw_stride + (size_t)(a.raw_off + (int)si_l);
        *reinterpret_cast<f32x4*>(a.raw + ri * 4) = f32x4{c0, c1, c2, sigma};
    }
}

}  // namespace nerf
