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
    static_assert(SAVE == 0 || SAVE == 2, "inference or 16-bit rows");
    static_assert(!RED || (SAVE == 0 && SP::F16), "the reduced products are an inference form of the fp16 split");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4;
    const long P = (long)a.n_rays * a.S;
    const long p_raw = (wg * FIELD_WAVES + wave) * PTS_PER_WAVE + (lane & 15);
    const bool valid = p_raw < P;
    const long p = valid ? p_raw : P - 1;
    const int ray = (int)(p / a.S);

    WeightRing ring;
    ring.start(a.packed3 + P16F, lds, wave, lane, FWD16_UNITS_TRUNK, FWD16_UNITS_SKIP, FWD16_UNITS);
    stage_small_ring(a.packed3 + P3_SMALL, lds, FIELD_WAVES * 64);

    const float* rp = a.rays + (long)ray * a.ray_stride;
    const long si = p - (long)ray * a.S;
    const long zi = (long)ray * a.z_stride + a.s_off + si;
    const long ri = (long)ray * a.raw_stride + a.raw_off + si;
    const float z = a.z_vals[zi];
    const float x0 = rp[0] + rp[3] * z;
    const float x1 = rp[1] + rp[4] * z;
    const float x2 = rp[2] + rp[5] * z;
    const float vd0 = rp[8], vd1 = rp[9], vd2 = rp[10];
    float e[16];
    encode_xyz(e, x0, x1, x2, q);

    // ---- saving (layouts: nerf_common.h)
    ActLayout3 al{};
    const size_t tile = (size_t)(p_raw >> 5);
    const int pp = (int)(p_raw & 31);
    const size_t layer_floats = pad32((size_t)P) * W;
    const unsigned tile16 = (unsigned)__builtin_amdgcn_readfirstlane((int)((wg * FIELD_WAVES + wave)));
    const unsigned odd = (unsigned)lane & 1u;
    const unsigned pair_sel = odd ? 0x03020706u : 0x05040100u;      // v_perm_b32 bytes of {neighbour word, own word}
    const unsigned lane_pair_off = (unsigned)((2 * q + (int)odd) * 8 + ((lane & 15) >> 1));
    const bool tile_ok = (size_t)tile16 * 16 < pad32((size_t)P);
    // rows (r0, r0 + 1) of block nb of this lane's point, paired with the neighbour point: one dword store (unconditional;
    // a wave whose tile lies beyond the padded range writes to the unused `feat` region)
    auto store_word = [&](size_t region, int F, int nb, int r0, unsigned own) __attribute__((always_inline)) {
        const unsigned nbr = (unsigned)__builtin_amdgcn_mov_dpp((int)own, 0xB1, 0xF, 0xF, true);
        const unsigned word = __builtin_amdgcn_perm(nbr, own, pair_sel);
        unsigned* tile_base = reinterpret_cast<unsigned*>(a.act + (tile_ok ? region : al.feat))
                              + (tile_ok ? (size_t)tile16 * (size_t)(F * 8) : (size_t)0);
        nt_store(tile_base + (16 * nb + 4 * r0) * 8 + lane_pair_off, word);
    };
    if (SAVE) {
        al = act_layout3((size_t)P, (size_t)a.n_rays);
        if (valid) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int col = encslot(s, q);
                if (col >= 0) nt_store(reinterpret_cast<unsigned short*>(a.act + al.enc) + tile * (size_t)(64 * 32) + (size_t)col * 32 + pp, SP::cvt1(e[s]));
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
    constexpr int NP = SAVE ? 4 : 0;        // row stores guaranteed behind the last fetch part (one per unit, positions 3..6)
    auto store_pair = [&](size_t region, int F, int nb, int r0, float v0, float v1) __attribute__((always_inline)) {
        store_word(region, F, nb, r0, SP::cvt_pk(v0, v1));
    };
    auto no_store = [](auto, auto, const u32x4&) __attribute__((always_inline)) {};
    // rows of the layer in h[] leave while the next contraction consumes them: unit (k-step kk, group gg) covers
    // block 2 kk + (gg >> 1), rows 2 (gg & 1), 2 (gg & 1) + 1 (row16h order)
    size_t row_region = 0;
    // (the bf16 values of rows (r0, r0 + 1) of block nb ARE word g of the B operand's hi fragment of k-step kk)
    // ... and so do the bits of their ReLU mask: two compares per unit in the shadow of its MFMAs instead of 64 at the layer's
    // end, where neither wave of the SIMD has an MFMA in flight (mw: the four mask words of the layer in h[], without the
    // lane-dependent shift)
    unsigned mw[4] = {0u, 0u, 0u, 0u};
    auto mask_bits = [&](auto nbc, auto rc) __attribute__((always_inline)) {
        constexpr int nb = decltype(nbc)::value, r = decltype(rc)::value;
        unsigned b = h[4 * nb + r] > 0.0f ? 1u << (8 * (nb & 3) + r) : 0u;
        asm volatile("" : "+v"(b));
        mw[nb >> 2] |= b;
    };
    auto store_rows = [&](auto kk, auto gg, const u32x4& bhi) __attribute__((always_inline)) {
        if (!SAVE) return;
        constexpr int nb = 2 * decltype(kk)::value + (decltype(gg)::value >> 1);
        constexpr int r0 = 2 * (decltype(gg)::value & 1);
        store_word(row_region, W, nb, r0, bhi[decltype(gg)::value]);
        mask_bits(std::integral_constant<int, nb>{}, std::integral_constant<int, r0>{});
        mask_bits(std::integral_constant<int, nb>{}, std::integral_constant<int, r0 + 1>{});
    };
    auto finish_mask = [&](int layer) __attribute__((always_inline)) {     // the words save_mask16 builds, bit for bit
        if (!SAVE) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mw[i] <<= 4 * (q >> 1);
            mw[i] |= __shfl_xor(mw[i], 32);
        }
        if (valid && q < 2)
            nt_store(reinterpret_cast<u32x4*>(a.act + al.mask) + ((size_t)layer * P + p) * 2 + q, u32x4{mw[0], mw[1], mw[2], mw[3]});
#pragma unroll
        for (int i = 0; i < 4; ++i) mw[i] = 0u;
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
    // ---- density head: alpha_linear 256 -> 1 (VALU dot + quarter reduction)
    float sigma = 0.0f;
    {
        const float* wa = ring_small_ptr(lds, SM_WALPHA) + 4 * q;
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
        float v7[7];
        encode_dir(v7, vd0, vd1, vd2, q);
#pragma unroll
        for (int i = 0; i < 7; ++i) dv[i] = v7[i];
        dv[7] = 0.0f;
    }
    if (SAVE && valid && si == 0) {
        float* dout = a.act + al.dir + (size_t)ray * 32;
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            const int col = dirslot(s, q);
            if (col >= 0) nt_store(dout + col, dv[s]);
        }
    }
    f32x4 av[8];
    load_bias<8>(av, ring_small_ptr(lds, SM_BVIEWS), q);
    {   // layer 7's rows leave under the 16 units of the trunk part: k-step kk = blocks 2 kk, 2 kk + 1, half of them per unit
        row_region = (size_t)(D - 1) * layer_floats;
        auto store_rows_v = [&](auto kk, auto gg, const u32x4& bhi) __attribute__((always_inline)) {
            if (!SAVE) return;
            constexpr int nb = 2 * decltype(kk)::value + decltype(gg)::value;
            store_word(row_region, W, nb, 0, bhi[2 * decltype(gg)::value]);
            store_word(row_region, W, nb, 2, bhi[2 * decltype(gg)::value + 1]);
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
            for (int r = 0; r < 4; ++r) w[nb >> 2] |= (hv[4 * nb + r] > 0.0f ? 1u : 0u) << (8 * (nb & 3) + 4 * (q >> 1) + r);
#pragma unroll
        for (int i = 0; i < 2; ++i) w[i] |= __shfl_xor(w[i], 32);
        if (valid && q < 2)
            nt_store(reinterpret_cast<u32x4*>(a.act + al.mask) + ((size_t)D * P + p) * 2 + q, u32x4{w[0], w[1], w[2], w[3]});
    }
    // ---- rgb_linear 128 -> 3
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    {
        const float* wr = ring_small_ptr(lds, SM_WRGB) + 4 * q;
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
    if (valid && q == 0 && !(a.skip_last && si == a.S - 1)) *reinterpret_cast<f32x4*>(a.raw + (size_t)ri * 4) = f32x4{c0, c1, c2, sigma};
}

}  // namespace nerf
