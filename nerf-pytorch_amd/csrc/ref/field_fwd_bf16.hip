// Split-bf16 ("bf16x3") variant of the fused field evaluation (same boundary, same outputs as field_fwd.hip):
//   W*x ~= W_hi*x_hi + W_hi*x_lo + W_lo*x_hi   on v_mfma_f32_32x32x16_bf16, fp32 accumulate.
// ~1e-5 relative error per product (fp32: 6e-8, plain bf16: 4e-3) at 16/3 = 5.3x the fp32-MFMA rate.
// One wave = 32 points, activations stay in VGPRs as fp32 and are split into (hi, lo) bf16 fragments
// just in time for each k-step (the conversion VALU work hides under the 24 MFMAs of the previous k-step);
// weights stream L2 -> LDS as pre-split (hi, lo) fragments.  256-thread workgroups, 1 wave / SIMD (~350 VGPRs).
#include <type_traits>
#include "field_device_bf16.h"
#include "launchers.h"
#include "ref_launchers.h"

namespace nerf {

struct FieldFwd3Args {
    const float* packed3;   // PACKED3_WORDS
    const float* rays;
    const float* z_vals;
    float* raw;
    float* act;             // nullable: act_layout3 (32-point feature-major tiles)
    int ray_stride, n_rays, S;
};

// SAVE: 0 = inference, 1 = save fp32 tiles (bf16x3 backward), 2 = save bf16 tiles (mixed-precision backward; the
// arithmetic and raw are identical in all three)
template <int SAVE>
__global__ __launch_bounds__(FIELD3_WAVES * 64) void field_fwd3_kernel(FieldFwd3Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;
    const long P = (long)a.n_rays * a.S;
    const long p_raw = ((long)blockIdx.x * FIELD3_WAVES + wave) * PTS_PER_WAVE3 + (lane & 31);
    const bool valid = p_raw < P;
    const long p = valid ? p_raw : P - 1;
    const int ray = (int)(p / a.S);
    // this wave's tile of the saved regions and the lane's slot in it (feature 4*half, point lane&31)
    const size_t tile = (size_t)blockIdx.x * FIELD3_WAVES + wave;
    const int lslot = half * 128 + (lane & 31);

    WeightStreamT<2, FIELD3_WAVES> ws;
    ws.start(a.packed3, lds, wave, lane, SAVE && valid);
    stage_small_from(a.packed3 + P3_SMALL, lds, FIELD3_WAVES * 64);

    const float* rp = a.rays + (long)ray * a.ray_stride;
    const float z = a.z_vals[p];
    const float x0 = rp[0] + rp[3] * z;
    const float x1 = rp[1] + rp[4] * z;
    const float x2 = rp[2] + rp[5] * z;
    const float vd0 = rp[8], vd1 = rp[9], vd2 = rp[10];

    // ---- xyz encoding: 32 values per lane (slot map enc3slot)
    float e[32];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int i = half * 16 + m;                 // (freq, axis) pair
        if (m < 14 || half == 0) {
            const int dim = i % 3, fr = i / 3;
            const float xv = dim == 0 ? x0 : (dim == 1 ? x1 : x2);
            float sn, cs;
            sincosf(xv * pow2f(fr), &sn, &cs);
            e[2 * m] = sn;
            e[2 * m + 1] = cs;
        } else if (m == 14) { e[28] = x0; e[29] = x1; }
        else { e[30] = x2; e[31] = 0.0f; }
    }
    ActLayout3 al{};
    if (SAVE) {
        al = act_layout3((size_t)P, (size_t)a.n_rays);
        if (valid) {
            const size_t eoff = tile * (64 * 32) + (lane & 31);
#pragma unroll
            for (int v = 0; v < 32; ++v) {
                const int col = enc3slot(v, half);
                if (col >= 0) {
                    if (SAVE == 2) nt_store(reinterpret_cast<__bf16*>(a.act + al.enc) + eoff + col * 32, (__bf16)e[v]);
                    else nt_store(a.act + al.enc + eoff + col * 32, e[v]);
                }
            }
        }
    }

    const float* bias = small_ptr(lds, SM_BIAS);
    f32x16 acc[8];
    float h[128];
    auto take = [&](bool relu) {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) h[16 * nb + r] = relu ? fmaxf(acc[nb][r], 0.0f) : acc[nb][r];
    };
    // Saved rows of layer L are written while layer L+1 computes, a quarter (32 stores) after each of its 4 chunk
    // acquires; the following acquire waits with a counted vmcnt so these stores stay in flight (acquire<N>;
    // the hardware counter holds 63).
    constexpr int NQ = SAVE ? STORES_PER_QUARTER3 : 0;
    constexpr int NQ2 = SAVE ? 63 : 0;          // two quarters issued back to back
    const size_t layer_floats = pad32((size_t)P) * W;
    auto save_quarter = [&](auto part, size_t base_off, bool with_mask, int layer) {
        if (SAVE && valid) {
            if (SAVE == 2) store_tile3h<2 * decltype(part)::value, 2>(reinterpret_cast<__bf16*>(a.act + base_off) + tile * (W * 32) + lslot, h);
            else store_tile3<2 * decltype(part)::value, 2>(a.act + base_off + tile * (W * 32) + lslot, h);
            if (with_mask) save_mask3<128>(a.act + al.mask, layer, (size_t)P, (size_t)p, half, h);
        }
    };
    using Q0 = std::integral_constant<int, 0>; using Q1 = std::integral_constant<int, 1>;
    using Q2 = std::integral_constant<int, 2>; using Q3 = std::integral_constant<int, 3>;

    // ---- layer 0
    load_bias3<8>(acc, bias, half);
    mma3_chunk<8, 4, 0, 32>(acc, e, ws.acquire(), lane);
    take(true);

    // ---- layers 1..7
#pragma unroll 1
    for (int l = 1; l < D; ++l) {
        load_bias3<8>(acc, bias + l * W, half);
        const size_t prev_off = (size_t)(l - 1) * layer_floats;             // al.h[l-1]
        const float* cur = ws.acquire();
        if (l == SKIP + 1) { mma3_chunk<8, 4, 0, 32>(acc, e, cur, lane); cur = ws.acquire(); }
        save_quarter(Q0{}, prev_off, true, l - 1);                         // 32 row stores + 1 mask store
        mma3_chunk<8, 4, 0, 128>(acc, h, cur, lane);
        cur = ws.template acquire<SAVE ? NQ + 1 : 0>();
        save_quarter(Q1{}, prev_off, false, 0);
        mma3_chunk<8, 4, 32, 128>(acc, h, cur, lane);
        cur = ws.template acquire<NQ>();
        save_quarter(Q2{}, prev_off, false, 0);
        mma3_chunk<8, 4, 64, 128>(acc, h, cur, lane);
        // (after layer 7 the feature_linear chunks of the stream are skipped: folded into the view branch)
        cur = ws.template acquire<NQ>(l == D - 1 ? FOLD_SKIP_CHUNKS_FWD : 0);
        save_quarter(Q3{}, prev_off, false, 0);
        mma3_chunk<8, 4, 96, 128>(acc, h, cur, lane);
        take(true);
    }

    // ---- density head (VALU dot + half reduction)
    float sigma = 0.0f;
    {
        const float* wa = small_ptr(lds, SM_WALPHA);
#pragma unroll
        for (int ob = 0; ob < 8; ++ob)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(wa + 32 * ob + 8 * g + 4 * half);
#pragma unroll
                for (int r = 0; r < 4; ++r) sigma = fmaf(h[16 * ob + 4 * g + r], w[r], sigma);
            }
        sigma = half_sum(sigma) + small_ptr(lds, SM_BALPHA)[0];
    }

    // ---- view branch on the trunk output (feature_linear folded into it, nerf_common.h): [h7, enc(dir)] -> 128, ReLU
    float dv[16];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int i = half * 8 + m;
        if (half == 0 || m < 4) {
            const int dim = i % 3, fr = i / 3;
            const float xv = dim == 0 ? vd0 : (dim == 1 ? vd1 : vd2);
            float sn, cs;
            sincosf(xv * pow2f(fr), &sn, &cs);
            dv[2 * m] = sn;
            dv[2 * m + 1] = cs;
        } else if (m == 4) { dv[8] = vd0; dv[9] = vd1; }
        else if (m == 5) { dv[10] = vd2; dv[11] = 0.0f; }
        else { dv[2 * m] = 0.0f; dv[2 * m + 1] = 0.0f; }
    }
    if (SAVE && valid && (p - (long)ray * a.S) == 0) {
        float* dout = a.act + al.dir + (size_t)ray * 32;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int col = dir3slot(v, half);
            if (col >= 0) nt_store(dout + col, dv[v]);
        }
    }
    f32x16 av[4];
    load_bias3<4>(av, small_ptr(lds, SM_BVIEWS), half);
    {
        const size_t h7 = (size_t)(D - 1) * layer_floats;
        const float* cur = ws.template acquire<NQ>();                      // quarter 3 of layer 6 is still draining
        save_quarter(Q0{}, h7, true, D - 1);                               // layer 7's rows + its ReLU bitmask
        save_quarter(Q1{}, h7, false, 0);
        mma3_chunk<4, 8, 0, 128>(av, h, cur, lane);
        cur = ws.template acquire<NQ2>();
        save_quarter(Q2{}, h7, false, 0);
        save_quarter(Q3{}, h7, false, 0);
        mma3_chunk<4, 8, 64, 128>(av, h, cur, lane);
        mma3_chunk<4, 2, 0, 16>(av, dv, ws.template acquire<NQ2>(), lane);
    }
    float hv[64];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[16 * nb + r] = fmaxf(av[nb][r], 0.0f);
    if (SAVE && valid) {
        if (SAVE == 2) store_tile3h<0, 4>(reinterpret_cast<__bf16*>(a.act + al.hv) + tile * (WV * 32) + lslot, hv);
        else store_tile3<0, 4>(a.act + al.hv + tile * (WV * 32) + lslot, hv);
        save_mask3<64>(a.act + al.mask, D, (size_t)P, (size_t)p, half, hv);
    }

    // ---- rgb_linear 128 -> 3
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    {
        const float* wr = small_ptr(lds, SM_WRGB);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = 32 * ob + 8 * g + 4 * half;
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr + col);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(wr + WV + col);
                const f32x4 w2 = *reinterpret_cast<const f32x4*>(wr + 2 * WV + col);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x = hv[16 * ob + 4 * g + r];
                    c0 = fmaf(x, w0[r], c0);
                    c1 = fmaf(x, w1[r], c1);
                    c2 = fmaf(x, w2[r], c2);
                }
            }
        c0 = half_sum(c0) + small_ptr(lds, SM_BRGB)[0];
        c1 = half_sum(c1) + small_ptr(lds, SM_BRGB)[1];
        c2 = half_sum(c2) + small_ptr(lds, SM_BRGB)[2];
    }
    if (valid && half == 0) *reinterpret_cast<f32x4*>(a.raw + (size_t)p * 4) = f32x4{c0, c1, c2, sigma};
}

// ------------------------------------------------------------------ forward, 16 points per wave
// Same arithmetic class as field_fwd3_kernel (3 bf16 MFMAs per product, fp32 accumulate) on
// v_mfma_f32_16x16x32_bf16 with the fp32 kernel's shape: a wave owns 16 points, 64 accumulator + 64 activation
// registers per lane => 2 waves / SIMD (8 waves = 128 points share the 2 x 64 KiB weight buffers), so a second wave
// covers the LDS latency and conversion VALU work that the 32-point kernel leaves exposed at 1 wave / SIMD.  Price:
// every A fragment serves 16 instead of 32 points (2x the ds_read_b128 traffic per MAC).  The contraction-slot maps
// are the fp32 datapath's (hcol / encslot / dirslot with value index 8*s + j); the products are summed in a different
// order than in the 32-point kernel, so the two agree to rounding (~1e-5 of |raw|), not bit for bit.
// SAVE (0 none / 1 fp32 / 2 bf16) writes the act buffer of field_fwd3_kernel<SAVE> -- the same regions, the same
// ReLU bitmask words (lane (pt, q) owns nibbles 8*nb + 4*(q>>1) of the 128-bit half q&1; lanes q and q^2 are OR-ed), the
// same encoding tiles -- so the backward kernels are shared, with ONE difference for SAVE == 1: the rows of the 256- /
// 128-wide regions go to 16-point tiles (nerf_common.h, row16) so that every non-temporal store instruction writes
// full 128-byte lines; the weight-gradient GEMM is told which of the two tilings its B operands have.
template <int SAVE>
__global__ __launch_bounds__(FIELD_WAVES * 64) void field_fwd16_kernel(FieldFwd3Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4;
    const long P = (long)a.n_rays * a.S;
    const long p_raw = ((long)blockIdx.x * FIELD_WAVES + wave) * PTS_PER_WAVE + (lane & 15);
    const bool valid = p_raw < P;
    const long p = valid ? p_raw : P - 1;
    const int ray = (int)(p / a.S);

    WeightStreamT<2, FIELD_WAVES> ws;                  // same chunk sizes as the 32-point forward stream
    ws.start(a.packed3 + P16F, lds, wave, lane, SAVE && valid);
    stage_small_from(a.packed3 + P3_SMALL, lds, FIELD_WAVES * 64);

    const float* rp = a.rays + (long)ray * a.ray_stride;
    const float z = a.z_vals[p];
    const float x0 = rp[0] + rp[3] * z;
    const float x1 = rp[1] + rp[4] * z;
    const float x2 = rp[2] + rp[5] * z;
    const float vd0 = rp[8], vd1 = rp[9], vd2 = rp[10];
    float e[16];
    encode_xyz(e, x0, x1, x2, q);

    // ---- saving: tile = 32 points = two waves; this lane's point is column pp of the tile
    ActLayout3 al{};
    const size_t tile = (size_t)(p_raw >> 5);
    const int pp = (int)(p_raw & 31);
    const size_t layer_floats = pad32((size_t)P) * W;
    auto store_val = [&](size_t region, int F, int f, float v) {
        const size_t idx = tile * (size_t)(F * 32) + (size_t)f * 32 + pp;
        if (SAVE == 2) nt_store(reinterpret_cast<__bf16*>(a.act + region) + idx, (__bf16)v);
        else nt_store(a.act + region + idx, v);
    };
    // rows of the 256- / 128-wide regions: 16-point tiles whose row order turns the four 64-byte runs of one store
    // instruction into two full 128-byte lines (nerf_common.h, row16); bf16 saves (SAVE == 2): see store_pair below
    // The wave's 16 points are one tile: the tile base is wave-uniform (SGPR pair), the lane contributes a 32-bit offset
    // (its row inside a 16-feature block and its point), the (block, register) part is an immediate -- no 64-bit VALU
    // address arithmetic per store, no address registers held across the MFMA loop.
    const unsigned tile16 = (unsigned)__builtin_amdgcn_readfirstlane((int)(((long)blockIdx.x * FIELD_WAVES + wave)));
    const unsigned lane_row_off = (unsigned)((8 * (q >> 1) + (q & 1)) * 16 + (lane & 15));      // floats
    auto store_row = [&](size_t region, int F, int nb, int r, float v) __attribute__((always_inline)) {
        // feature f = 16*nb + 4*q + r: row16(f) * 16 = nb*256 + (8*(q>>1) + (q&1))*16 + r*32
        float* tile_base = a.act + region + (size_t)tile16 * (size_t)(F * 16);                 // uniform
        nt_store(tile_base + (nb * 256 + r * 32) + lane_row_off, v);
    };
    // SAVE == 2 (bf16 rows, operands of the bf16 weight-gradient GEMM): 16-point tiles as well, [tile][F rows][16 points]
    // of 2 bytes, rows in row16h order (nerf_common.h).  Two lanes with adjacent points pair up: both pack their rows
    // (r, r+1), swap the packed word with the neighbour (DPP quad_perm [1,0,3,2]) and pick — the even lane row r of both
    // points, the odd lane row r+1 — with one v_perm: 3 VALU ops and ONE dword store per two values.  The 16 lanes of
    // quarter q then hold rows 4q + r, 4q + r + 1 = 64 bytes; row16h puts the four quarters' pairs next to each other, so
    // one store instruction writes 256 contiguous bytes = two full 128-byte lines.  (Natural row order — 64-byte halves
    // of a line written by instructions ~400 clk apart — cost 1.39x the write requests: PMC WRITE_SIZE 6.0 KB/point for
    // 4.35 KB of rows; 2-byte stores into 32-point tiles left 32-byte runs and took the HBM time of the fp32 rows.)
    const unsigned odd = (unsigned)lane & 1u;
    const unsigned pair_sel = odd ? 0x03020706u : 0x05040100u;      // v_perm_b32 bytes of {neighbour word, own word}
    const unsigned lane_pair_off = (unsigned)((2 * q + (int)odd) * 8 + ((lane & 15) >> 1));     // dwords: row16h(4q + r0 + odd) - 4*r0, point pair
    // The stores are unconditional (a per-lane predicate is an exec-mask branch and a scheduling barrier per store):
    // padding points of a ragged last tile are written (the weight-gradient GEMM masks points >= P), and a wave whose whole
    // tile lies beyond the padded point range (last workgroup only) is pointed at a dump tile — the `feat` region, which
    // the folded datapaths never use (wave-uniform select of the base).
    const bool tile_ok = (size_t)tile16 * 16 < pad32((size_t)P);
    auto store_pair = [&](size_t region, int F, int nb, int r0, float v0, float v1) __attribute__((always_inline)) {
        const unsigned own = pack_bf16x2(v0, v1);                                       // rows r0 (lo), r0 + 1 (hi) of this point
        const unsigned nbr = (unsigned)__builtin_amdgcn_mov_dpp((int)own, 0xB1, 0xF, 0xF, true);
        const unsigned word = __builtin_amdgcn_perm(nbr, own, pair_sel);                // (point 2j, point 2j+1) of row r0 + odd
        unsigned* tile_base = reinterpret_cast<unsigned*>(a.act + (tile_ok ? region : al.feat))
                              + (tile_ok ? (size_t)tile16 * (size_t)(F * 8) : (size_t)0);                          // uniform
        nt_store(tile_base + (16 * nb + 4 * r0) * 8 + lane_pair_off, word);                        // r0 in {0, 2}: rows 0-7 / 8-15 of the block
    };
    if (SAVE) {
        al = act_layout3((size_t)P, (size_t)a.n_rays);
        if (valid) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int col = encslot(s, q);
                if (col >= 0) store_val(al.enc, 64, col, e[s]);
            }
        }
    }

    const float* bias = small_ptr(lds, SM_BIAS);
    f32x4 acc[16];
    float h[64];
    auto take = [&](bool relu) {
#pragma unroll
        for (int nb = 0; nb < 16; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) h[4 * nb + r] = relu ? fmaxf(acc[nb][r], 0.0f) : acc[nb][r];
    };
    // ---- saving.  HBM absorbs a workgroup's rows (128 points x 256 x 4 B = 128 KiB per layer) at about the rate the
    // layer computes (~10 B/clk/CU fair share of 6 TB/s vs ~13k clk of MFMA issue per layer), so the rows must leave as a
    // STEADY stream: 64 store instructions issued back to back after a layer stall the wave on the full store queue
    // (measured: saving cost 0.70 ms of a 2.9 ms launch, all of it issue stalls, SQ_WAIT_INST_ANY).  The rows of layer L
    // (they stay live in h[] as the B operand of layer L+1) are therefore written DURING layer L+1, one group of 8
    // stores per k-step, i.e. 16 per weight chunk; every acquire<> of that layer waits with a counted vmcnt so that the
    // 16 newest stores keep draining under the next chunk's MFMAs (WeightStreamT::acquire).
    constexpr int ST_K = SAVE == 2 ? 4 : (SAVE ? 8 : 0);  // row stores per k-step (bf16 rows: one paired store per two rows)
    constexpr int ST_C = 2 * ST_K;                       // per 2-k-step chunk
    // value 4*nb + r of h = feature 16*nb + 4*q + r; k-step k writes blocks 2k, 2k+1: two stores after each of its four
    // MFMA groups (group g: block 2k + (g >> 1), registers 2*(g & 1), 2*(g & 1) + 1)
    auto save_pair = [&](auto kk, auto gg, size_t region) __attribute__((always_inline)) {
        if (!SAVE || (SAVE == 1 && !valid)) return;      // (paired stores: every lane takes part in the exchange)
        constexpr int nb = 2 * decltype(kk)::value + (decltype(gg)::value >> 1);
        constexpr int r0 = 2 * (decltype(gg)::value & 1);
        if (SAVE == 2) { store_pair(region, W, nb, r0, h[4 * nb + r0], h[4 * nb + r0 + 1]); return; }
        store_row(region, W, nb, r0, h[4 * nb + r0]);
        store_row(region, W, nb, r0 + 1, h[4 * nb + r0 + 1]);
    };
    auto save_part = [&](auto part, size_t region) __attribute__((always_inline)) {      // the 8 stores of one k-step at once
        save_pair(part, std::integral_constant<int, 0>{}, region);
        save_pair(part, std::integral_constant<int, 1>{}, region);
        save_pair(part, std::integral_constant<int, 2>{}, region);
        save_pair(part, std::integral_constant<int, 3>{}, region);
    };
    auto save_mask16 = [&](int layer) __attribute__((always_inline)) {                   // ReLU bitmask of the rows in h[] (one 16-byte store, lanes q < 2)
        if (!SAVE) return;
        unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int nb = 0; nb < 16; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) w[nb >> 2] |= (h[4 * nb + r] > 0.0f ? 1u : 0u) << (8 * (nb & 3) + 4 * (q >> 1) + r);
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] |= __shfl_xor(w[i], 32);            // the other nibbles: lane q ^ 2
        if (valid && q < 2)
            nt_store(reinterpret_cast<u32x4*>(a.act + al.mask) + ((size_t)layer * P + p) * 2 + q, u32x4{w[0], w[1], w[2], w[3]});
    };
    using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;
    using K4 = std::integral_constant<int, 4>; using K5 = std::integral_constant<int, 5>;
    using K6 = std::integral_constant<int, 6>; using K7 = std::integral_constant<int, 7>;
    // one 256-wide contraction (4 chunks x 2 k-steps) of h into acc while the rows in h go to `region`.  `first` is the
    // already acquired first chunk.
    auto kstep_h = [&](auto part, const float* kstep_base, size_t region, bool store_rows) __attribute__((always_inline)) {
        mma16_kstep_with<8 * decltype(part)::value, 64>(acc, h, kstep_base, lane,
                                                         [&](auto gg) __attribute__((always_inline)) { if (store_rows) save_pair(part, gg, region); });
    };
    // skip_after: chunks of the stream to jump over behind this contraction's last chunk (the folded feature layer)
    auto contract_h = [&](const float* first, size_t region, bool store_rows, int skip_after) __attribute__((always_inline)) {
        const float* cur = first;
        kstep_h(K0{}, cur, region, store_rows);
        kstep_h(K1{}, cur + KSTEP16_W16, region, store_rows);
        cur = store_rows ? ws.template acquire<ST_C>() : ws.acquire();
        kstep_h(K2{}, cur, region, store_rows);
        kstep_h(K3{}, cur + KSTEP16_W16, region, store_rows);
        cur = store_rows ? ws.template acquire<ST_C>() : ws.acquire();
        kstep_h(K4{}, cur, region, store_rows);
        kstep_h(K5{}, cur + KSTEP16_W16, region, store_rows);
        cur = store_rows ? ws.template acquire<ST_C>(skip_after) : ws.acquire(skip_after);      // prefetches the chunk after this layer
        kstep_h(K6{}, cur, region, store_rows);
        kstep_h(K7{}, cur + KSTEP16_W16, region, store_rows);
    };

    // ---- layer 0: 63 -> 256 (2 k-steps = one chunk)
    load_bias<16>(acc, bias, q);
    mma16_chunk<16, 2, 0, 16>(acc, e, ws.acquire(), lane);
    take(true);
    save_mask16(0);
    // ---- layers 1..7 (layer 5 also contracts the xyz encoding: skip connection); 8 k-steps = 4 chunks.  Layer l writes
    // the rows of layer l-1.  Pending stores at the first acquire of a layer: the last chunk's 16 (+ the mask store)
#pragma unroll 1
    for (int l = 1; l < D; ++l) {
        load_bias<16>(acc, bias + l * W, q);
        const float* cur = (l == 1) ? ws.template acquire<SAVE ? 1 : 0>() : ws.template acquire<SAVE ? ST_C + 1 : 0>();
        if (l == SKIP + 1) { mma16_chunk<16, 2, 0, 16>(acc, e, cur, lane); cur = ws.acquire(); }
        // after layer 7 the stream continues with the view branch: the feature_linear chunks are skipped (folded W')
        contract_h(cur, (size_t)(l - 1) * layer_floats, SAVE != 0, l == D - 1 ? FOLD_SKIP_CHUNKS_FWD : 0);
        take(true);
        save_mask16(l);
    }
    // ---- density head: alpha_linear 256 -> 1 (VALU dot + quarter reduction)
    float sigma = 0.0f;
    {
        const float* wa = small_ptr(lds, SM_WALPHA) + 4 * q;
#pragma unroll
        for (int nb = 0; nb < 16; ++nb) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(wa + 16 * nb);
#pragma unroll
            for (int r = 0; r < 4; ++r) sigma = fmaf(h[4 * nb + r], w[r], sigma);
        }
        sigma = quarter_sum(sigma) + small_ptr(lds, SM_BALPHA)[0];
    }
    // ---- view branch on the trunk output (feature_linear folded into it: W' = Wv[:, :256] Wf, b' in SM_BVIEWS;
    // nerf_common.h): [h7, enc(dir)] -> 128, ReLU: 4 + 4 k-steps of 128 outputs, then the dir k-step
    float dv[8];
    {
        float v7[7];
        encode_dir(v7, vd0, vd1, vd2, q);
#pragma unroll
        for (int i = 0; i < 7; ++i) dv[i] = v7[i];
        dv[7] = 0.0f;
    }
    if (SAVE && valid && (p - (long)ray * a.S) == 0) {
        float* dout = a.act + al.dir + (size_t)ray * 32;
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            const int col = dirslot(s, q);
            if (col >= 0) nt_store(dout + col, dv[s]);
        }
    }
    f32x4 av[8];
    load_bias<8>(av, small_ptr(lds, SM_BVIEWS), q);
    {   // layer 7's rows (h[]) are written under the two trunk chunks of the view branch (4 k-steps of 128 outputs each)
        const size_t h7 = (size_t)(D - 1) * layer_floats;
        const float* cur = ws.template acquire<SAVE ? ST_C + 1 : 0>();      // pending: layer 6's last 16 rows + layer 7's mask
        save_part(K0{}, h7); save_part(K1{}, h7); save_part(K2{}, h7); save_part(K3{}, h7);
        mma16_chunk<8, 4, 0, 64>(av, h, cur, lane);
        cur = ws.template acquire<4 * ST_K>();
        save_part(K4{}, h7); save_part(K5{}, h7); save_part(K6{}, h7); save_part(K7{}, h7);
        mma16_chunk<8, 4, 32, 64>(av, h, cur, lane);
        mma16_chunk<8, 1, 0, 8>(av, dv, ws.template acquire<4 * ST_K>(), lane);
    }
    float hv[32];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[4 * nb + r] = fmaxf(av[nb][r], 0.0f);
    if (SAVE) {
        if (SAVE == 2) {
#pragma unroll
            for (int nb = 0; nb < 8; ++nb)
#pragma unroll
                for (int r = 0; r < 4; r += 2) store_pair(al.hv, WV, nb, r, hv[4 * nb + r], hv[4 * nb + r + 1]);
        } else if (valid) {
#pragma unroll
            for (int nb = 0; nb < 8; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) store_row(al.hv, WV, nb, r, hv[4 * nb + r]);
        }
        // view-branch mask: 128 features = 64 bits per half; value 4*nb + r -> bit 8*nb + 4*(q>>1) + r of half q&1
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
        const float* wr = small_ptr(lds, SM_WRGB) + 4 * q;
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
        c0 = quarter_sum(c0) + small_ptr(lds, SM_BRGB)[0];
        c1 = quarter_sum(c1) + small_ptr(lds, SM_BRGB)[1];
        c2 = quarter_sum(c2) + small_ptr(lds, SM_BRGB)[2];
    }
    if (valid && q == 0) *reinterpret_cast<f32x4*>(a.raw + (size_t)p * 4) = f32x4{c0, c1, c2, sigma};
}

hipError_t launch_field_fwd16(const float* packed3, const float* rays, int ray_stride, const float* z_vals,
                              int n_rays, int S, float* raw, float* act, int bf16_save, hipStream_t stream) {
    FieldFwd3Args a{packed3, rays, z_vals, raw, act, ray_stride, n_rays, S};
    const long P = (long)n_rays * S;
    if (P <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((P + PTS_PER_WG - 1) / PTS_PER_WG);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e0 = hipFuncSetAttribute((const void*)field_fwd16_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, FIELD_LDS_FLOATS * 4);
        hipError_t e1 = hipFuncSetAttribute((const void*)field_fwd16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, FIELD_LDS_FLOATS * 4);
        hipError_t e2 = hipFuncSetAttribute((const void*)field_fwd16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, FIELD_LDS_FLOATS * 4);
        if (e0 != hipSuccess) return e0;
        if (e1 != hipSuccess) return e1;
        if (e2 != hipSuccess) return e2;
        attr_set = true;
    }
    if (act && bf16_save)
        hipLaunchKernelGGL(field_fwd16_kernel<2>, dim3(blocks), dim3(FIELD_WAVES * 64), FIELD_LDS_FLOATS * 4, stream, a);
    else if (act)
        hipLaunchKernelGGL(field_fwd16_kernel<1>, dim3(blocks), dim3(FIELD_WAVES * 64), FIELD_LDS_FLOATS * 4, stream, a);
    else
        hipLaunchKernelGGL(field_fwd16_kernel<0>, dim3(blocks), dim3(FIELD_WAVES * 64), FIELD_LDS_FLOATS * 4, stream, a);
    return hipGetLastError();
}

hipError_t launch_field_fwd3(const float* packed3, const float* rays, int ray_stride, const float* z_vals,
                             int n_rays, int S, float* raw, float* act, int bf16_save, hipStream_t stream) {
    FieldFwd3Args a{packed3, rays, z_vals, raw, act, ray_stride, n_rays, S};
    const long P = (long)n_rays * S;
    if (P <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((P + PTS_PER_WG3 - 1) / PTS_PER_WG3);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e0 = hipFuncSetAttribute((const void*)field_fwd3_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, FIELD_LDS_FLOATS * 4);
        hipError_t e1 = hipFuncSetAttribute((const void*)field_fwd3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, FIELD_LDS_FLOATS * 4);
        hipError_t e2 = hipFuncSetAttribute((const void*)field_fwd3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, FIELD_LDS_FLOATS * 4);
        if (e0 != hipSuccess) return e0;
        if (e1 != hipSuccess) return e1;
        if (e2 != hipSuccess) return e2;
        attr_set = true;
    }
    if (act && bf16_save)
        hipLaunchKernelGGL(field_fwd3_kernel<2>, dim3(blocks), dim3(FIELD3_WAVES * 64), FIELD_LDS_FLOATS * 4, stream, a);
    else if (act)
        hipLaunchKernelGGL(field_fwd3_kernel<1>, dim3(blocks), dim3(FIELD3_WAVES * 64), FIELD_LDS_FLOATS * 4, stream, a);
    else
        hipLaunchKernelGGL(field_fwd3_kernel<0>, dim3(blocks), dim3(FIELD3_WAVES * 64), FIELD_LDS_FLOATS * 4, stream, a);
    return hipGetLastError();
}

}  // namespace nerf
