// Shared layout definitions for the gfx950 NeRF field kernels (host + device).
//
// Field model = reference NeRF(D=8, W=256, input_ch=63, input_ch_views=27,
// skips=[4], use_viewdirs=True)  (run_nerf_helpers.py:67-119).
//
// Register-resident "transposed" formulation: every wave owns 16 sample points
// and computes  H_out^T[256 x 16] = W[256 x K] * H_in^T[K x 16]  with
// v_mfma_f32_16x16x4_f32, so the MFMA D layout of one layer (lane (p = l&15,
// q = l>>4) holds features 16*nb + 4*q + r, r = 0..3 of point p) is directly
// the B operand of the next layer: contraction index k is free to be permuted
// as long as the A operand (weights) is packed with the same permutation.
// Activations therefore never leave VGPRs between layers; only weights stream
// L2 -> LDS -> A fragments.
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifndef __HIPCC__
#define __host__
#define __device__
#endif

namespace nerf {

constexpr int W = 256;          // hidden width
constexpr int D = 8;            // depth of the xyz trunk
constexpr int SKIP = 4;         // xyz encoding re-enters after layer SKIP
constexpr int L_XYZ = 10;       // multires
constexpr int L_DIR = 4;        // multires_views
constexpr int IN_XYZ = 63;      // 3 + 6*L_XYZ
constexpr int IN_DIR = 27;      // 3 + 6*L_DIR
constexpr int WV = 128;         // W/2, view-branch width

// ---- canonical (state_dict order, PyTorch [out,in] row-major) parameter offsets
struct Canon {
    int w[D], b[D];             // pts_linears.i
    int wv, bv;                 // views_linears.0  [128,283]
    int wf, bf;                 // feature_linear   [256,256]
    int wa, ba;                 // alpha_linear     [1,256]
    int wr, br;                 // rgb_linear       [3,128]
    int total;
};
__host__ __device__ constexpr int fan_in(int layer) {
    return layer == 0 ? IN_XYZ : (layer == SKIP + 1 ? W + IN_XYZ : W);
}
__host__ __device__ constexpr Canon canon() {
    Canon c{};
    int o = 0;
    for (int i = 0; i < D; ++i) { c.w[i] = o; o += W * fan_in(i); c.b[i] = o; o += W; }
    c.wv = o; o += WV * (W + IN_DIR); c.bv = o; o += WV;
    c.wf = o; o += W * W;             c.bf = o; o += W;
    c.wa = o; o += W;                 c.ba = o; o += 1;
    c.wr = o; o += 3 * WV;            c.br = o; o += 3;
    c.total = o;
    return c;
}
constexpr int N_PARAMS = 595844;
static_assert(canon().total == N_PARAMS, "parameter count");

// ---- contraction-slot maps (k-step s, lane quarter q) -> source column
// hidden features: the D-layout of the producing layer
__host__ __device__ constexpr int hcol(int s, int q) { return 16 * (s >> 2) + 4 * q + (s & 3); }
// xyz encoding, 16 k-steps: quarter q owns (freq,dim) pairs i = q + 4m, register
// 2m = sin, 2m+1 = cos (one sincosf per pair); the identity columns fill m = 7 of q = 2,3.
__host__ __device__ constexpr int encslot(int s, int q) {
    const int m = s >> 1, fn = s & 1, i = q + 4 * m;
    if (i < 3 * L_XYZ) return 3 + (i / 3) * 6 + fn * 3 + (i % 3);
    if (q == 2) return fn;                  // x, y
    return fn == 0 ? 2 : -1;                // z, pad
}
// view-direction encoding, 7 k-steps
__host__ __device__ constexpr int dirslot(int s, int q) {
    if (s < 6) { const int m = s >> 1, fn = s & 1, i = q + 4 * m; return 3 + (i / 3) * 6 + fn * 3 + (i % 3); }
    return q < 3 ? q : -1;
}
constexpr int KS_ENC = 16;      // k-steps of the xyz encoding (64 slots, 63 used)
constexpr int KS_DIR = 7;       // k-steps of the dir encoding (28 slots, 27 used)
constexpr int KS_H = 64;        // k-steps of a 256-wide hidden vector
constexpr int KS_HV = 32;       // k-steps of the 128-wide view-branch vector

// ---- packed weight stream.  One k-step of an NB-block layer is NB*64 floats:
//   [(s*(NB/4) + g)*64 + lane]*4 + j  =  A-fragment of out-block 4g+j:
//   W[16*(4g+j) + (lane&15)][slotcol(s, lane>>4)]
// Forward stream order: L0 | L1..L4 | L5(enc,h) | L6 L7 | FEAT | VIEWS(feat,dir)
constexpr int CHUNK_KS = 16;                        // k-steps staged per LDS buffer
constexpr int KSTEP_F16 = 1024;                     // floats per k-step, NB = 16
constexpr int KSTEP_F8 = 512;                       // floats per k-step, NB = 8
constexpr int FWD_L0 = 0;
constexpr int FWD_L1 = FWD_L0 + KS_ENC * KSTEP_F16;                 // layers 1..4
constexpr int FWD_L5 = FWD_L1 + 4 * KS_H * KSTEP_F16;
constexpr int FWD_L6 = FWD_L5 + (KS_ENC + KS_H) * KSTEP_F16;        // layers 6,7
constexpr int FWD_FEAT = FWD_L6 + 2 * KS_H * KSTEP_F16;
constexpr int FWD_VIEWS = FWD_FEAT + KS_H * KSTEP_F16;
constexpr int FWD_END = FWD_VIEWS + (KS_H + KS_DIR) * KSTEP_F8;
static_assert(FWD_END == 593408, "forward weight stream size");
// Backward (dgrad) stream order: VIEWS^T | FEAT^T | L7^T .. L1^T   (all NB = 16)
constexpr int BWD_VIEWS = FWD_END;
constexpr int BWD_FEAT = BWD_VIEWS + KS_HV * KSTEP_F16;
constexpr int BWD_L7 = BWD_FEAT + KS_H * KSTEP_F16;                 // then L6 .. L1
constexpr int BWD_END = BWD_L7 + 7 * KS_H * KSTEP_F16;
// small parameters, copied verbatim (16-byte aligned)
constexpr int SM_BIAS = BWD_END;                    // 8 x 256 trunk biases
constexpr int SM_BFEAT = SM_BIAS + D * W;
constexpr int SM_BVIEWS = SM_BFEAT + W;
constexpr int SM_WALPHA = SM_BVIEWS + WV;           // [256]
constexpr int SM_WRGB = SM_WALPHA + W;              // [3][128]
constexpr int SM_BALPHA = SM_WRGB + 3 * WV;         // [1] + pad
constexpr int SM_BRGB = SM_BALPHA + 4;              // [3] + pad
constexpr int PACKED_FLOATS = SM_BRGB + 4;
static_assert(PACKED_FLOATS % 4 == 0 && SM_BIAS % 4 == 0, "alignment");

// ---- saved activations / deltas for one field evaluation over P points, N rays
// (row-major, so the weight-gradient GEMM streams them coalesced)
struct ActLayout {
    size_t h[D];        // [P][256] post-ReLU trunk activations
    size_t feat;        // [P][256]
    size_t hv;          // [P][128] post-ReLU view-branch activations
    size_t enc;         // [P][64]  xyz encoding, canonical column order (col 63 = 0)
    size_t dir;         // [N][32]  dir encoding per ray (written by the forward), canonical column order
    size_t dir_pt;      // [P][32]  the same, expanded per point right before the weight-gradient GEMM so that
                        //          every operand is point-indexed (cols 27..31 unused)
    size_t mask;        // [9][P][4] uint64 ReLU sign bits in D-layout (bit 4*nb+r of lane quarter q); 9th = view branch
    size_t total;       // floats
};
__host__ __device__ inline ActLayout act_layout(size_t P, size_t N) {
    ActLayout a{};
    size_t o = 0;
    for (int i = 0; i < D; ++i) { a.h[i] = o; o += P * W; }
    a.feat = o; o += P * W;
    a.hv = o;   o += P * WV;
    a.enc = o;  o += P * 64;
    a.dir = o;  o += N * 32;
    a.dir_pt = o; o += P * 32;
    o = (o + 3) & ~(size_t)3;
    a.mask = o; o += (size_t)(D + 1) * P * 8;
    a.total = o;
    return a;
}
// deltas: dL/d(pre-activation) per layer, same shapes as h/feat/hv
struct DeltaLayout { size_t h[D]; size_t feat; size_t hv; size_t total; };
__host__ __device__ inline DeltaLayout delta_layout(size_t P) {
    DeltaLayout a{};
    size_t o = 0;
    for (int i = 0; i < D; ++i) { a.h[i] = o; o += P * W; }
    a.feat = o; o += P * W;
    a.hv = o;   o += P * WV;
    a.total = o;
    return a;
}

constexpr int PTS_PER_WAVE = 16;
constexpr int FIELD_WAVES = 8;                      // 512-thread workgroups, 2 waves / SIMD
constexpr int PTS_PER_WG = PTS_PER_WAVE * FIELD_WAVES;

}  // namespace nerf

// =====================================================================================
// Split-bf16 ("bf16x3") datapath: W*x ~= W_hi*x_hi + W_hi*x_lo + W_lo*x_hi on v_mfma_f32_32x32x16_bf16
// with fp32 accumulation (hi = bf16(v), lo = bf16(v - hi): 16-17 significant bits per operand, ~1e-5
// relative error per product instead of bf16's 4e-3).  One wave owns 32 points; the D layout of a
// 32-feature block (lane (p = l&31, half = l>>5) holds features 32*ob + (r&3) + 8*(r>>2) + 4*half,
// r = 0..15) is again the B operand of the next layer after a permutation of the contraction index.
// =====================================================================================
namespace nerf {

// feature held in accumulator register r of lane half `half` inside a 32-feature block
__host__ __device__ constexpr int d32row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
// hidden vector, 16 k-steps of 16 slots: k-step s, lane half, element j (0..7) -> feature
__host__ __device__ constexpr int h3slot(int s, int half, int j) { return 32 * (s >> 1) + d32row(8 * (s & 1) + j, half); }
constexpr int KS3_H = 16, KS3_HV = 8;
// The packed buffer of the split datapaths ("packed3", 32-bit words):
//   [ transposed (hi, lo) fragment streams of the delta chain | fp32 small parameters | 16-point forward stream | derived W', b' ]
// (rounds 1-4 also kept a 32-point forward stream and a hi-only copy of the transposed streams for kernels that no longer exist.)
// one k-step of an 8-block layer = 8 blocks x 2 (hi, lo) x 64 lanes x 16 B; sizes below in 32-bit words
constexpr int KSTEP3_W8 = 8 * 2 * 64 * 4;      // 4096 words = 16 KiB (256 outputs)
// element (nb, hl, lane, j) of a k-step lives at 16-bit index (((nb*2 + hl)*64 + lane)*8 + j)
constexpr int P3B_VIEWS = 0;                                            // transposed streams of the delta chain, all 8 blocks:
constexpr int P3B_FEAT = P3B_VIEWS + KS3_HV * KSTEP3_W8;                // W'^T | feature_linear^T (skipped by the kernel) |
constexpr int P3B_L7 = P3B_FEAT + KS3_H * KSTEP3_W8;                    // L7^T .. L1^T
constexpr int P3B_END = P3B_L7 + 7 * KS3_H * KSTEP3_W8;
constexpr int P3_SMALL = P3B_END;                                       // fp32 small parameters, same order as SM_*
// forward (hi, lo) fragments of the 16-point-per-wave kernels (v_mfma_f32_16x16x32_{f16,bf16}: lane = (row l&15, k-group l>>4), 8
// consecutive contraction slots per lane; the contraction slots are the fp32 datapath's hcol / encslot / dirslot maps with value
// index 8*s + j).  k-step = NB blocks x (hi, lo) x 64 lanes x 16 B: L0 | L1..L4 | L5(enc,h) | L6 L7 | FEAT (skipped) | VIEWS(h7,dir)
constexpr int P16F = P3_SMALL + (PACKED_FLOATS - SM_BIAS);
constexpr int KSTEP16_W16 = 16 * 2 * 64 * 4;                             // 8192 words = 32 KiB (256 outputs)
constexpr int KSTEP16_W8 = 8 * 2 * 64 * 4;                               // 4096 words (128 outputs)
constexpr int KS16_ENC = 2, KS16_H = 8, KS16_DIR = 1;                    // k-steps of 32 contraction slots
constexpr int P16F_L1 = KS16_ENC * KSTEP16_W16;                          // offsets inside the region
constexpr int P16F_L5 = P16F_L1 + 4 * KS16_H * KSTEP16_W16;
constexpr int P16F_L6 = P16F_L5 + (KS16_ENC + KS16_H) * KSTEP16_W16;
constexpr int P16F_FEAT = P16F_L6 + 2 * KS16_H * KSTEP16_W16;
constexpr int P16F_VIEWS = P16F_FEAT + KS16_H * KSTEP16_W16;
constexpr int P16F_WORDS = P16F_VIEWS + (KS16_H + KS16_DIR) * KSTEP16_W8;
static_assert(P16F_WORDS == 593920 && P16F % 4 == 0, "16-point forward stream");
// ---- FOLDED FEATURE LAYER (split datapaths).  feature_linear has no activation and feeds only
// views_linears.0 (run_nerf_helpers.py:111-115), so the two consecutive linear maps compose:
//     views_pre = Wv[:, :256] (Wf h7 + bf) + Wv[:, 256:] enc(dir) + bv = W' h7 + Wv[:, 256:] enc(dir) + b'
//     W' = Wv[:, :256] Wf  [128][256],   b' = Wv[:, :256] bf + bv  [128]
// The split kernels evaluate the view branch directly on the trunk output with W' (one 256x256 layer less in
// the forward, in dgrad and in the weight-gradient GEMM: -11 % MFMA work, and `feature` / its delta are neither saved
// nor re-read: -4 KB of the 43 KB of HBM traffic per point and step).  The parameters stay Wf, bf, Wv, bv; their
// gradients follow from ONE contraction G = delta_hv^T h7 [128][256] and dbv = sum delta_hv:
//     dWv[:, :256] = G Wf^T + dbv bf^T,   dWf = Wv[:, :256]^T G,   dbf = Wv[:, :256]^T dbv
// (wgrad_fold_kernel, exact fp32 FMA loops over K = 256 / 128).  W' and b' are derived at pack time (fp64 accumulation,
// rounded once to fp32: 6e-8 relative, two orders below the datapath's product error) and appended to the packed
// buffer; in the pack tables they appear as "canonical" indices N_PARAMS + k*256 + j and N_PARAMS + 128*256 + k.
// The exact-fp32 datapath keeps the reference's two-layer formulation (it is the bitwise anchor).
constexpr int DERIVED_WVF = N_PARAMS;
constexpr int DERIVED_BV = DERIVED_WVF + WV * W;
constexpr int N_DERIVED = WV * W + WV;                                   // 32,896 floats
constexpr int P3_DERIVED = P16F + P16F_WORDS;                            // offset of (W', b') inside the packed3 buffer
constexpr int PACKED3_WORDS = P3_DERIVED + N_DERIVED;
static_assert(P3_DERIVED % 4 == 0, "alignment of the derived parameters");

constexpr int PTS_PER_WAVE3 = 32;
constexpr int FIELD3_WAVES = 4;                                         // 256-thread workgroups, 1 wave / SIMD
constexpr int PTS_PER_WG3 = PTS_PER_WAVE3 * FIELD3_WAVES;

// ---- saved deltas and encodings of the split datapaths: 32-POINT TILES, FEATURE-MAJOR INSIDE A TILE (16-bit elements).
// Element (point p, feature f) of an F-feature region lives at  (p >> 5) * F * 32 + f * 32 + (p & 31):
//   * a wave of the 32-point delta chain owns exactly one tile, and one store instruction writes one feature of its
//     32 points per lane half = two full 128-byte lines (the point-major rows of the fp32 datapath would scatter
//     64 x 16 B per instruction);
//   * the weight-gradient GEMM contracts over points: a tile is its k-extent of 32, read as one contiguous block
//     (16 B per lane, lane-linear) and every feature's 32 points land next to each other -- the MFMA fragment order.
// Regions are sized for P rounded up to a tile; the pad points of the last tile are never written and never used.
__host__ __device__ constexpr size_t pad32(size_t P) { return (P + 31) & ~(size_t)31; }
__host__ __device__ inline size_t tile_index(size_t p, int F, int f) { return (p >> 5) * (size_t)(F * 32) + (size_t)f * 32 + (p & 31); }
// ---- 16-POINT TILES of the rows saved by the 16-point forward (regions h[0..7], hv).
// A wave of that kernel owns 16 points; lane (pt, q) holds features 16*nb + 4*q + j.  Non-temporal stores of HALF lines reach
// only 3.2-3.6 TB/s on MI355X against 5.6-6.0 TB/s for full lines (tools/probe/wr_probe.hip), so the rows are laid out so that
// each store instruction's runs pair up into full 128-byte lines.  Two consecutive 16-point tiles cover the same 32 points (and
// the same bytes) as one 32-point tile, so the weight-gradient GEMM stages them as one contiguous block; it contracts over points
// and only has to know which feature a staged row is.
// 16-bit rows (operands of the streaming weight-gradient GEMM): 16-point tiles of 2-byte
// elements, 32 bytes per row; lane pairs store (row 4q + r, row 4q + r + 1) x 2 points per dword for r in {0, 2}, so the
// feature 4*q + r sits at row 8*(r>>1) + 2*q + (r&1): the eight rows one instruction writes are contiguous (256 B).
//     element (p, f): 2-byte index (p >> 4) * F * 16 + row16h(f) * 16 + (p & 15)
__host__ __device__ constexpr int row16h(int f) { return (f & ~15) + 8 * ((f >> 1) & 1) + 2 * ((f >> 2) & 3) + (f & 1); }

// Offsets are in 32-bit words (the buffers are float* at the C ABI); a region of F 16-bit features holds Pp * F / 2 words -- rounds
// 1-4 sized these regions as if their elements were fp32 (twice the bytes), round 5 sizes them for what they hold: 4.8 KB / point
// saved, 4.4 KB / point of deltas, 40 GB instead of 90 GB for the 32,768-ray batch of BASELINE configs[3].
constexpr size_t DUMP_WORDS3 = 8192;     // >= one 32-point tile of 256 16-bit features (4096 words)
__host__ __device__ constexpr size_t region_words3(size_t Pp, int F) { return Pp * (size_t)(F / 2); }
struct ActLayout3 {
    size_t h[D];        // tiles of 256 features (16-bit elements)
    size_t feat;        // DUMP_WORDS3 words nobody reads: waves whose tile lies beyond the padded point range store here
    //                    (unconditional stores, split_types.h) -- the feature layer itself is folded, it saves nothing
    size_t hv;          // tiles of 128
    size_t enc;         // tiles of 64 (canonical column order, feature 63 unused)
    size_t dir;         // [N][32] fp32 per ray, row-major (written by the forward)
    size_t dir_pt;      // 16-bit tiles of 32: the same per point (or [N][32][8 copies] when a ray's samples fill whole tiles), expanded
    //                    right before the weight-gradient GEMM
    size_t mask;        // [9][P][2] x 128 ReLU sign bits in the lane order of the split kernels (9th = view branch)
    size_t lo;          // two-word saves (round 6, "fp16x3w"): word offset of a MIRROR of this whole layout that holds the LO words
    //                    (lo = T(v - hi)) of h[0..7], hv, enc, dir_pt at the same offsets; 0 = the one-word layout
    size_t total;
};
__host__ __device__ inline ActLayout3 act_layout3(size_t P, size_t N, bool two_word = false) {
    ActLayout3 a{};
    const size_t Pp = pad32(P);
    size_t o = 0;
    for (int i = 0; i < D; ++i) { a.h[i] = o; o += region_words3(Pp, W); }
    a.feat = o; o += DUMP_WORDS3;
    a.hv = o;   o += region_words3(Pp, WV);
    a.enc = o;  o += region_words3(Pp, 64);
    a.dir = o;  o += N * 32;
    const size_t per_point = region_words3(Pp, 32), per_ray = N * 32 * 4;     // [N][32] x 8 copies x 2 bytes
    a.dir_pt = o; o += per_point > per_ray ? per_point : per_ray;
    o = (o + 3) & ~(size_t)3;
    a.mask = o; o += (size_t)(D + 1) * P * 8;
    o += 2048;          // slack for the weight-gradient staging's reads past a narrow operand's last tile
    a.lo = two_word ? o : 0;        // (o is a multiple of 4 words: every mirrored region keeps its 16-byte alignment)
    a.total = two_word ? 2 * o : o;
    return a;
}
// graw: tiles of 4 = copy of d_raw (rgb3, sigma).  scale: 4 words, word 0 = the bit pattern of max|d_raw| over the launch
// (delta_amax_kernel, field_bwd_ring.hip) -- the fp16 split's delta chain runs on s * d_raw, s = delta_scale_of(max) an exact
// power of two, so that the deltas sit in fp16's range; every stored delta and every partial weight gradient carries the factor
// s, wgrad_reduce_kernel removes it (both kernels derive s / 1/s from the same word).
// lo: as ActLayout3::lo -- the mirror holds the lo words of h[0..7], hv, graw; the scale word exists once (hi part)
struct DeltaLayout3 { size_t h[D], feat, hv, graw, scale, lo, total; };   // feat: dump region, as in ActLayout3
__host__ __device__ inline DeltaLayout3 delta_layout3(size_t P, bool two_word = false) {
    DeltaLayout3 a{};
    const size_t Pp = pad32(P);
    size_t o = 0;
    for (int i = 0; i < D; ++i) { a.h[i] = o; o += region_words3(Pp, W); }
    a.feat = o; o += DUMP_WORDS3;
    a.hv = o;   o += region_words3(Pp, WV);
    a.graw = o; o += region_words3(Pp, 4);
    o += 2048;          // the weight-gradient staging reads 64 rows of 128 B from a tile of this 4-row region
    a.scale = o; o += 4;
    a.lo = two_word ? o : 0;
    a.total = two_word ? 2 * o : o;
    return a;
}
// the scaled d_raw's largest magnitude: 2^DELTA_SCALE_TARGET_LOG2 <= s * max|d_raw| < 2 * that.  fp16 overflows at 2^16: a delta may
// grow 2^11-fold along the chain before it does (measured growth on the test scenes: <= 8); hi + lo is exact to 2^-25 absolute,
// i.e. 2^-29 of the largest upstream gradient.
constexpr int DELTA_SCALE_TARGET_LOG2 = 4;
// s = 2^k (inverse = false) or 2^-k (true) for a launch whose max|d_raw| has the bit pattern `amax_bits`; zero, subnormal or
// non-finite maxima give 1
__host__ __device__ inline unsigned delta_scale_bits(unsigned amax_bits, bool inverse) {
    const int e = (int)((amax_bits >> 23) & 0xffu);
    int k = (e == 0 || e == 255) ? 0 : DELTA_SCALE_TARGET_LOG2 - (e - 127);
    k = k < -100 ? -100 : (k > 100 ? 100 : k);
    return (unsigned)(127 + (inverse ? -k : k)) << 23;
}

}  // namespace nerf
