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
    size_t dir;         // [N][32]  dir encoding, canonical column order (cols 27..31 unused)
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
