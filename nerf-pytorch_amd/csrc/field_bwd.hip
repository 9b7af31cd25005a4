// Backward of the fused field evaluation (autograd of run_nerf_helpers.py:96-119
// restricted to the parameter gradients the reference trains with, SURVEY §8 a-9:
// no gradient flows into the encodings / ray points).
//
//   field_dgrad_kernel : d_raw[P,4] -> per-layer deltas (dL/d pre-activation),
//                        same register-resident transposed MFMA chain as the
//                        forward, fed by the transposed weight stream; ReLU masks
//                        come from the bitmasks the forward saved.
//   wgrad_kernel       : dW[n][k] = sum_p delta[p][n] * input[p][k]  and
//                        db[n] = sum_p delta[p][n]  for all 13 (delta, input)
//                        pairs of the network in ONE launch: LDS-tiled
//                        v_mfma_f32_16x16x4_f32 GEMM, contraction over points,
//                        split over point chunks; per-chunk partial gradients are
//                        written in canonical layout and summed by
//   wgrad_reduce_kernel (deterministic, no atomics).
#include <stdlib.h>
#include <type_traits>
#include "split_types.h"

#include "launchers.h"

// tuning knobs of tools/build_variant.sh (A/B timing on one box); the shipped values: tools/EXPERIMENTS.md, round 5
#ifndef NERF_WG_MERGE_ALPHA
#define NERF_WG_MERGE_ALPHA 1
#endif
#ifndef NERF_WG_REDUCE_BATCH
#define NERF_WG_REDUCE_BATCH 1
#endif
#ifndef NERF_WG1_STAGES
#define NERF_WG1_STAGES 4
#endif

namespace nerf {

// ------------------------------------------------------------------ dgrad chain
struct FieldBwdArgs {
    const float* packed;
    const float* act;       // saved by field_fwd_kernel<true>
    const float* d_raw;     // [P][4]
    float* delta;           // delta_layout(P)
    int n_rays, S;
};

template <int NV>
__device__ inline void apply_mask(float (&d)[NV], const f32x4* acc, uint2 m) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const unsigned bit = i < 32 ? (m.x >> i) & 1u : (m.y >> (i - 32)) & 1u;
        d[i] = bit ? acc[i >> 2][i & 3] : 0.0f;
    }
}

__global__ __launch_bounds__(FIELD_WAVES * 64) void field_dgrad_kernel(FieldBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane >> 4;
    const size_t P = (size_t)a.n_rays * a.S;
    const size_t p_raw = ((size_t)blockIdx.x * FIELD_WAVES + wave) * PTS_PER_WAVE + (lane & 15);
    const bool valid = p_raw < P;
    const size_t p = valid ? p_raw : P - 1;

    WeightStream<false> ws;
    ws.start(a.packed + BWD_VIEWS, lds, wave, lane);
    stage_small(a.packed, lds);

    const ActLayout al = act_layout(P, (size_t)a.n_rays);
    const DeltaLayout dl = delta_layout(P);
    const f32x4 g = *reinterpret_cast<const f32x4*>(a.d_raw + p * 4);       // (d_rgb3, d_sigma)
    uint2 msk[D + 1];
    {
        const uint2* mp = reinterpret_cast<const uint2*>(a.act + al.mask) + p * 4 + q;
#pragma unroll
        for (int l = 0; l <= D; ++l) msk[l] = mp[(size_t)l * P * 4];
    }

    // ---- rgb_linear^T (VALU) + ReLU mask of the view branch
    float dhv[32];
    {
        const float* wr = small_ptr(lds, SM_WRGB) + 4 * q;
        __syncthreads();                 // stage_small visible
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr + 16 * nb);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(wr + WV + 16 * nb);
            const f32x4 w2 = *reinterpret_cast<const f32x4*>(wr + 2 * WV + 16 * nb);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = g[0] * w0[r] + g[1] * w1[r] + g[2] * w2[r];
                dhv[4 * nb + r] = ((msk[D].x >> (4 * nb + r)) & 1u) ? v : 0.0f;
            }
        }
    }

    f32x4 acc[16];
    float d[64];
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- views_linears.0^T (feature columns only): 128 -> 256, no activation on feature
#pragma unroll
    for (int nb = 0; nb < 16; ++nb) acc[nb] = zero4;
    // every delta is stored one chunk late: right after the next acquire() issued its DMA (see field_fwd.hip)
    {
        const float* first = ws.acquire();
        if (valid) {
            float* o = a.delta + dl.hv + p * WV + 4 * q;
#pragma unroll
            for (int nb = 0; nb < 8; ++nb)
                nt_store(reinterpret_cast<f32x4*>(o + 16 * nb), f32x4{dhv[4 * nb], dhv[4 * nb + 1], dhv[4 * nb + 2], dhv[4 * nb + 3]});
        }
        mma_chunk<16, 16, 0, 32>(acc, dhv, first, lane);
    }
    mma_chunk<16, 16, 16, 32>(acc, dhv, ws.acquire(), lane);
#pragma unroll
    for (int i = 0; i < 64; ++i) d[i] = acc[i >> 2][i & 3];
    auto store_d = [&](size_t off) {
        if (valid) {
            float* o = a.delta + off + p * W + 4 * q;
#pragma unroll
            for (int nb = 0; nb < 16; ++nb)
                nt_store(reinterpret_cast<f32x4*>(o + 16 * nb), f32x4{d[4 * nb], d[4 * nb + 1], d[4 * nb + 2], d[4 * nb + 3]});
        }
    };

    // ---- feature_linear^T + alpha_linear^T, ReLU mask of layer 7
    {
        const float* wa = small_ptr(lds, SM_WALPHA) + 4 * q;
#pragma unroll
        for (int nb = 0; nb < 16; ++nb) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(wa + 16 * nb);
            acc[nb] = f32x4{g[3] * w[0], g[3] * w[1], g[3] * w[2], g[3] * w[3]};
        }
    }
    {
        const float* first = ws.acquire();
        store_d(dl.feat);
        mma_chunk<16, 16, 0, 64>(acc, d, first, lane);
    }
    mma_chunk<16, 16, 16, 64>(acc, d, ws.acquire(), lane);
    mma_chunk<16, 16, 32, 64>(acc, d, ws.acquire(), lane);
    mma_chunk<16, 16, 48, 64>(acc, d, ws.acquire(), lane);
    apply_mask<64>(d, acc, msk[D - 1]);

    // ---- trunk: delta_{l-1} = (W_l^T delta_l) * relu'(h_{l-1}),  l = 7 .. 1
#pragma unroll 1
    for (int l = D - 1; l >= 1; --l) {
#pragma unroll
        for (int nb = 0; nb < 16; ++nb) acc[nb] = zero4;
        {
            const float* first = ws.acquire();
            store_d((size_t)l * P * W);                                  // == dl.h[l]: delta of layer l (input of this step)
            mma_chunk<16, 16, 0, 64>(acc, d, first, lane);
        }
        mma_chunk<16, 16, 16, 64>(acc, d, ws.acquire(), lane);
        mma_chunk<16, 16, 32, 64>(acc, d, ws.acquire(), lane);
        mma_chunk<16, 16, 48, 64>(acc, d, ws.acquire(), lane);
        uint2 m = msk[0];
#pragma unroll
        for (int t = 1; t < D; ++t) if (t == l - 1) m = msk[t];      // select without dynamic register indexing
        apply_mask<64>(d, acc, m);
    }
    store_d(0);                                                          // dl.h[0]
}

// ------------------------------------------------------------------ weight gradients
constexpr int WG_TILE = 128;        // output tile 128 (n) x 128 (k) per 256-thread workgroup
constexpr int WG_STAGE = 32;        // points per LDS stage
constexpr int WG_MAX_JOBS = 14;       // 8 trunk layers (+1: layer 5 has two inputs) + feature + alpha + views (2 inputs) + rgb

struct WgradJob {
    const float* A; const float* B;
    int lda, nA, ldb, nB, b_rowdiv;
    int c_off, ldc, bias_off;       // offsets into the canonical gradient vector; bias_off < 0: none
    int tiles_k, tile_base;         // tiles of this job: [tile_base, tile_base + tiles_n*tiles_k)
    int vecA, vecB;                 // 16-byte aligned full-row loads allowed
    int b_tile16;                   // split datapaths: B is stored in 16-point tiles, row16h row order (the forward's saved rows)
    int b_ray_tiles;                // wgrad1_kernel: > 0 = B is constant along a ray (the direction encoding) and stored ONCE per
                                    // ray as [ray][feature][8 copies] bf16 (512 B per ray); value = 32-point tiles per ray
    // wgrad1_kernel: the LAST row of A (row a2_row = nA - 1 > 0) comes from a second region of the same buffer -- the alpha head's
    // d_sigma next to the view layer's 128 delta rows: both contract against h7, which is then read once instead of twice.
    int a2_row;                     // 0: none
    unsigned a2_off;                // bytes from A (tile 0) to that row in tile 0 of the second region
    int a2_tile_bytes;              // bytes per 32-point tile of the second region
    int c_off2, bias_off2;          // where that row's gradient goes: out[c_off2 + k], out[bias_off2]
};
struct WgradArgs {
    WgradJob job[WG_MAX_JOBS];
    int n_jobs, total_tiles;
    long P;
    int chunk_pts, n_chunks;
    float* partial;                 // [n_chunks][N_PARAMS]
    // wgrad1_kernel<SP, 3> (two-word operands, round 6): bytes from any A / B operand pointer to the LO words of the same operand
    // (the mirrors of the delta / save layouts: DeltaLayout3::lo, ActLayout3::lo)
    long a_lo_bytes, b_lo_bytes;
};

__device__ inline f32x4 load_row4(const float* base, int ld, long row, bool row_ok, int col, int ncols, int vec) {
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (row_ok) {
        const float* ptr = base + row * ld + col;
        if (vec && col + 3 < ncols) v = *reinterpret_cast<const f32x4*>(ptr);
        else {
            if (col < ncols) v[0] = ptr[0];
            if (col + 1 < ncols) v[1] = ptr[1];
            if (col + 2 < ncols) v[2] = ptr[2];
            if (col + 3 < ncols) v[3] = ptr[3];
        }
    }
    return v;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// One 128 x 128 output tile of a narrow job for one point chunk.  WN x WK = how the four waves share the tile: 2 x 2 (wave tile
// 64 n x 64 k), 4 x 1 (32 n x 64 k: jobs whose tile has <= 64 valid k columns -- the encodings -- where the 2 x 2 form leaves the
// two wave_k = 1 waves, i.e. two of the CU's four matrix pipes, without work), 1 x 4 (64 n x 32 k: the one-row / three-row jobs of
// the heads).  Lane c of a wave holds NI (NJ) consecutive n (k) columns -- the NI x NJ blocks of 16 x 16 outputs interleave -- so a
// fragment is one ds_read of 8 or 16 bytes.  Every output element accumulates its points in the same order in every form:
// results are bit-identical to the 2 x 2 form.
template <int WN, int WK>
__device__ __forceinline__ void wgrad_tile(const WgradArgs& a, const WgradJob& jb, float (*sA)[WG_STAGE][WG_TILE], float (*sB)[WG_STAGE][WG_TILE],
                                           int n0, int k0, int tk, int chunk) {
    constexpr int NI = WN == 4 ? 2 : 4, NJ = WK == 4 ? 2 : 4;       // 16-row blocks per wave along n / k (= floats per fragment read)
    constexpr int WAVE_N = 16 * NI, WAVE_K = 16 * NJ;               // wave tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_n = WN == 1 ? 0 : (WN == 4 ? wave : wave >> 1), wave_k = WK == 1 ? 0 : (WK == 4 ? wave : wave & 1);
    const long p_begin = (long)chunk * a.chunk_pts;
    const long p_end = min(p_begin + (long)a.chunk_pts, a.P);
    const int n_stages = (int)((p_end - p_begin + WG_STAGE - 1) / WG_STAGE);

    // staging map: thread -> rows (tid>>5) + 8*i, columns 4*(tid&31)
    const int srow = tid >> 5, scol = (tid & 31) * 4;
    const float* Ab = jb.A + n0;
    const float* Bb = jb.B + k0;
    const int nA = jb.nA - n0, nB = jb.nB - k0;          // valid columns left in this tile
    f32x4 ra[4], rb[4];
    auto gload = [&](int st) {
        const long r0 = p_begin + (long)st * WG_STAGE + srow;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long r = r0 + 8 * i;
            ra[i] = load_row4(Ab, jb.lda, r, r < p_end, scol, nA, jb.vecA);
            rb[i] = load_row4(Bb, jb.ldb, jb.b_rowdiv > 1 ? r / jb.b_rowdiv : r, r < p_end, scol, nB, jb.vecB);
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(&sA[buf][srow + 8 * i][scol]) = ra[i];
            *reinterpret_cast<f32x4*>(&sB[buf][srow + 8 * i][scol]) = rb[i];
        }
    };

    f32x4 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) bsum[i] = 0.f;

    gload(0);
    swrite(0);
    __syncthreads();
    const int c = lane & 15, pp = lane >> 4;
    // a wave whose whole n- or k-range is zero padding only helps with staging
    const bool wave_has_work = (wave_n * WAVE_N < nA) && (wave_k * WAVE_K < nB);
    for (int st = 0; st < n_stages; ++st) {
        const int buf = st & 1;
        if (st + 1 < n_stages) gload(st + 1);
        if (wave_has_work) {
#pragma unroll
            for (int ps = 0; ps < WG_STAGE / 4; ++ps) {
                float av[NI], bv[NJ];
                const float* ap = &sA[buf][4 * ps + pp][wave_n * WAVE_N + NI * c];
                const float* bp = &sB[buf][4 * ps + pp][wave_k * WAVE_K + NJ * c];
                if constexpr (NI == 4) { const f32x4 t = *reinterpret_cast<const f32x4*>(ap); av[0] = t[0]; av[1] = t[1]; av[2] = t[2]; av[3] = t[3]; }
                else { const f32x2 t = *reinterpret_cast<const f32x2*>(ap); av[0] = t[0]; av[1] = t[1]; }
                if constexpr (NJ == 4) { const f32x4 t = *reinterpret_cast<const f32x4*>(bp); bv[0] = t[0]; bv[1] = t[1]; bv[2] = t[2]; bv[3] = t[3]; }
                else { const f32x2 t = *reinterpret_cast<const f32x2*>(bp); bv[0] = t[0]; bv[1] = t[1]; }
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    bsum[i] += av[i];
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        if (st + 1 < n_stages) swrite(buf ^ 1);
        __syncthreads();
    }

    // acc[i][j][r] = dW[n0 + wave_n*WAVE_N + NI*(4*pp + r) + i][k0 + wave_k*WAVE_K + NJ*c + j]
    float* out = a.partial + (size_t)chunk * N_PARAMS;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = wave_n * WAVE_N + NI * (4 * pp + r) + i;
            if (n < nA) {
                float* row = out + jb.c_off + (size_t)(n0 + n) * jb.ldc + k0;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int k = wave_k * WAVE_K + NJ * c + j;
                    if (k < nB) row[k] = acc[i][j][r];
                }
            }
        }
    if (jb.bias_off >= 0 && tk == 0 && wave_k == 0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float v = bsum[i];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            const int n = wave_n * WAVE_N + NI * c + i;
            if (pp == 0 && n < nA) out[jb.bias_off + n0 + n] = v;
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
    __shared__ __attribute__((aligned(16))) float sA[2][WG_STAGE][WG_TILE];
    __shared__ __attribute__((aligned(16))) float sB[2][WG_STAGE][WG_TILE];
    const int tile = blockIdx.x % a.total_tiles;
    const int chunk = blockIdx.x / a.total_tiles;
    int ji = 0;
#pragma unroll 1
    for (int j = 1; j < a.n_jobs; ++j) if (tile >= a.job[j].tile_base) ji = j;
    const WgradJob& jb = a.job[ji];
    const int lt = tile - jb.tile_base;
    const int tn = lt / jb.tiles_k, tk = lt % jb.tiles_k;
    const int n0 = tn * WG_TILE, k0 = tk * WG_TILE;
    const int nA = jb.nA - n0, nB = jb.nB - k0;          // valid columns of this tile: how the four waves share it
    if (nB <= 64 && nA > 64) wgrad_tile<4, 1>(a, jb, sA, sB, n0, k0, tk, chunk);
    else if (nA <= 64 && nB > 64) wgrad_tile<1, 4>(a, jb, sA, sB, n0, k0, tk, chunk);
    else wgrad_tile<2, 2>(a, jb, sA, sB, n0, k0, tk, chunk);
}

// Full-width jobs (256 x 256 outputs, 16-byte aligned rows): one workgroup owns the WHOLE output of a job for its
// point chunk, so delta and input rows are fetched from HBM exactly once (the 128x128 kernel above reads each twice).
// 8 waves = 2 (n) x 4 (k), wave tile 128 x 64 = 32 accumulator blocks, 3 ds_read_b128 per 32 MFMAs.
constexpr int WG256_LDS_FLOATS = 2 * 2 * WG_STAGE * 256;        // sA[2][32][256] | sB[2][32][256] = 128 KiB

__global__ __launch_bounds__(512) void wgrad256_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm256[];
    float (*sA)[WG_STAGE][256] = reinterpret_cast<float (*)[WG_STAGE][256]>(sm256);
    float (*sB)[WG_STAGE][256] = reinterpret_cast<float (*)[WG_STAGE][256]>(sm256 + 2 * WG_STAGE * 256);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_n = wave >> 2, wave_k = wave & 3;
    const int ji = blockIdx.x % a.n_jobs;
    const int chunk = blockIdx.x / a.n_jobs;
    const WgradJob& jb = a.job[ji];
    const long p_begin = (long)chunk * a.chunk_pts;
    const long p_end = min(p_begin + (long)a.chunk_pts, a.P);
    const int n_stages = (int)((p_end - p_begin + WG_STAGE - 1) / WG_STAGE);
    const int srow = tid >> 6, scol = (tid & 63) * 4;
    f32x4 ra[4], rb[4];
    auto gload = [&](int st) {
        const long r0 = p_begin + (long)st * WG_STAGE + srow;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long r = r0 + 8 * i;
            const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            ra[i] = r < p_end ? *reinterpret_cast<const f32x4*>(jb.A + r * jb.lda + scol) : z;
            rb[i] = r < p_end ? *reinterpret_cast<const f32x4*>(jb.B + r * jb.ldb + scol) : z;
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(&sA[buf][srow + 8 * i][scol]) = ra[i];
            *reinterpret_cast<f32x4*>(&sB[buf][srow + 8 * i][scol]) = rb[i];
        }
    };
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bsum0 = f32x4{0.f, 0.f, 0.f, 0.f}, bsum1 = bsum0;
    gload(0);
    swrite(0);
    __syncthreads();
    const int c = lane & 15, pp = lane >> 4;
    for (int st = 0; st < n_stages; ++st) {
        const int buf = st & 1;
        if (st + 1 < n_stages) gload(st + 1);
#pragma unroll
        for (int ps = 0; ps < WG_STAGE / 4; ++ps) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(&sA[buf][4 * ps + pp][wave_n * 128 + 4 * c]);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(&sA[buf][4 * ps + pp][wave_n * 128 + 64 + 4 * c]);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(&sB[buf][4 * ps + pp][wave_k * 64 + 4 * c]);
            bsum0 += a0;
            bsum1 += a1;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i], bv[j], acc[i][j], 0, 0, 0);
                    acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i], bv[j], acc[4 + i][j], 0, 0, 0);
                }
        }
        if (st + 1 < n_stages) swrite(buf ^ 1);
        __syncthreads();
    }
    // acc[4h+i][j][r] = dW[wave_n*128 + 64h + 4*(4*pp + r) + i][wave_k*64 + 4*c + j]
    float* out = a.partial + (size_t)chunk * N_PARAMS;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = wave_n * 128 + 64 * h + 4 * (4 * pp + r) + i;
                float* row = out + jb.c_off + (size_t)n * jb.ldc + wave_k * 64 + 4 * c;
#pragma unroll
                for (int j = 0; j < 4; ++j) row[j] = acc[4 * h + i][j][r];
            }
    if (jb.bias_off >= 0 && wave_k == 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = h == 0 ? bsum0[i] : bsum1[i];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if (pp == 0) out[jb.bias_off + wave_n * 128 + 64 * h + 4 * c + i] = v;
            }
    }
}

// ------------------------------------------------------------------ streaming weight gradients of the split datapaths (16-bit operands)
// 12 jobs (13 before the alpha head's row joined the view layer's job), operands saved as 16-bit elements (bf16 or fp16: SP) in 32-point feature-major tiles (64-byte rows): a lane's MFMA
// fragment (8 consecutive points of one feature = 16 B) is in memory as is, so the kernel is pure streaming:
// HBM -> LDS by DMA into a ring of 4 stages (one stage = one 32-point tile of both operands = 32 KiB = 32
// wave-instructions of global_load_lds_dwordx4, three stages = 96 KiB per CU in flight), ds_read_b128, one bf16 MFMA
// per product, fp32 accumulation.  No VALU conversion, no staging registers; 11.4 KB per point instead of 22.7.
// Every DMA lane picks its own 16-byte source piece, which XOR-swizzles the four pieces of a row inside its 64 LDS
// bytes: the fragment reads are bank-conflict free.  Bias gradients = fp32 row sums of the delta fragments.
constexpr int WG1_STAGE_PTS = 32;
constexpr int WG1_OP_BYTES = 256 * 64;                           // 16 KiB: [256 features][32 points] bf16
constexpr int WG1_STAGE_BYTES = 2 * WG1_OP_BYTES;
constexpr int WG1_STAGES = NERF_WG1_STAGES;                      // 4: 128 KiB of LDS, three stages (96 KiB) in flight; 5: all 160 KiB, four in flight
constexpr int WG1_LDS_BYTES = WG1_STAGES * WG1_STAGE_BYTES;

__device__ inline void dma_1k_s(const void* sbase, unsigned voff, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2 nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst_uniform)
        : "memory");
}
// SP: element type of the stored operands (split_types.h: bf16, or fp16 = the hi words of the fp16 split -- then every delta
// carries the launch's power-of-two scale, which wgrad_reduce_kernel removes)
// TERMS = 1: the operands are the stored hi words (11 / 8 significant bits).  TERMS = 3 (round 6, "fp16x3w"): TWO-WORD operands --
// dW = d_hi^T X_hi + d_hi^T X_lo + d_lo^T X_hi, the same three-term product as the forward's.  A stage is then one 32-point tile of
// FOUR operand blocks [d_hi | X_hi | d_lo | X_lo] = 64 KiB (the lo words live at a_lo_bytes / b_lo_bytes from the hi words), two stages
// in the same 128 KiB of LDS, every word fetched from HBM once; a k-step reads 12 fragments and issues 24 MFMAs.  (First form, measured:
// three 32-KiB stages per tile -- hi.hi, hi.lo, lo.hi -- re-fetching the hi tiles: 3.05x the one-word kernel's time, i.e. bound by its
// 3x DMA traffic, which L2 does not absorb; this form moves 2x the bytes.)
template <typename SP, int TERMS = 1>
__global__ __launch_bounds__(512) void wgrad1_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm1[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_n = wave >> 2, wave_k = wave & 3;
    const int ji = blockIdx.x % a.n_jobs;
    const int chunk = blockIdx.x / a.n_jobs;
    const WgradJob& jb = a.job[ji];
    const long p_begin = (long)chunk * a.chunk_pts;              // multiple of 32: chunks start on a tile
    const long p_end = min(p_begin + (long)a.chunk_pts, a.P);
    const int nrows = (int)(p_end - p_begin);
    const int n_tiles = (nrows + WG1_STAGE_PTS - 1) / WG1_STAGE_PTS;
    const int n_st = n_tiles;                                    // one stage per 32-point tile
    constexpr int STAGE_BYTES = TERMS == 3 ? 2 * WG1_STAGE_BYTES : WG1_STAGE_BYTES;      // 64 KiB: [d_hi | X_hi | d_lo | X_lo]
    constexpr int STAGES = TERMS == 3 ? WG1_LDS_BYTES / STAGE_BYTES : WG1_STAGES;        // 2 (one in flight) / 4 (three in flight)

    // DMA role: waves 0-3 copy delta (A), waves 4-7 the input (B); instruction u of a wave copies LDS pieces
    // [64*m, 64*m + 64) of its operand, m = 4*(wave&3) + u; LDS piece 4*f + jj holds points 8*(jj ^ swz(f))..+7 of
    // feature f, swz(f) = (f >> 2) & 3.  jb.A / jb.B point at bf16 data, lda / ldb = features per tile.
    const int sop = wave >> 2;
    const int swidth = sop == 0 ? jb.nA : jb.nB;
    const int sld = sop == 0 ? jb.lda : jb.ldb;
    const unsigned tile_bytes = 64u * (unsigned)sld;
    const char* cbase = reinterpret_cast<const char*>(sop == 0 ? jb.A : jb.B) + (size_t)(p_begin >> 5) * tile_bytes;
    // B operands saved by the 16-point forward: the stage's 32 points are two consecutive 16-point tiles of sld rows x
    // 32 B (row16h order, nerf_common.h); 8-point group g of feature f sits at tile (g >> 1), row16h(f), bytes 16 * (g & 1)
    const bool t16 = sop == 1 && jb.b_tile16 != 0;
    // B constant along a ray: every 8-point group of feature f of a tile is the ray's 16-byte record of f (8 copies of the
    // value); the stage's source is the record block of the ray its tile belongs to
    const int ray_tiles = sop == 1 ? jb.b_ray_tiles : 0;
    unsigned doff[4], dstep[4];     // source offset of the lane's piece in stage 0 of this chunk, and what it moves by per stage on top of the tile pitch
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = 64 * (4 * (wave & 3) + u) + lane;
        const int f = q >> 2, jj = q & 3;
        const unsigned fc = (unsigned)min(f, swidth - 1), g = (unsigned)(jj ^ ((f >> 2) & 3));
        doff[u] = ray_tiles > 0 ? 16u * fc
                : t16 ? (g >> 1) * 32u * (unsigned)sld + 32u * (unsigned)row16h((int)fc) + 16u * (g & 1u) : 64u * fc + 16u * g;
        dstep[u] = 0u;
        if (sop == 0 && jb.a2_row > 0 && f >= jb.a2_row) {
            // rows from a2_row on (the pad rows repeat it): second region, whose tiles are a2_tile_bytes apart.  Unsigned wrap-around
            // arithmetic: the sums below are the true (non-negative, < 4 GiB) offsets from this chunk's first tile of A
            dstep[u] = (unsigned)jb.a2_tile_bytes - tile_bytes;
            doff[u] = jb.a2_off + (unsigned)(p_begin >> 5) * dstep[u] + 16u * g;
        }
    }
    const char* rbase = reinterpret_cast<const char*>(jb.B);
    const long tile0 = p_begin >> 5;
    const unsigned lds0 = lds_addr(sm1) + (unsigned)(sop * WG1_OP_BYTES + (wave & 3) * 4096);
    const long lo_bytes = TERMS == 3 ? (sop == 0 ? a.a_lo_bytes : a.b_lo_bytes) : 0;     // this wave's operand: hi -> lo words
    auto issue = [&](int st) {
        const char* src = ray_tiles > 0 ? rbase + (size_t)((tile0 + st) / ray_tiles) * (size_t)(16 * sld) : cbase + (size_t)st * tile_bytes;
        const unsigned dst = lds0 + (unsigned)(st % STAGES) * STAGE_BYTES;
#pragma unroll
        for (int u = 0; u < 4; ++u) dma_1k_s(src, doff[u] + (unsigned)st * dstep[u], dst + 1024u * u);
        if constexpr (TERMS == 3) {             // the lo words of the same pieces, behind the hi blocks of the stage
#pragma unroll
            for (int u = 0; u < 4; ++u) dma_1k_s(src + lo_bytes, doff[u] + (unsigned)st * dstep[u], dst + WG1_STAGE_BYTES + 1024u * u);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float rowsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int row = lane & 31, hf = lane >> 5;
    const int ni = min(4, (jb.nA - wave_n * 128 + 31) / 32);     // 32-row output blocks of this wave that hold real rows
    const int nj = min(2, (jb.nB - wave_k * 64 + 31) / 32);
    const bool wave_has_work = ni > 0 && nj > 0;
    const bool want_bias = jb.bias_off >= 0 && wave_k == 0;
    const int fswz = (row >> 2) & 3;
    // the lane's fragment of k-step t (points 16*t + 8*hf .. +7) = source piece 2*t + hf of feature `row` of a block
    const int frag[2] = {row * 64 + ((hf ^ fswz) << 4), row * 64 + (((2 + hf) ^ fswz) << 4)};
    auto fragment = [&](const unsigned char* blk, int t, int left) {       // left: points of this stage < P
        u32x4 w = *reinterpret_cast<const u32x4*>(blk + frag[t]);
        if (left < WG1_STAGE_PTS) {         // ragged last stage of the last chunk: the pad points were never written
            const int p0 = 16 * t + 8 * hf;
#pragma unroll
            for (int d = 0; d < 4; ++d)
                w[d] &= (p0 + 2 * d < left ? 0x0000ffffu : 0u) | (p0 + 2 * d + 1 < left ? 0xffff0000u : 0u);
        }
        return w;
    };
    auto compute = [&](int st) {
        if (!wave_has_work) return;
        const unsigned char* stage = sm1 + (st % STAGES) * STAGE_BYTES;
        const unsigned char* sa = stage + (wave_n * 128) * 64;
        const unsigned char* sb = stage + WG1_OP_BYTES + (wave_k * 64) * 64;
        const int left = nrows - st * WG1_STAGE_PTS;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            u32x4 bf[2], bl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bf[j] = fragment(sb + j * 32 * 64, t, left);
                if constexpr (TERMS == 3) bl[j] = fragment(sb + WG1_STAGE_BYTES + j * 32 * 64, t, left);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i >= ni) break;
                const u32x4 af = fragment(sa + i * 32 * 64, t, left);
                if (want_bias) rowsum[i] += SP::sum8(af);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = SP::mfma(af, bf[j], acc[i][j]);
                if constexpr (TERMS == 3) {     // + d_hi X_lo + d_lo X_hi;  db = sum (d_hi + d_lo)
                    const u32x4 al = fragment(sa + WG1_STAGE_BYTES + i * 32 * 64, t, left);
                    if (want_bias) rowsum[i] += SP::sum8(al);
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = SP::mfma(af, bl[j], acc[i][j]);
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = SP::mfma(al, bf[j], acc[i][j]);
                }
            }
        }
    };

    // The common case — a full 128 x 64 wave tile, all 32 points of the stage exist — as straight-line code: all twelve
    // fragment reads of a k-step are issued together, then its 8 MFMAs.  (compute() above decides per
    // fragment whether its rows exist, whether the stage is ragged and whether bias sums are wanted: a branch, i.e. a
    // scheduling barrier, between every pair of MFMAs, each pair waiting for its own ds_read_b128.)
    auto compute_full = [&](int st, auto with_bias) {
        const unsigned char* stage = sm1 + (st % STAGES) * STAGE_BYTES;
        const unsigned char* sa = stage + (wave_n * 128) * 64;
        const unsigned char* sb = stage + WG1_OP_BYTES + (wave_k * 64) * 64;
#pragma unroll
        for (int t = 0; t < 2; ++t) {           // six reads, then eight MFMAs, per k-step (all twelve reads up front spill)
            u32x4 bf[2], af[4];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const u32x4*>(sb + j * 32 * 64 + frag[t]);
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const u32x4*>(sa + i * 32 * 64 + frag[t]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (decltype(with_bias)::value) rowsum[i] += SP::sum8(af[i]);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = SP::mfma(af[i], bf[j], acc[i][j]);
            }
            if constexpr (TERMS == 3) {
                // + d_hi X_lo (two more reads, eight MFMAs), + d_lo X_hi (four more reads, eight MFMAs): at most ten fragments live
                u32x4 bl[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) bl[j] = *reinterpret_cast<const u32x4*>(sb + WG1_STAGE_BYTES + j * 32 * 64 + frag[t]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = SP::mfma(af[i], bl[j], acc[i][j]);
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const u32x4*>(sa + WG1_STAGE_BYTES + i * 32 * 64 + frag[t]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (decltype(with_bias)::value) rowsum[i] += SP::sum8(af[i]);       // db = sum (d_hi + d_lo)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = SP::mfma(af[i], bf[j], acc[i][j]);
                }
            }
        }
    };
    const bool full_tile = wave_has_work && ni == 4 && nj == 2;                 // wave-uniform
    const int n_full = (nrows % WG1_STAGE_PTS == 0) ? n_tiles : n_tiles - 1;     // stages whose 32 points all exist

    // ring: stages st+1 .. st+3 are in flight while st is consumed.  vmcnt retires in order: "at most 4 * (younger
    // stages in flight) outstanding" = this wave's pieces of st have landed.  (Two-word form: two 64-KiB stages, st+1 in flight.)
#pragma unroll
    for (int st = 0; st < STAGES - 1; ++st)
        if (st < n_st) issue(st);
    auto enter = [&](int st) {
        if constexpr (TERMS == 3) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // stage st is the only one in flight here
        } else {
            if (WG1_STAGES >= 5 && st + 3 < n_st) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (st + 2 < n_st) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (st + 1 < n_st) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();        // every wave's pieces landed; everybody is done with stage st-1, whose slot is reused now
        if (st + STAGES - 1 < n_st) issue(st + STAGES - 1);
    };
    // one loop per variant (wave-uniform choice): both bodies inside one loop cost 116 spilled VGPRs
    int st = 0;
    if (full_tile && want_bias) for (; st < n_full; ++st) { enter(st); compute_full(st, std::true_type{}); }
    else if (full_tile) for (; st < n_full; ++st) { enter(st); compute_full(st, std::false_type{}); }
    for (; st < n_st; ++st) { enter(st); compute(st); }

    // acc[i][j][r] at lane (col = lane&31, hf) = dW[wave_n*128 + 32*i + d32row(r, hf)][wave_k*64 + 32*j + col]
    float* out = a.partial + (size_t)chunk * N_PARAMS;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = wave_n * 128 + 32 * i + d32row(r, hf);
            if (n < jb.nA) {
                float* orow = (jb.a2_row > 0 && n == jb.a2_row) ? out + jb.c_off2 : out + jb.c_off + (size_t)n * jb.ldc;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int k = wave_k * 64 + 32 * j + row;
                    if (k < jb.nB) orow[k] = acc[i][j][r];
                }
            }
        }
    if (want_bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float t = rowsum[i] + __shfl_xor(rowsum[i], 32);
            const int n = wave_n * 128 + 32 * i + row;
            if (hf == 0 && n < jb.nA) out[(jb.a2_row > 0 && n == jb.a2_row) ? jb.bias_off2 : jb.bias_off + n] = t;
        }
    }
}

// fold == 0: every entry of the canonical gradient is the sum of its per-chunk partials.
// fold == 1 (split datapaths, feature layer folded into the view branch, nerf_common.h): the job (delta_hv, h7) left
//   G = delta_hv^T h7 in the slot of views_linears.0.weight[:, :256]; G and this call's dbv = sum delta_hv go to
//   `scratch` ([128][256] | [128]) for wgrad_fold_kernel, and feature_linear.{weight,bias} / Wv[:, :256] are left to it.
// amax (nullable): device word holding the bit pattern of the launch's max|d_raw| (fp16 split, DeltaLayout3::scale): every partial
// sum carries the factor s = delta_scale_bits(amax) = 2^k, removed here exactly.
__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, int n_chunks, float* __restrict__ grad, int accumulate,
                                    int fold, float* __restrict__ scratch, const unsigned* __restrict__ amax) {
    constexpr Canon cn = canon();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N_PARAMS) return;
    if (fold && i >= cn.wf && i < cn.bf + W) return;                    // produced by wgrad_fold_kernel (no partials exist)
    // loads in batches of 8 (all issued before the first is added: the loop is latency-bound otherwise), sums in chunk order
    float s = 0.0f;
    const float* src = partial + i;
    int cix = 0;
#if NERF_WG_REDUCE_BATCH
    for (; cix + 8 <= n_chunks; cix += 8) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = src[(size_t)(cix + t) * N_PARAMS];
#pragma unroll
        for (int t = 0; t < 8; ++t) s += v[t];
    }
#endif
    for (; cix < n_chunks; ++cix) s += src[(size_t)cix * N_PARAMS];
    if (amax) s *= __uint_as_float(delta_scale_bits(amax[0], true));
    if (fold) {
        if (i >= cn.wv && i < cn.bv) {
            const int k = (i - cn.wv) / (W + IN_DIR), col = (i - cn.wv) % (W + IN_DIR);
            if (col < W) { scratch[k * W + col] = s; return; }         // G[k][col]
        } else if (i >= cn.bv && i < cn.bv + WV) {
            scratch[WV * W + (i - cn.bv)] = s;                          // this call's dbv (grad may accumulate)
        }
    }
    grad[i] = accumulate ? grad[i] + s : s;
}

// Gradients of the two layers the split datapaths evaluate as one (nerf_common.h, folded feature layer), from
// G = delta_hv^T h7 [128][256] and dbv [128]:   dWv[:, :256] = G Wf^T + dbv bf^T,  dWf = Wv[:, :256]^T G,
// dbf = Wv[:, :256]^T dbv.   Plain fp32 FMA loops (K = 256 / 128): 16.8 MFLOP, one thread per output
// (eight lanes per output with a shuffle reduction, as derive_folded_kernel does, measured slower: 17.7 vs 14.2 us per launch; four
// interleaved partial sums per output instead of one 256-long FMA chain: no change, round 5).
__global__ void wgrad_fold_kernel(const float* __restrict__ params, const float* __restrict__ scratch, float* __restrict__ grad,
                                  int accumulate) {
    constexpr Canon cn = canon();
    const float* G = scratch;
    const float* dbv = scratch + WV * W;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    float v;
    int dst;
    if (idx < WV * W) {                                 // dWv[k][i], i < 256
        const int k = idx / W, i = idx % W;
        const float* wf = params + cn.wf + i * W;
        float acc = 0.0f;
        for (int j = 0; j < W; ++j) acc = fmaf(G[k * W + j], wf[j], acc);
        v = fmaf(dbv[k], params[cn.bf + i], acc);
        dst = cn.wv + k * (W + IN_DIR) + i;
    } else if (idx < WV * W + W * W) {                  // dWf[i][j]
        const int t = idx - WV * W, i = t / W, j = t % W;
        float acc = 0.0f;
        for (int k = 0; k < WV; ++k) acc = fmaf(params[cn.wv + k * (W + IN_DIR) + i], G[k * W + j], acc);
        v = acc;
        dst = cn.wf + t;
    } else if (idx < WV * W + W * W + W) {              // dbf[i]
        const int i = idx - WV * W - W * W;
        float acc = 0.0f;
        for (int k = 0; k < WV; ++k) acc = fmaf(params[cn.wv + k * (W + IN_DIR) + i], dbv[k], acc);
        v = acc;
        dst = cn.bf + i;
    } else return;
    grad[dst] = accumulate ? grad[dst] + v : v;
}

// ------------------------------------------------------------------ host side
static int wgrad_chunks(long P, int* chunk_pts, int n_jobs = 14) {
    // point chunks such that jobs x chunks fills whole rounds of 256 workgroups (one workgroup per CU).  fp32: 14 jobs x
    // 128 = 7 x 256 (the 8 full-width jobs x 128 = 4 x 256).  Split datapaths (12 jobs, 16-bit operands, wgrad1_kernel): 12 x 21 = 252
    // workgroups (13 jobs: 13 x 19 = 247) = ONE round for every launch size.  Rounds 2-4 gave launches above 400 k points 39 chunks (507 workgroups = two
    // rounds); measured in round 5 on one box, alternating builds, fine launch (786,432 points): 39 chunks GEMM 1.312 ms + reduction
    // 0.037 ms; 26 chunks (1.3 rounds) 1.726 + 0.030; 19 chunks 1.281 + 0.026 -- a workgroup streaming twice the points keeps its
    // 96 KiB of DMA in flight just the same, and the deterministic reduction reads half the partial sums.  (Pairing the narrow jobs
    // into one workgroup so that all workgroups stream equal bytes -- 11 "virtual jobs" x 23 chunks -- measured 1.84 ms: the job loop
    // costs the kernel its schedule, 192 -> 255 VGPRs; not kept.)  Small inputs get >= 256-point chunks.
    long n = n_jobs == 14 ? 128 : 256 / n_jobs;       // 12 jobs: 21 chunks, 13 jobs: 19
    const long cap = (P + 255) / 256;
    if (n > cap) n = cap;
    if (n < 1) n = 1;
    long pts = (P + n - 1) / n;
    pts = (pts + WG_STAGE - 1) / WG_STAGE * WG_STAGE;
    *chunk_pts = (int)pts;
    return (int)((P + pts - 1) / pts);
}

size_t wgrad_partial_floats(long P) {
    // sized for either job count, plus the scratch of the folded feature layer (G | dbv) behind the partial sums
    int pts;
    const int n14 = wgrad_chunks(P, &pts, 14), n13 = wgrad_chunks(P, &pts, 12);      // (12 jobs: the most chunks a split launch uses)
    return (size_t)(n14 > n13 ? n14 : n13) * N_PARAMS + N_DERIVED;
}

hipError_t launch_field_dgrad(const float* packed, const float* act, const float* d_raw, int n_rays, int S,
                              float* delta, hipStream_t stream) {
    const long P = (long)n_rays * S;
    if (P <= 0) return hipSuccess;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)field_dgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FIELD_LDS_FLOATS * 4);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    FieldBwdArgs ba{packed, act, d_raw, delta, n_rays, S};
    const unsigned blocks = (unsigned)((P + PTS_PER_WG - 1) / PTS_PER_WG);
    hipLaunchKernelGGL(field_dgrad_kernel, dim3(blocks), dim3(FIELD_WAVES * 64), FIELD_LDS_FLOATS * 4, stream, ba);
    return hipGetLastError();
}

// per-ray view-direction encoding [N][32] -> per point [P][32] (float4 per thread)
__global__ void expand_dir_kernel(const f32x4* __restrict__ dir_ray, f32x4* __restrict__ dir_pt, long P, int S) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * 8) return;
    const long p = i >> 3;
    dir_pt[i] = dir_ray[(p / S) * 8 + (i & 7)];
}
// the same into 32-point feature-major tiles of 32 features with 16-bit elements (split datapaths): thread = (tile, feature, 8-point group)
template <typename SP>
__device__ inline u32x4 pack8_sp(const float* v) {
    return u32x4{SP::cvt_pk(v[0], v[1]), SP::cvt_pk(v[2], v[3]), SP::cvt_pk(v[4], v[5]), SP::cvt_pk(v[6], v[7])};
}
// LO: the remainders T(v - hi) instead of the hi words (two-word operands)
template <typename SP>
__device__ inline float split_rem(float x) {
    const unsigned short h = SP::cvt1(x);
    if constexpr (SP::F16) return x - (float)__builtin_bit_cast(_Float16, h);
    else return x - __uint_as_float((unsigned)h << 16);
}
template <typename SP, bool LO = false>
__global__ void expand_dir_tiles_bf16_kernel(const float* __restrict__ dir_ray, u32x4* __restrict__ dir_pt, long P, int S) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n_tiles = (P + 31) >> 5;
    if (i >= n_tiles * 128) return;
    const long tile = i >> 7;
    const int f = (int)(i >> 2) & 31, g = (int)i & 3;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const long p = min(tile * 32 + 8 * g + e, P - 1);
        v[e] = dir_ray[(p / S) * 32 + f];
        if (LO) v[e] = split_rem<SP>(v[e]);
    }
    dir_pt[i] = pack8_sp<SP>(v);
}

// ... or, when a ray's samples fill whole 32-point tiles, ONCE per ray: [ray][feature][8 copies] bf16 -- 16 bytes per (ray,
// feature), which is the fragment wgrad1_kernel's DMA fetches for every 8-point group of that ray (2 MB instead of 50 MB for
// 4096 x 192 points, and no per-point kernel)
template <typename SP, bool LO = false>
__global__ void replicate_dir_bf16_kernel(const float* __restrict__ dir_ray, u32x4* __restrict__ dir_rep, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // (ray, feature)
    if (i >= n) return;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = LO ? split_rem<SP>(dir_ray[i]) : dir_ray[i];
    dir_rep[i] = pack8_sp<SP>(v);
}

// phases: bit 0 = full-width jobs, bit 1 = narrow jobs, bit 2 = chunk reduction (7 = everything)
hipError_t launch_field_wgrad(const float* act, const float* delta, const float* d_raw, int n_rays, int S,
                              float* partial, float* grad, int accumulate, int datapath, int phases, hipStream_t stream,
                              const float* params) {
    // datapath: 0 = fp32 (point-major rows); 4 = bf16 operands, rows saved by the 16-point forward (16-point tiles, row16h order),
    // deltas in 32-point tiles; 5 = the same with fp16 elements (deltas scaled by the launch's power of two, removed in the reduction);
    // 6 = fp16 TWO-WORD operands (hi and lo words saved by the SAVE = 3 forward and the TWO dgrad: three MFMAs per product)
    if (datapath != 0 && datapath != 4 && datapath != 5 && datapath != 6) return hipErrorInvalidValue;
    const bool two = datapath == 6;
    const bool f16 = datapath == 5 || two;
    const bool split16 = datapath != 0;      // split datapaths: 16-bit operands streamed by wgrad1_kernel,
    const bool fold = split16;               //   feature layer folded into the view branch (nerf_common.h)
    if (fold && !params) return hipErrorInvalidValue;
    const long P = (long)n_rays * S;
    if (P <= 0) return hipSuccess;
    hipError_t e;
    constexpr Canon cn = canon();
    // operand bases.  fp32 datapath: point-major rows (lda = row pitch); split datapaths: 32-point feature-major tiles
    // (lda = features per tile, a feature offset f0 is folded into the base as f0 * 32)
    const float *d_h[D], *d_feat, *d_hv, *d_rgb, *d_sigma, *x_h[D], *x_feat, *x_hv, *x_enc, *x_dir;
    const unsigned* amax = nullptr;
    int ld_graw;
    long lo_a = 0, lo_b = 0;        // two-word operands: bytes from the hi words of a delta / save region to its lo words
    if (split16) {
        const ActLayout3 al = act_layout3((size_t)P, (size_t)n_rays, two);
        const DeltaLayout3 dl = delta_layout3((size_t)P, two);
        lo_a = 4 * (long)dl.lo;
        lo_b = 4 * (long)al.lo;
        if (f16) amax = reinterpret_cast<const unsigned*>(delta + dl.scale);
        for (int l = 0; l < D; ++l) { d_h[l] = delta + dl.h[l]; x_h[l] = act + al.h[l]; }
        d_feat = delta + dl.feat; d_hv = delta + dl.hv; d_rgb = delta + dl.graw;
        // feature 3 of the 4-wide tile: 3 rows of 32 two-byte elements
        d_sigma = reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(delta + dl.graw) + 3 * 32);
        x_feat = act + al.feat; x_hv = act + al.hv; x_enc = act + al.enc; x_dir = act + al.dir_pt;
        ld_graw = 4;
        if (phases & 1) {   // act is written by the forward; the expanded copy is scratch inside the same buffer
            if (S % 32 == 0) {
                const dim3 grid((unsigned)(((long)n_rays * 32 + 255) / 256));
                u32x4* dst = reinterpret_cast<u32x4*>(const_cast<float*>(act) + al.dir_pt);
                if (f16) hipLaunchKernelGGL(replicate_dir_bf16_kernel<SplitF16>, grid, dim3(256), 0, stream, act + al.dir, dst, (long)n_rays * 32);
                else hipLaunchKernelGGL(replicate_dir_bf16_kernel<SplitBF16>, grid, dim3(256), 0, stream, act + al.dir, dst, (long)n_rays * 32);
                if (two) hipLaunchKernelGGL((replicate_dir_bf16_kernel<SplitF16, true>), grid, dim3(256), 0, stream, act + al.dir,
                                            reinterpret_cast<u32x4*>(const_cast<float*>(act) + al.lo + al.dir_pt), (long)n_rays * 32);
            } else {
                const long n_thr = ((P + 31) >> 5) * 128;
                const dim3 grid((unsigned)((n_thr + 255) / 256));
                u32x4* dst = reinterpret_cast<u32x4*>(const_cast<float*>(act) + al.dir_pt);
                if (f16) hipLaunchKernelGGL(expand_dir_tiles_bf16_kernel<SplitF16>, grid, dim3(256), 0, stream, act + al.dir, dst, P, S);
                else hipLaunchKernelGGL(expand_dir_tiles_bf16_kernel<SplitBF16>, grid, dim3(256), 0, stream, act + al.dir, dst, P, S);
                if (two) hipLaunchKernelGGL((expand_dir_tiles_bf16_kernel<SplitF16, true>), grid, dim3(256), 0, stream, act + al.dir,
                                            reinterpret_cast<u32x4*>(const_cast<float*>(act) + al.lo + al.dir_pt), P, S);
            }
            e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
    } else {
        const ActLayout al = act_layout((size_t)P, (size_t)n_rays);
        const DeltaLayout dl = delta_layout((size_t)P);
        for (int l = 0; l < D; ++l) { d_h[l] = delta + dl.h[l]; x_h[l] = act + al.h[l]; }
        d_feat = delta + dl.feat; d_hv = delta + dl.hv; d_rgb = d_raw; d_sigma = d_raw + 3;
        x_feat = act + al.feat; x_hv = act + al.hv; x_enc = act + al.enc; x_dir = act + al.dir_pt;
        ld_graw = 4;
        if (phases & 1) {
            hipLaunchKernelGGL(expand_dir_kernel, dim3((unsigned)((P * 8 + 255) / 256)), dim3(256), 0, stream,
                               reinterpret_cast<const f32x4*>(act + al.dir), reinterpret_cast<f32x4*>(const_cast<float*>(act) + al.dir_pt), P, S);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
    }
    WgradArgs wa{};
    int nj = 0, tiles = 0;
    auto add = [&](const float* A, int lda, int nA, const float* B, int ldb, int nB, int rowdiv, int c_off, int ldc, int bias_off) {
        if (nj >= WG_MAX_JOBS) { ++nj; return; }
        WgradJob& j = wa.job[nj++];
        j.A = A; j.B = B; j.lda = lda; j.nA = nA; j.ldb = ldb; j.nB = nB; j.b_rowdiv = rowdiv;
        // rows saved by the 16-point forward (256- / 128-wide regions) are in 16-point tiles; encodings are not
        j.b_tile16 = (split16 && (ldb == W || ldb == WV)) ? 1 : 0;
        j.b_ray_tiles = 0;
        j.c_off = c_off; j.ldc = ldc; j.bias_off = bias_off;
        const int tn = (nA + WG_TILE - 1) / WG_TILE;
        j.tiles_k = (nB + WG_TILE - 1) / WG_TILE;
        j.tile_base = tiles;
        tiles += tn * j.tiles_k;
        j.vecA = ((reinterpret_cast<uintptr_t>(A) & 15) == 0 && lda % 4 == 0) ? 1 : 0;
        j.vecB = ((reinterpret_cast<uintptr_t>(B) & 15) == 0 && ldb % 4 == 0) ? 1 : 0;
    };
    add(d_h[0], W, W, x_enc, 64, IN_XYZ, 1, cn.w[0], IN_XYZ, cn.b[0]);
    for (int l = 1; l < D; ++l) {
        if (l == SKIP + 1) {
            add(d_h[l], W, W, x_enc, 64, IN_XYZ, 1, cn.w[l], W + IN_XYZ, cn.b[l]);
            add(d_h[l], W, W, x_h[l - 1], W, W, 1, cn.w[l] + IN_XYZ, W + IN_XYZ, -1);
        } else {
            add(d_h[l], W, W, x_h[l - 1], W, W, 1, cn.w[l], W, cn.b[l]);
        }
    }
    if (!fold) add(d_feat, W, W, x_h[D - 1], W, W, 1, cn.wf, W, cn.bf);
    // (the merged job addresses d_sigma through a 32-bit lane offset from d_hv: kept below 2 GiB -- launches beyond ~9 M points keep
    // the alpha head as a job of its own)
    // ... and the kernel's wrap-around arithmetic (a2_off + tile * (a2_tile_bytes - tile_bytes)) wants the graw region BEHIND hv, as
    // delta_layout3 lays them out: a reordered layout falls back to 13 jobs instead of reading the wrong rows
    // (NERF_WG_MERGE_ALPHA=0 in the environment, read per call: the 13-job plan, for the test that pins the merged row against it)
    const char* merge_env = getenv("NERF_WG_MERGE_ALPHA");
    const bool merge_alpha = split16 && NERF_WG_MERGE_ALPHA && !(merge_env && merge_env[0] == '0') && d_sigma > d_hv &&
                             reinterpret_cast<const char*>(d_sigma) - reinterpret_cast<const char*>(d_hv) < (1L << 31);
    if (!merge_alpha) add(d_sigma, ld_graw, 1, x_h[D - 1], W, W, 1, cn.wa, W, cn.ba);           // alpha_linear (A = d_sigma, one row)
    // fold: G = delta_hv^T h7 lands in the slot of Wv[:, :256]; wgrad_fold_kernel turns it into dWv[:, :256], dWf, dbf
    if (fold) {
        add(d_hv, WV, WV, x_h[D - 1], W, W, 1, cn.wv, W + IN_DIR, cn.bv);
        if (merge_alpha) {
            // the alpha head's one row rides on the view layer's job (both contract against h7: 512 B per point read once instead of
            // twice): row 128 of A = d_sigma = feature 3 of the 4-wide graw tiles of the same delta buffer
            WgradJob& j = wa.job[nj - 1];
            j.nA = WV + 1;
            j.a2_row = WV;
            j.a2_off = (unsigned)((reinterpret_cast<const char*>(d_sigma) - reinterpret_cast<const char*>(d_hv)));
            j.a2_tile_bytes = 64 * ld_graw;
            j.c_off2 = cn.wa;
            j.bias_off2 = cn.ba;
        }
    } else add(d_hv, WV, WV, x_feat, W, W, 1, cn.wv, W + IN_DIR, cn.bv);
    add(d_hv, WV, WV, x_dir, 32, IN_DIR, 1, cn.wv + W, W + IN_DIR, -1);
    if (split16 && S % 32 == 0 && nj <= WG_MAX_JOBS) wa.job[nj - 1].b_ray_tiles = S / 32;       // direction encoding: one record per ray
    add(d_rgb, ld_graw, 3, x_hv, WV, WV, 1, cn.wr, WV, cn.br);
    if (nj != WG_MAX_JOBS - (fold ? 1 : 0) - (merge_alpha ? 1 : 0)) return hipErrorInvalidValue;
    // full-width jobs -> wgrad256_kernel (whole 256x256 output per workgroup); the rest -> 128x128 tiles
    WgradArgs big{}, small{};
    int small_tiles = 0;
    for (int j = 0; j < nj; ++j) {
        const WgradJob& src = wa.job[j];
        if (split16 || (src.nA == 256 && src.nB == 256 && src.vecA && src.vecB && src.b_rowdiv == 1)) {
            big.job[big.n_jobs++] = src;
        } else {
            WgradJob& dst = small.job[small.n_jobs++];
            dst = src;
            dst.tile_base = small_tiles;
            small_tiles += ((src.nA + WG_TILE - 1) / WG_TILE) * src.tiles_k;
        }
    }
    int chunk_pts = 0;
    const int n_chunks = wgrad_chunks(P, &chunk_pts, nj);
    big.P = small.P = P;
    big.chunk_pts = small.chunk_pts = chunk_pts;
    big.n_chunks = small.n_chunks = n_chunks;
    big.partial = small.partial = partial;
    big.a_lo_bytes = lo_a;
    big.b_lo_bytes = lo_b;
    small.total_tiles = small_tiles;
    static bool attr_set = false;
    if (!attr_set) {
        e = hipFuncSetAttribute((const void*)wgrad256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WG256_LDS_FLOATS * 4);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    static bool attr1_set = false;
    if (split16 && !attr1_set) {
        e = hipFuncSetAttribute((const void*)wgrad1_kernel<SplitBF16>, hipFuncAttributeMaxDynamicSharedMemorySize, WG1_LDS_BYTES);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)wgrad1_kernel<SplitF16>, hipFuncAttributeMaxDynamicSharedMemorySize, WG1_LDS_BYTES);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)wgrad1_kernel<SplitF16, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, WG1_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr1_set = true;
    }
    if (big.n_jobs > 0 && split16 && (phases & 1)) {
        if (two) hipLaunchKernelGGL((wgrad1_kernel<SplitF16, 3>), dim3((unsigned)(big.n_jobs * n_chunks)), dim3(512), WG1_LDS_BYTES, stream, big);
        else if (f16) hipLaunchKernelGGL(wgrad1_kernel<SplitF16>, dim3((unsigned)(big.n_jobs * n_chunks)), dim3(512), WG1_LDS_BYTES, stream, big);
        else hipLaunchKernelGGL(wgrad1_kernel<SplitBF16>, dim3((unsigned)(big.n_jobs * n_chunks)), dim3(512), WG1_LDS_BYTES, stream, big);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    } else if (big.n_jobs > 0 && (phases & 1)) {
        hipLaunchKernelGGL(wgrad256_kernel, dim3((unsigned)(big.n_jobs * n_chunks)), dim3(512), WG256_LDS_FLOATS * 4, stream, big);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (small.n_jobs > 0 && (phases & 2)) {
        hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)(small_tiles * n_chunks)), dim3(256), 0, stream, small);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (phases & 4) {
        float* scratch = partial + wgrad_partial_floats(P) - N_DERIVED;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((N_PARAMS + 255) / 256), dim3(256), 0, stream,
                           (const float*)partial, n_chunks, grad, accumulate, fold ? 1 : 0, scratch, amax);
        if (fold)
            hipLaunchKernelGGL(wgrad_fold_kernel, dim3((WV * W + W * W + W + 255) / 256), dim3(256), 0, stream,
                               params, (const float*)scratch, grad, accumulate);
    }
    return hipGetLastError();
}

hipError_t launch_field_bwd(const float* packed, const float* act, const float* d_raw, int n_rays, int S,
                            float* delta, float* partial, float* grad, int accumulate, hipStream_t stream) {
    hipError_t e = launch_field_dgrad(packed, act, d_raw, n_rays, S, delta, stream);
    if (e != hipSuccess) return e;
    return launch_field_wgrad(act, delta, d_raw, n_rays, S, partial, grad, accumulate, 0, 7, stream, nullptr);
}

}  // namespace nerf
