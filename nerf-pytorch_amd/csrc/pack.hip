// Repack canonical (state_dict-order) parameters into the MFMA A-fragment
// streams consumed by the field kernels (layout: nerf_common.h).
// Replaces nothing in the reference: it is the price of keeping nn.Module
// parameters in PyTorch layout (run_nerf_helpers.py:79-94) while the kernels
// read lane-linear fragments.  ~1.2 M elements, one launch per optimizer step.
#include <hip/hip_runtime.h>
#include "nerf_common.h"
#include "split_types.h"
#include "field_ring8.h"

#include "launchers.h"

namespace nerf {

// canonical source index of packed element idx, or -1 for zero padding
__host__ __device__ inline int pack_source(int idx) {
    constexpr Canon c = canon();
    if (idx < FWD_END) {
        if (idx >= FWD_VIEWS) {                                   // NB = 8
            const int r = idx - FWD_VIEWS;
            const int s = r / KSTEP_F8, rem = r % KSTEP_F8;
            const int g = rem >> 8, lane = (rem & 255) >> 2, j = rem & 3;
            const int n = 16 * (4 * g + j) + (lane & 15), q = lane >> 4;
            int col;
            if (s < KS_H) col = hcol(s, q);
            else { const int d = dirslot(s - KS_H, q); if (d < 0) return -1; col = W + d; }
            return c.wv + n * (W + IN_DIR) + col;
        }
        int base, layer;                                          // NB = 16 regions
        if (idx < FWD_L1) { base = FWD_L0; layer = 0; }
        else if (idx < FWD_L5) { layer = 1 + (idx - FWD_L1) / (KS_H * KSTEP_F16); base = FWD_L1 + (layer - 1) * KS_H * KSTEP_F16; }
        else if (idx < FWD_L6) { base = FWD_L5; layer = 5; }
        else if (idx < FWD_FEAT) { layer = 6 + (idx - FWD_L6) / (KS_H * KSTEP_F16); base = FWD_L6 + (layer - 6) * KS_H * KSTEP_F16; }
        else { base = FWD_FEAT; layer = 8; }
        const int r = idx - base;
        const int s = r / KSTEP_F16, rem = r % KSTEP_F16;
        const int g = rem >> 8, lane = (rem & 255) >> 2, j = rem & 3;
        const int n = 16 * (4 * g + j) + (lane & 15), q = lane >> 4;
        if (layer == 8) return c.wf + n * W + hcol(s, q);
        if (layer == 0) { const int e = encslot(s, q); return e < 0 ? -1 : c.w[0] + n * IN_XYZ + e; }
        if (layer == SKIP + 1) {
            if (s < KS_ENC) { const int e = encslot(s, q); return e < 0 ? -1 : c.w[layer] + n * (W + IN_XYZ) + e; }
            return c.w[layer] + n * (W + IN_XYZ) + IN_XYZ + hcol(s - KS_ENC, q);
        }
        return c.w[layer] + n * W + hcol(s, q);
    }
    if (idx < BWD_END) {                                          // transposed streams, NB = 16
        int r, kind;                                              // kind: -2 views, -1 feat, else layer
        if (idx < BWD_FEAT) { r = idx - BWD_VIEWS; kind = -2; }
        else if (idx < BWD_L7) { r = idx - BWD_FEAT; kind = -1; }
        else { const int t = (idx - BWD_L7) / (KS_H * KSTEP_F16); kind = 7 - t; r = idx - BWD_L7 - t * KS_H * KSTEP_F16; }
        const int s = r / KSTEP_F16, rem = r % KSTEP_F16;
        const int g = rem >> 8, lane = (rem & 255) >> 2, j = rem & 3;
        const int k = 16 * (4 * g + j) + (lane & 15), q = lane >> 4;   // output row = input feature of the layer
        const int n = hcol(s, q);                                       // contraction slot = output feature
        if (kind == -2) return c.wv + n * (W + IN_DIR) + k;
        if (kind == -1) return c.wf + n * W + k;
        if (kind == SKIP + 1) return c.w[kind] + n * (W + IN_XYZ) + IN_XYZ + k;
        return c.w[kind] + n * W + k;
    }
    if (idx < SM_BFEAT) { const int l = (idx - SM_BIAS) / W; return c.b[l] + (idx - SM_BIAS) % W; }
    if (idx < SM_BVIEWS) return c.bf + (idx - SM_BFEAT);
    if (idx < SM_WALPHA) return c.bv + (idx - SM_BVIEWS);
    if (idx < SM_WRGB) return c.wa + (idx - SM_WALPHA);
    if (idx < SM_BALPHA) return c.wr + (idx - SM_WRGB);
    if (idx < SM_BRGB) return idx == SM_BALPHA ? c.ba : -1;
    return (idx - SM_BRGB) < 3 ? c.br + (idx - SM_BRGB) : -1;
}

__global__ void pack_params_kernel(const float* __restrict__ canon_params, float* __restrict__ packed) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= PACKED_FLOATS) return;
    const int src = pack_source(idx);
    packed[idx] = src < 0 ? 0.0f : canon_params[src];
}

// ---- three-term-split repack: every weight becomes (hi, lo) 16-bit parts; layout in nerf_common.h
// returns the canonical source index of 16-bit element e16 of the TRANSPOSED streams of the delta chain (-1: zero padding);
// *is_lo tells whether the element is the low part.  Output row = input feature k of the layer, contraction slot = output feature n.
__host__ __device__ inline int pack3_source(int e16, int* is_lo) {
    constexpr Canon c = canon();
    const int word = e16 >> 1;
    int base, kind;          // kind: 0 = W'^T (folded view matrix), 1 = feature_linear^T, 2 + t: layer (7 - t)^T
    if (word < P3B_FEAT) { base = P3B_VIEWS; kind = 0; }
    else if (word < P3B_L7) { base = P3B_FEAT; kind = 1; }
    else { const int t = (word - P3B_L7) / (KS3_H * KSTEP3_W8); kind = 2 + t; base = P3B_L7 + t * KS3_H * KSTEP3_W8; }
    const int per_kstep16 = 8 * 2 * 64 * 8;
    const int r = e16 - 2 * base;
    const int s = r / per_kstep16, rem = r % per_kstep16;
    const int nb = rem / 1024, hl = (rem / 512) & 1, lane = (rem >> 3) & 63, j = rem & 7;
    const int row = 32 * nb + (lane & 31), half = lane >> 5;
    *is_lo = hl;
    const int n = h3slot(s, half, j);
    if (kind == 0) return DERIVED_WVF + n * W + row;        // W'^T
    if (kind == 1) return c.wf + n * W + row;               // (feature_linear^T: packed, skipped by the kernel)
    const int l = 7 - (kind - 2);
    if (l == SKIP + 1) return c.w[l] + n * (W + IN_XYZ) + IN_XYZ + row;
    return c.w[l] + n * W + row;
}

__device__ inline unsigned short bf16_rne(float x) {
    return __builtin_bit_cast(unsigned short, (__bf16)x);
}

// W' = Wv[:, :256] Wf and b' = Wv[:, :256] bf + bv (nerf_common.h, folded feature layer): fp64 accumulation.  Eight lanes
// share one output (contraction index i = part, part + 8, ...; fp64 shuffle reduction): 263 k threads with 32 dependent
// loads each instead of 33 k threads with 256 (the kernel runs after every optimizer step: 21 -> ~4 us).
// blockIdx.y selects the network: a training step repacks the coarse and the fine network in the same two launches (PackPair)
struct PackPair { const float* params[2]; float* packed[2]; };
__global__ void derive_folded_kernel(PackPair pp) {
    constexpr Canon c = canon();
    const float* __restrict__ p = pp.params[blockIdx.y];
    float* __restrict__ derived = pp.packed[blockIdx.y] + P3_DERIVED;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int part = tid & 7;
    const int idx = min(tid >> 3, N_DERIVED - 1);        // (every lane takes part in the shuffles)
    double acc = 0.0;
    if (idx < WV * W) {
        const int k = idx / W, j = idx % W;
#pragma unroll 8
        for (int i = part; i < W; i += 8) acc += (double)p[c.wv + k * (W + IN_DIR) + i] * (double)p[c.wf + i * W + j];
    } else {
        const int k = idx - WV * W;
#pragma unroll 8
        for (int i = part; i < W; i += 8) acc += (double)p[c.wv + k * (W + IN_DIR) + i] * (double)p[c.bf + i];
        if (part == 0) acc += (double)p[c.bv + k];
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    acc += __shfl_xor(acc, 4);
    if (part == 0 && (tid >> 3) < N_DERIVED) derived[idx] = (float)acc;
}
__device__ inline float param_or_derived(const float* canon_params, const float* derived, int src) {
    return src < N_PARAMS ? canon_params[src] : derived[src - N_PARAMS];
}

// 16-point-per-wave forward stream (nerf_common.h, P16F): canonical source of 16-bit element e16 of the region
__host__ __device__ inline int pack16_source(int e16, int* is_lo) {
    constexpr Canon c = canon();
    const int word = e16 >> 1;
    int base, kind;          // kind: 0..7 trunk layer, 8 feature, 9 views
    if (word < P16F_L1) { base = 0; kind = 0; }
    else if (word < P16F_L5) { kind = 1 + (word - P16F_L1) / (KS16_H * KSTEP16_W16); base = P16F_L1 + (kind - 1) * KS16_H * KSTEP16_W16; }
    else if (word < P16F_L6) { base = P16F_L5; kind = 5; }
    else if (word < P16F_FEAT) { kind = 6 + (word - P16F_L6) / (KS16_H * KSTEP16_W16); base = P16F_L6 + (kind - 6) * KS16_H * KSTEP16_W16; }
    else if (word < P16F_VIEWS) { base = P16F_FEAT; kind = 8; }
    else { base = P16F_VIEWS; kind = 9; }
    const int nblk = kind == 9 ? 8 : 16;
    const int per_kstep16 = nblk * 2 * 64 * 8;
    const int r = e16 - 2 * base;
    const int s = r / per_kstep16, rem = r % per_kstep16;
    const int nb = rem / 1024, hl = (rem / 512) & 1, lane = (rem >> 3) & 63, j = rem & 7;
    const int row = 16 * nb + (lane & 15), kq = lane >> 4;
    *is_lo = hl;
    if (kind <= 7) {
        const int ld = fan_in(kind);
        if (kind == 0) { const int e = encslot(8 * s + j, kq); return e < 0 ? -1 : c.w[0] + row * ld + e; }
        if (kind == SKIP + 1) {
            if (s < KS16_ENC) { const int e = encslot(8 * s + j, kq); return e < 0 ? -1 : c.w[kind] + row * ld + e; }
            return c.w[kind] + row * ld + IN_XYZ + hcol(8 * (s - KS16_ENC) + j, kq);
        }
        return c.w[kind] + row * ld + hcol(8 * s + j, kq);
    }
    if (kind == 8) return c.wf + row * W + hcol(8 * s + j, kq);
    if (s < KS16_H) return DERIVED_WVF + row * W + hcol(8 * s + j, kq);      // folded W'

    const int d = j < KS_DIR ? dirslot(j, kq) : -1;
    return d < 0 ? -1 : c.wv + row * (W + IN_DIR) + W + d;
}

void pack16_table_host(int* out) {
    for (int e = 0; e < 2 * P16F_WORDS; ++e) { int lo; const int s = pack16_source(e, &lo); out[e] = s < 0 ? -1 : 2 * s + lo; }
}

void pack3_table_host(int* out) {
    for (int e = 0; e < 2 * P3B_END; ++e) { int lo; const int s = pack3_source(e, &lo); out[e] = s < 0 ? -1 : 2 * s + lo; }
}

// ONE repack launch for everything a split-datapath step reads (after derive_folded, which it depends on).  `streams` selects
// the fragment streams to write: bit 0 = 16-point forward (P16F), bit 2 = transposed (hi, lo) streams of the delta chain (P3B)
// (bits 1 and 3 named streams of kernels that no longer exist).  The small fp32 parameters are always written.
// SP (split_types.h): the 16-bit type the weights are split into -- the same buffer layout either way.
template <typename SP>
__global__ void pack3_all_kernel(PackPair pp, int n16f, int n3b) {
    const float* __restrict__ canon_params = pp.params[blockIdx.y];
    float* __restrict__ packed = pp.packed[blockIdx.y];
    const float* __restrict__ derived = packed + P3_DERIVED;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned short* p16 = reinterpret_cast<unsigned short*>(packed);
    auto emit = [&](unsigned short* dst, int src, int is_lo) {
        unsigned short v = 0;
        if (src >= 0) {
            const float x = param_or_derived(canon_params, derived, src);
            v = is_lo ? split_lo<SP>(x) : split_hi<SP>(x);
        }
        *dst = v;
    };
    if (idx < n16f) {                                        // 16-point forward stream
        int lo;
        const int src = pack16_source((int)idx, &lo);
        emit(p16 + 2L * P16F + idx, src, lo);
        return;
    }
    idx -= n16f;
    if (idx < n3b) {                                         // transposed streams
        int lo;
        const int src = pack3_source((int)idx, &lo);
        emit(p16 + 2L * P3B_VIEWS + idx, src, lo);
        return;
    }
    idx -= n3b;
    if (idx < PACKED_FLOATS - SM_BIAS) {                     // small fp32 parameters (b' replaces the view-branch bias)
        const int i = (int)idx, pidx = SM_BIAS + i;
        if (pidx >= SM_BVIEWS && pidx < SM_WALPHA) { packed[P3_SMALL + i] = derived[WV * W + (pidx - SM_BVIEWS)]; return; }
        const int src = pack_source(pidx);
        packed[P3_SMALL + i] = src < 0 ? 0.0f : canon_params[src];
    }
}

// ---- reduced inference stream ("fp16 main + fp8 corrections", field_ring8.h): the P16F region with the units of the 256-wide
// contractions (layers 1..7 over their hidden inputs, the trunk part of the view branch) re-filled KIND-major per 128 contraction
// slots; everything else in the region keeps the three-term fp16 content.
// matrix m = 0..6: layer m + 1 (hidden-input columns), m = 7: the folded view matrix W' (derived)
struct RedSeg { int unit0, n_units, ng, m; };
__host__ __device__ inline bool red_segment(int unit, RedSeg* seg) {
    // units of the P16F region: L0 [0, 8) | L1..L4 [8, 136) | L5 enc [136, 144), L5 hidden [144, 176) | L6 L7 [176, 240) | FEAT [240, 272) |
    // VIEWS trunk [272, 288), direction [288, 290)
    if (unit >= 8 && unit < 136) { const int l = 1 + (unit - 8) / 32; *seg = RedSeg{8 + 32 * (l - 1), 32, 4, l - 1}; return true; }
    if (unit >= 144 && unit < 176) { *seg = RedSeg{144, 32, 4, 4}; return true; }
    if (unit >= 176 && unit < 240) { const int l = 6 + (unit - 176) / 32; *seg = RedSeg{176 + 32 * (l - 6), 32, 4, l - 1}; return true; }
    if (unit >= 272 && unit < 288) { *seg = RedSeg{272, 16, 2, 7}; return true; }
    return false;
}
constexpr int RED_SCALE_WORD = P16F + P16F_FEAT;         // 4 words = 16 scale bytes, parked in the (skipped) feature_linear units
// canonical (or derived) index of W_m[row][feature]
__host__ __device__ inline int red_source(int m, int row, int feature) {
    constexpr Canon c = canon();
    if (m == 7) return DERIVED_WVF + row * W + feature;
    const int l = m + 1;
    return l == SKIP + 1 ? c.w[l] + row * (W + IN_XYZ) + IN_XYZ + feature : c.w[l] + row * W + feature;
}
__device__ inline float f16_hi(float x) { return (float)(_Float16)x; }
// largest |hi16| and |x - hi16| of every reduced matrix -> E8M0 scale bytes (what the MFMA multiplies the fp8 values by):
// byte = 127 - k, fp8 value = part * 2^k with k = 7 - floor(log2(max)): the maximum lands in [128, 256)
__global__ __launch_bounds__(1024) void weight_scale_kernel(const float* __restrict__ canon_params, const float* __restrict__ derived,
                                                             float* __restrict__ packed) {
    const int m = blockIdx.x;
    const int rows = m == 7 ? WV : W;
    float mh = 0.0f, ml = 0.0f;
    for (int i = threadIdx.x; i < rows * W; i += 1024) {
        const float x = param_or_derived(canon_params, derived, red_source(m, i / W, i % W));
        const float h = f16_hi(x);
        mh = fmaxf(mh, fabsf(h));
        ml = fmaxf(ml, fabsf(x - h));
    }
    __shared__ float sh[16], sl[16];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { mh = fmaxf(mh, __shfl_xor(mh, o)); ml = fmaxf(ml, __shfl_xor(ml, o)); }
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = mh; sl[threadIdx.x >> 6] = ml; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) { mh = fmaxf(mh, sh[i]); ml = fmaxf(ml, sl[i]); }
        auto scale_byte = [](float mx) {
            const int e = (int)((__float_as_uint(mx) >> 23) & 0xffu);          // biased exponent; 0 = zero / subnormal maximum
            const int k = e == 0 ? 0 : 7 - (e - 127);
            return (unsigned char)max(1, min(254, 127 - k));
        };
        unsigned char* dst = reinterpret_cast<unsigned char*>(packed + RED_SCALE_WORD);
        dst[2 * m] = scale_byte(mh);
        dst[2 * m + 1] = scale_byte(ml);
    }
}
// 16-bit element e16 of the P16F region of a REDUCED buffer: true if it belongs to a reduced unit (*out = its value)
__device__ inline bool pack8_element(int e16, const float* canon_params, const float* derived, const float* packed, unsigned short* out) {
    const int unit = e16 / 4096, r = e16 % 4096;
    RedSeg sg;
    if (!red_segment(unit, &sg)) return false;
    const int v = unit - sg.unit0;
    const int T = v / (4 * sg.ng), k = (v % (4 * sg.ng)) / sg.ng, g = v % sg.ng;
    const int f = r / 512, lane = (r >> 3) & 63, j = r & 7;
    const int kq = lane >> 4;
    if (k < 2) {                        // main: frag f = 2 i + k-step of the pair, 16-bit element j of the k-step's eight
        const int i = f >> 1, ksl = f & 1;
        const int row = 16 * (4 * g + i) + (lane & 15);
        const int vi = 8 * (4 * T + 2 * k + ksl) + j;
        *out = split_hi<SplitF16>(param_or_derived(canon_params, derived, red_source(sg.m, row, hcol(vi, kq))));
        return true;
    }
    // fp8: frag f = 2 jj + half; block = 2 half + (jj >> 1); 16-byte part jj & 1; this 16-bit element = bytes 2 j, 2 j + 1 of the part
    const int half = f & 1, jj = f >> 1;
    const int row = 16 * (4 * g + 2 * half + (jj >> 1)) + (lane & 15);
    const int b0 = 16 * (jj & 1) + 2 * j;
    const unsigned char sbyte = reinterpret_cast<const unsigned char*>(packed + RED_SCALE_WORD)[2 * sg.m + (k - 2)];
    const float inv_scale = __uint_as_float((unsigned)sbyte << 23);          // 2^(byte - 127) = 2^-k: value / inv_scale = value * 2^k
    float part[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float x = param_or_derived(canon_params, derived, red_source(sg.m, row, hcol(32 * T + b0 + t, kq)));
        const float h = f16_hi(x);
        part[t] = k == 2 ? h : x - h;
    }
    typedef short i16x2 __attribute__((ext_vector_type(2)));
    const i16x2 pk = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(i16x2{0, 0}, part[0], part[1], inv_scale, false);
    *out = (unsigned short)pk[0];
    return true;
}
// the reduced units of the P16F region + the scale bytes into the small-parameter pads the forward stages into LDS
__global__ void pack8_kernel(const float* __restrict__ canon_params, const float* __restrict__ derived, float* __restrict__ packed) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < 2L * P16F_WORDS) {
        unsigned short v;
        if (pack8_element((int)idx, canon_params, derived, packed, &v)) reinterpret_cast<unsigned short*>(packed)[2L * P16F + idx] = v;
        return;
    }
    if (idx == 2L * P16F_WORDS) {       // 12 + 4 scale bytes: pads behind alpha_linear.bias and rgb_linear.bias (nerf_common.h SM_*)
        const unsigned* src = reinterpret_cast<const unsigned*>(packed + RED_SCALE_WORD);
        unsigned* small = reinterpret_cast<unsigned*>(packed + P3_SMALL);
        small[SM_BALPHA - SM_BIAS + 1] = src[0];
        small[SM_BALPHA - SM_BIAS + 2] = src[1];
        small[SM_BALPHA - SM_BIAS + 3] = src[2];
        small[SM_BRGB - SM_BIAS + 3] = src[3];
    }
}

// one or two networks (params_b == nullptr: one) in the same two launches: derive_folded, then every selected stream
hipError_t launch_pack3_pair(const float* params_a, float* packed_a, const float* params_b, float* packed_b, int streams, hipStream_t stream, int split) {
    const int threads = 256;
    const unsigned nets = params_b ? 2 : 1;
    PackPair pp{{params_a, params_b ? params_b : params_a}, {packed_a, params_b ? packed_b : packed_a}};
    hipLaunchKernelGGL(derive_folded_kernel, dim3((8 * N_DERIVED + threads - 1) / threads, nets), dim3(threads), 0, stream, pp);
    const int n16f = (streams & 1) ? 2 * P16F_WORDS : 0, n3b = (streams & 4) ? 2 * (P3B_END - P3B_VIEWS) : 0;
    const long total = (long)n16f + n3b + (PACKED_FLOATS - SM_BIAS);
    const dim3 grid((unsigned)((total + threads - 1) / threads), nets);
    if (split) hipLaunchKernelGGL(pack3_all_kernel<SplitF16>, grid, dim3(threads), 0, stream, pp, n16f, n3b);
    else hipLaunchKernelGGL(pack3_all_kernel<SplitBF16>, grid, dim3(threads), 0, stream, pp, n16f, n3b);
    return hipGetLastError();
}

hipError_t launch_pack3_sel(const float* canon_params, float* packed, int streams, hipStream_t stream, int split) {
    hipError_t e = launch_pack3_pair(canon_params, packed, nullptr, nullptr, streams, stream, split);
    if (e != hipSuccess) return e;
    if (split == 2) {       // reduced inference stream on top of the fp16 three-term stream (its narrow units and small parameters stay)
        const int threads = 256;
        const float* derived = packed + P3_DERIVED;
        hipLaunchKernelGGL(weight_scale_kernel, dim3(N_RED_MATRICES), dim3(1024), 0, stream, canon_params, derived, packed);
        const long n8 = 2L * P16F_WORDS + 1;
        hipLaunchKernelGGL(pack8_kernel, dim3((unsigned)((n8 + threads - 1) / threads)), dim3(threads), 0, stream, canon_params, derived, packed);
    }
    return hipGetLastError();
}

// host copy of the gather table (CPU tests emulate the MFMA data flow with it)
void pack_table_host(int* out) {
    for (int i = 0; i < PACKED_FLOATS; ++i) out[i] = pack_source(i);
}

hipError_t launch_pack(const float* canon_params, float* packed, hipStream_t stream) {
    const int threads = 256;
    const int blocks = (PACKED_FLOATS + threads - 1) / threads;
    hipLaunchKernelGGL(pack_params_kernel, dim3(blocks), dim3(threads), 0, stream, canon_params, packed);
    return hipGetLastError();
}

}  // namespace nerf
