// Per-ray device code of the volumetric renderer, one wavefront per ray (see ray_ops.hip for the launchable kernels and
// render_fused.hip for the one-launch inference kernel that calls the same functions):
//   coarse_depth        run_nerf.py:357-379   (z_vals: linspace / lindisp / stratified jitter)
//   composite_ray       run_nerf.py:262-305   (raw2outputs) and its adjoint
//   sample_fine_ray     run_nerf.py:392-396,412 + run_nerf_helpers.py:196-239 (sample_pdf, sort, z_std)
// Sync: how the lanes of the ray's wavefront order their LDS traffic -- BlockSync when the wavefront is the whole workgroup
// (__syncthreads), WaveSync when other wavefronts of the workgroup work on other rays (LDS operations of one wavefront
// execute in order; only the compiler has to be held to the program order).
#pragma once
#include <hip/hip_runtime.h>
#include "nerf_common.h"
#include "launchers.h"

namespace nerf {

struct BlockSync { __device__ static __forceinline__ void sync() { __syncthreads(); } };
struct WaveSync {
    __device__ static __forceinline__ void sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};

// torch.linspace(0, 1, n)[i] as ATen evaluates it: start + step * i for the first half, end - step * (n - 1 - i) for the second,
// each ONE fused multiply-add (checked against torch for n = 37 .. 192, tests/test_gpu_round3.py)
__device__ __forceinline__ float linspace01_at(int i, int n) {
    if (n == 1) return 0.0f;
    const float step = (1.0f - 0.0f) / (float)(n - 1);
    return i < n / 2 ? fmaf(step, (float)i, 0.0f) : fmaf(-step, (float)(n - 1 - i), 1.0f);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ wave scans
__device__ inline float wave_incl_scan_mul(float v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const float t = __shfl_up(v, d); if (lane >= d) v *= t; }
    return v;
}
__device__ inline float wave_incl_scan_add(float v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const float t = __shfl_up(v, d); if (lane >= d) v += t; }
    return v;
}
// inclusive suffix sum (lane i gets sum over lanes >= i): direct, no total-minus-prefix cancellation
__device__ inline float wave_incl_scan_add_rev(float v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const float t = __shfl_down(v, d); if (lane + d < 64) v += t; }
    return v;
}
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// ------------------------------------------------------------------ coarse depths
__device__ inline float z_at(float near, float far, float t, int lindisp) {
    if (!lindisp) return near * (1.0f - t) + far * t;
    return 1.0f / (1.0f / near * (1.0f - t) + 1.0f / far * t);
}

// depth j of a ray (near, far): t(j) = t_vals[j] (or linspace01_at when t_vals is null); jittered inside its stratum when
// t_rand points at the ray's draws
template <typename T>
__device__ __forceinline__ float coarse_depth(float near, float far, T tv, int j, int S, int lindisp, const float* t_rand_row) {
    const float zj = z_at(near, far, tv(j), lindisp);
    if (!t_rand_row) return zj;
    const float z_last = z_at(near, far, tv(S - 1), lindisp);
    const float z_first = z_at(near, far, tv(0), lindisp);
    float upper, lower;
    if (j < S - 1) upper = 0.5f * (z_at(near, far, tv(j + 1), lindisp) + zj); else upper = z_last;
    if (j > 0) lower = 0.5f * (zj + z_at(near, far, tv(j - 1), lindisp)); else lower = z_first;
    return lower + (upper - lower) * t_rand_row[j];
}

// ------------------------------------------------------------------ compositing
__device__ inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// LDS per ray: 2*S floats (fwd) / 4*S floats (bwd); each lane touches only its own segment: no synchronisation
template <bool BWD>
__device__ __forceinline__ void composite_ray(const CompositeArgs& a, int ray, int lane, float* sm) {
    const int S = a.S;
    const int C = (S + 63) >> 6;
    const int lo = lane * C, hi = min(lo + C, S);
    const float* raw = a.raw + (size_t)ray * S * 4;
    const float* z = a.z + (size_t)ray * S;
    const float* dp = a.dirs + (size_t)ray * a.dir_stride;
    const float dn = sqrtf(dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2]);
    float* s_alpha = sm;            // alpha_i
    float* s_t = sm + S;            // 1 - alpha_i + 1e-10
    float* s_e = sm + 2 * S;        // exp(-relu(sigma) * dist)        (bwd)
    float* s_gw = sm + 3 * S;       // g_i * w_i                       (bwd)

    // pass 1: opacity per sample, segment product of (1 - alpha + 1e-10)
    float seg = 1.0f;
    for (int i = lo; i < hi; ++i) {
        float dist = (i + 1 < S) ? (z[i + 1] - z[i]) : 1e10f;
        dist = dist * dn;
        float sg = raw[4 * i + 3];
        if (a.noise) sg = sg + a.noise[(size_t)ray * S + i] * a.noise_std;
        const float ex = expf(-fmaxf(sg, 0.0f) * dist);
        const float al = 1.0f - ex;
        const float t = 1.0f - al + 1e-10f;
        s_alpha[i] = al;
        s_t[i] = t;
        if (BWD) s_e[i] = (sg > 0.0f) ? ex * dist : 0.0f;      // d alpha / d sigma
        seg *= t;
    }
    const float incl = wave_incl_scan_mul(seg, lane);
    float T = __shfl_up(incl, 1);
    if (lane == 0) T = 1.0f;

    // pass 2: weights and ray integrals
    float r0 = 0.f, r1 = 0.f, r2 = 0.f, dsum = 0.f, asum = 0.f;
    const float Tstart = T;
    for (int i = lo; i < hi; ++i) {
        const float w = s_alpha[i] * T;
        T *= s_t[i];
        const float c0 = sigmoidf_(raw[4 * i]), c1 = sigmoidf_(raw[4 * i + 1]), c2 = sigmoidf_(raw[4 * i + 2]);
        r0 += w * c0; r1 += w * c1; r2 += w * c2;
        dsum += w * z[i];
        asum += w;
        if (!BWD && a.weights) a.weights[(size_t)ray * S + i] = w;
    }
    r0 = wave_sum(r0); r1 = wave_sum(r1); r2 = wave_sum(r2);
    dsum = wave_sum(dsum); asum = wave_sum(asum);
    const float ratio = dsum / asum;
    // torch.max(1e-10, ratio) propagates NaN (0/0 on empty rays): reference quirk, kept
    const float m = (ratio != ratio) ? ratio : fmaxf(1e-10f, ratio);

    if (!BWD) {
        if (lane == 0) {
            float w0 = r0, w1 = r1, w2 = r2;
            if (a.white_bkgd) { const float bg = 1.0f - asum; w0 = w0 + bg; w1 = w1 + bg; w2 = w2 + bg; }
            a.rgb[(size_t)ray * 3] = w0; a.rgb[(size_t)ray * 3 + 1] = w1; a.rgb[(size_t)ray * 3 + 2] = w2;
            a.disp[ray] = 1.0f / m;
            a.acc[ray] = asum;
            if (a.depth) a.depth[ray] = dsum;
        }
        return;
    }

    // ---- backward: g_i = dL/dw_i, R_i = sum_{j>i} g_j w_j,
    //      dL/dalpha_i = g_i T_i - R_i / t_i
    const float g0 = a.d_rgb[(size_t)ray * 3], g1 = a.d_rgb[(size_t)ray * 3 + 1], g2 = a.d_rgb[(size_t)ray * 3 + 2];
    float gacc = a.d_acc ? a.d_acc[ray] : 0.0f;
    float gdepth = a.d_depth ? a.d_depth[ray] : 0.0f;
    const float* gw = a.d_weights ? a.d_weights + (size_t)ray * S : nullptr;
    if (a.white_bkgd) gacc -= (g0 + g1 + g2);
    if (a.d_disp) {
        const float gd = a.d_disp[ray];
        if (ratio > 1e-10f) {               // disp = acc / depth on this branch
            const float inv = -gd / (m * m);
            gdepth += inv / asum;
            gacc += inv * (-dsum / (asum * asum));
        }
    }
    T = Tstart;
    float segsum = 0.0f;
    for (int i = lo; i < hi; ++i) {
        const float w = s_alpha[i] * T;
        const float c0 = sigmoidf_(raw[4 * i]), c1 = sigmoidf_(raw[4 * i + 1]), c2 = sigmoidf_(raw[4 * i + 2]);
        float g = g0 * c0 + g1 * c1 + g2 * c2 + gacc + gdepth * z[i];
        if (gw) g += gw[i];
        s_gw[i] = g * w;
        segsum += g * w;
        // colour gradients are local
        float* dr = a.d_raw + ((size_t)ray * S + i) * 4;
        dr[0] = w * g0 * (c0 * (1.0f - c0));
        dr[1] = w * g1 * (c1 * (1.0f - c1));
        dr[2] = w * g2 * (c2 * (1.0f - c2));
        // stash g*T for the alpha gradient
        s_alpha[i] = g * T;
        T *= s_t[i];
    }
    // suffix sums over later lanes (reverse scan, like ATen's reversed cumsum in cumprod_backward)
    const float suf = wave_incl_scan_add_rev(segsum, lane);
    float R = __shfl_down(suf, 1);          // sum over lanes > this one
    if (lane == 63) R = 0.0f;
    for (int i = hi - 1; i >= lo; --i) {
        const float dalpha = s_alpha[i] - R / s_t[i];
        a.d_raw[((size_t)ray * S + i) * 4 + 3] = dalpha * s_e[i];
        R += s_gw[i];
    }
}

// ------------------------------------------------------------------ hierarchical sampling
// LDS: bins[nb] | cdf[nb] | vals[n_in + Nf]
template <typename Sync>
__device__ __forceinline__ void sample_fine_ray(const FineArgs& a, int ray, int lane, float* sm) {
    const int Nf = a.Nf;
    const int nb = a.direct ? a.n_in : a.n_in - 1;      // bins
    const int nw = nb - 1;                              // pdf entries
    const int Sc = a.direct ? 0 : a.n_in;
    const int tot = Sc + Nf;
    float* bins = sm;
    float* cdf = sm + nb;
    float* vals = sm + 2 * nb;
    const float* in0 = a.in0 + (size_t)ray * a.n_in;
    const float* w = a.direct ? a.in1 + (size_t)ray * nw : a.in1 + (size_t)ray * a.n_in + 1;   // weights[1:-1]

    if (a.direct) { for (int i = lane; i < nb; i += 64) bins[i] = in0[i]; }
    else {
        for (int i = lane; i < nb; i += 64) bins[i] = 0.5f * (in0[i + 1] + in0[i]);
        for (int i = lane; i < Sc; i += 64) vals[i] = in0[i];
    }
    float part = 0.0f;
    for (int i = lane; i < nw; i += 64) part += w[i] + 1e-5f;
    const float wsum = wave_sum(part);
    Sync::sync();
    // cdf = [0, cumsum(pdf)]: the ADDITIONS are sequential like torch.cumsum so rounding follows the reference; the divisions
    // (independent) are done by all lanes first
    for (int i = lane; i < nw; i += 64) cdf[i + 1] = (w[i] + 1e-5f) / wsum;
    Sync::sync();
    if (lane == 0) {
        float c = 0.0f;
        cdf[0] = 0.0f;
        for (int i = 0; i < nw; ++i) { c += cdf[i + 1]; cdf[i + 1] = c; }
    }
    Sync::sync();
    float s1 = 0.0f;
    for (int k = lane; k < Nf; k += 64) {
        const float u = a.u ? a.u[(size_t)ray * Nf + k] : (a.u_lin ? a.u_lin[k] : linspace01_at(k, Nf));
        // searchsorted(cdf, u, right=True) = number of entries <= u: the cdf is non-decreasing, so a binary search counts them
        int lo_ = 0, hi_ = nb;
        while (lo_ < hi_) { const int mid = (lo_ + hi_) >> 1; if (cdf[mid] <= u) lo_ = mid + 1; else hi_ = mid; }
        const int idx = lo_;
        const int below = max(idx - 1, 0), above = min(idx, nb - 1);
        const float c0 = cdf[below], c1 = cdf[above];
        float denom = c1 - c0;
        if (denom < 1e-5f) denom = 1.0f;
        const float t = (u - c0) / denom;
        const float smp = bins[below] + t * (bins[above] - bins[below]);
        vals[Sc + k] = smp;
        if (a.z_samples) a.z_samples[(size_t)ray * Nf + k] = smp;
        s1 += smp;
    }
    const float mean = wave_sum(s1) / (float)Nf;
    Sync::sync();
    if (a.z_std) {
        float s2 = 0.0f;
        for (int k = lane; k < Nf; k += 64) { const float d = vals[Sc + k] - mean; s2 += d * d; }
        s2 = wave_sum(s2);
        if (lane == 0) a.z_std[ray] = sqrtf(s2 / (float)Nf);
    }
    if (!a.z_all) return;
    // sorted union (values only are used downstream, run_nerf.py:396: torch.sort(torch.cat([z_vals, z_samples]))).  The
    // coarse depths are ascending already; the fine samples (unsorted when u is random) are sorted in LDS by a bitonic
    // network over the next power of two (padding = +inf), then every element finds its place in the union by one binary
    // search in the OTHER list (coarse before fine on ties: a stable merge).  O(n log^2 n) instead of the O(n^2) rank count;
    // the output is the same sorted sequence.
    float* fs = vals + Sc;
    int np2 = 1;
    while (np2 < Nf) np2 <<= 1;
    for (int k = Nf + lane; k < np2; k += 64) fs[k] = __int_as_float(0x7f800000);
    Sync::sync();
    for (int size = 2; size <= np2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = lane; t < (np2 >> 1); t += 64) {
                const int i = ((t / stride) * (stride << 1)) + (t % stride), j = i + stride;
                const bool up = ((i & size) == 0);
                const float x = fs[i], y = fs[j];
                if ((x > y) == up) { fs[i] = y; fs[j] = x; }
            }
            Sync::sync();
        }
    float* zo = a.z_all + (size_t)ray * tot;
    for (int i = lane; i < Sc; i += 64) {             // coarse element i: + number of fine samples strictly below it
        const float x = vals[i];
        int lo_ = 0, hi_ = Nf;
        while (lo_ < hi_) { const int mid = (lo_ + hi_) >> 1; if (fs[mid] < x) lo_ = mid + 1; else hi_ = mid; }
        zo[i + lo_] = x;
    }
    for (int k = lane; k < Nf; k += 64) {             // fine element k (sorted position): + number of coarse depths <= it
        const float x = fs[k];
        int lo_ = 0, hi_ = Sc;
        while (lo_ < hi_) { const int mid = (lo_ + hi_) >> 1; if (vals[mid] <= x) lo_ = mid + 1; else hi_ = mid; }
        zo[k + lo_] = x;
    }
}

}  // namespace nerf
