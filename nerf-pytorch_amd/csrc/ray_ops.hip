// Per-ray kernels of the volumetric renderer: one wavefront per ray, every
// along-ray scan (transmittance cumprod, CDF, suffix sums) is a per-lane
// sequential segment + one wave-level shuffle scan.
//
//   sample_coarse_kernel   run_nerf.py:357-379   (z_vals: linspace / lindisp / stratified jitter)
//   composite_fwd_kernel   run_nerf.py:262-305   (raw2outputs)
//   composite_bwd_kernel   autograd of run_nerf.py:275-303
//   sample_fine_kernel     run_nerf.py:392-396,412 + run_nerf_helpers.py:196-239 (sample_pdf, sort, z_std)
//   embed_kernel           run_nerf_helpers.py:15-45 (Embedder.embed, standalone form)
//
// Compiled with -ffp-contract=off: the elementwise arithmetic below is written
// in the reference's operation order so z_vals / dists / pts round identically.
#include <hip/hip_runtime.h>
#include "nerf_common.h"
#include "launchers.h"

namespace nerf {

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

__global__ void sample_coarse_kernel(const float* __restrict__ rays, int ray_stride, int n_rays,
                                     const float* __restrict__ t_vals, int S, int lindisp,
                                     const float* __restrict__ t_rand, float* __restrict__ z_out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n_rays * S) return;
    const int r = (int)(idx / S), j = (int)(idx - (long)r * S);
    const float near = rays[(long)r * ray_stride + 6], far = rays[(long)r * ray_stride + 7];
    const float zj = z_at(near, far, t_vals[j], lindisp);
    if (!t_rand) { z_out[idx] = zj; return; }
    const float z_last = z_at(near, far, t_vals[S - 1], lindisp);
    const float z_first = z_at(near, far, t_vals[0], lindisp);
    float upper, lower;
    if (j < S - 1) upper = 0.5f * (z_at(near, far, t_vals[j + 1], lindisp) + zj); else upper = z_last;
    if (j > 0) lower = 0.5f * (zj + z_at(near, far, t_vals[j - 1], lindisp)); else lower = z_first;
    z_out[idx] = lower + (upper - lower) * t_rand[idx];
}

// ------------------------------------------------------------------ compositing
__device__ inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// LDS per wave: 4*S floats (fwd) / 8*S floats (bwd)
template <bool BWD>
__global__ __launch_bounds__(64) void composite_kernel(CompositeArgs a) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x;
    const int ray = blockIdx.x;
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
__global__ __launch_bounds__(64) void sample_fine_kernel(FineArgs a) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x, ray = blockIdx.x;
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
    __syncthreads();
    // cdf = [0, cumsum(pdf)]: the ADDITIONS are sequential like torch.cumsum so rounding follows the reference; the divisions
    // (independent) are done by all lanes first
    for (int i = lane; i < nw; i += 64) cdf[i + 1] = (w[i] + 1e-5f) / wsum;
    __syncthreads();
    if (lane == 0) {
        float c = 0.0f;
        cdf[0] = 0.0f;
        for (int i = 0; i < nw; ++i) { c += cdf[i + 1]; cdf[i + 1] = c; }
    }
    __syncthreads();
    float s1 = 0.0f;
    for (int k = lane; k < Nf; k += 64) {
        const float u = a.u ? a.u[(size_t)ray * Nf + k] : a.u_lin[k];
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
    __syncthreads();
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
    __syncthreads();
    for (int size = 2; size <= np2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = lane; t < (np2 >> 1); t += 64) {
                const int i = ((t / stride) * (stride << 1)) + (t % stride), j = i + stride;
                const bool up = ((i & size) == 0);
                const float x = fs[i], y = fs[j];
                if ((x > y) == up) { fs[i] = y; fs[j] = x; }
            }
            __syncthreads();
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

// ------------------------------------------------------------------ standalone positional encoding
__global__ void embed_kernel(const float* __restrict__ x, long n_pts, int n_freqs, float* __restrict__ out) {
    const int width = 3 + 6 * n_freqs;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_pts * width) return;
    const long p = idx / width;
    const int c = (int)(idx - p * width);
    float v;
    if (c < 3) v = x[p * 3 + c];
    else {
        const int g = c - 3, f = g / 6, fn = (g % 6) / 3, dim = g % 3;
        const float arg = x[p * 3 + dim] * __int_as_float((127 + f) << 23);
        v = fn == 0 ? sinf(arg) : cosf(arg);
    }
    out[idx] = v;
}

// ------------------------------------------------------------------ ray set-up of render(c2w=...)
// get_rays (run_nerf_helpers.py:153-162) + view directions (run_nerf.py:100-107) + ndc_rays (helpers:175-192) +
// near / far columns (run_nerf.py:117-123) in one pass: pixel (j, i) -> rays[j*W + i][0..10] = (o3, d3, near, far,
// viewdir3).  No [H,W,3] intermediates; the arithmetic follows the reference's operation order.
struct RayGenArgs {
    int H, W;
    float fx, fy, cx, cy;
    float pose[12];         // c2w[:3,:4] row-major
    float pose_static[12];  // c2w_staticcam[:3,:4] (rays come from this camera, view directions from `pose`)
    int has_static, ndc;
    float near, far;
    float* rays;
    int stride;
};
__device__ inline void camera_ray(const float* m, float dx, float dy, float dz, float (&o)[3], float (&d)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        d[k] = (dx * m[4 * k] + dy * m[4 * k + 1]) + dz * m[4 * k + 2];      // torch.sum(dirs[..., None, :] * c2w[:3,:3], -1)
        o[k] = m[4 * k + 3];
    }
}
// ndc_rays(H, W, focal, near = 1., rays_o, rays_d) (run_nerf_helpers.py:175-192), same operation order
__device__ inline void ndc_warp_inplace(int H, int W, float focal, float (&o)[3], float (&d)[3]) {
    const float t = -(1.0f + o[2]) / d[2];
    o[0] = o[0] + t * d[0];
    o[1] = o[1] + t * d[1];
    o[2] = o[2] + t * d[2];
    const float sw = -1.0f / ((float)W / (2.0f * focal)), sh = -1.0f / ((float)H / (2.0f * focal));
    const float n0 = sw * o[0] / o[2], n1 = sh * o[1] / o[2], n2 = 1.0f + 2.0f * 1.0f / o[2];
    const float e0 = sw * (d[0] / d[2] - o[0] / o[2]), e1 = sh * (d[1] / d[2] - o[1] / o[2]), e2 = -2.0f * 1.0f / o[2];
    o[0] = n0; o[1] = n1; o[2] = n2;
    d[0] = e0; d[1] = e1; d[2] = e2;
}

// Ray records of render(rays=(rays_o, rays_d), use_viewdirs=True) (run_nerf.py:95-123 without c2w): view directions
// d / |d| from the WORLD-space directions, the NDC warp iff ndc, the near / far columns -- one launch instead of the
// ~10 (no NDC) / ~35 (NDC) elementwise torch kernels of the reference formulation.
__global__ void assemble_rays_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, long n, int ndc,
                                     int H, int W, float focal, float near, float far, float* __restrict__ rays, int stride) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    float o[3] = {rays_o[3 * idx], rays_o[3 * idx + 1], rays_o[3 * idx + 2]};
    float d[3] = {rays_d[3 * idx], rays_d[3 * idx + 1], rays_d[3 * idx + 2]};
    const float nrm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    const float v0 = d[0] / nrm, v1 = d[1] / nrm, v2 = d[2] / nrm;
    if (ndc) ndc_warp_inplace(H, W, focal, o, d);
    float* r = rays + idx * stride;
    r[0] = o[0]; r[1] = o[1]; r[2] = o[2];
    r[3] = d[0]; r[4] = d[1]; r[5] = d[2];
    r[6] = near; r[7] = far;
    r[8] = v0; r[9] = v1; r[10] = v2;
}

__global__ void make_rays_kernel(RayGenArgs a) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)a.H * a.W) return;
    const int j = (int)(idx / a.W), i = (int)(idx - (long)j * a.W);
    const float dx = ((float)i - a.cx) / a.fx, dy = -((float)j - a.cy) / a.fy, dz = -1.0f;
    float o[3], d[3];
    camera_ray(a.pose, dx, dy, dz, o, d);
    const float nrm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    const float v0 = d[0] / nrm, v1 = d[1] / nrm, v2 = d[2] / nrm;
    if (a.has_static) camera_ray(a.pose_static, dx, dy, dz, o, d);
    if (a.ndc) ndc_warp_inplace(a.H, a.W, a.fx, o, d);
    float* r = a.rays + idx * a.stride;
    r[0] = o[0]; r[1] = o[1]; r[2] = o[2];
    r[3] = d[0]; r[4] = d[1]; r[5] = d[2];
    r[6] = a.near; r[7] = a.far;
    r[8] = v0; r[9] = v1; r[10] = v2;
}

// ------------------------------------------------------------------ launchers
hipError_t launch_sample_coarse(const float* rays, int ray_stride, int n_rays, const float* t_vals, int S,
                                int lindisp, const float* t_rand, float* z_out, hipStream_t stream) {
    const long n = (long)n_rays * S;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(sample_coarse_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       rays, ray_stride, n_rays, t_vals, S, lindisp, t_rand, z_out);
    return hipGetLastError();
}

hipError_t launch_composite(const CompositeArgs& a, bool bwd, hipStream_t stream) {
    if (a.n_rays <= 0) return hipSuccess;
    if (bwd) hipLaunchKernelGGL(composite_kernel<true>, dim3(a.n_rays), dim3(64), 4 * a.S * sizeof(float), stream, a);
    else hipLaunchKernelGGL(composite_kernel<false>, dim3(a.n_rays), dim3(64), 2 * a.S * sizeof(float), stream, a);
    return hipGetLastError();
}

hipError_t launch_sample_fine(const FineArgs& a, hipStream_t stream) {
    if (a.n_rays <= 0) return hipSuccess;
    int np2 = 1;
    while (np2 < a.Nf) np2 <<= 1;                        // the fine samples are sorted over the next power of two
    const size_t lds = (size_t)(3 * a.n_in + np2) * sizeof(float);
    hipLaunchKernelGGL(sample_fine_kernel, dim3(a.n_rays), dim3(64), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_make_rays(int H, int W, const float* K9, const float* pose12, const float* pose_static12, int ndc,
                            float near, float far, float* rays, int ray_stride, hipStream_t stream) {
    const long n = (long)H * W;
    if (n <= 0) return hipSuccess;
    RayGenArgs a{};
    a.H = H; a.W = W;
    a.fx = K9[0]; a.cx = K9[2]; a.fy = K9[4]; a.cy = K9[5];
    for (int i = 0; i < 12; ++i) { a.pose[i] = pose12[i]; a.pose_static[i] = pose_static12 ? pose_static12[i] : 0.0f; }
    a.has_static = pose_static12 ? 1 : 0;
    a.ndc = ndc; a.near = near; a.far = far; a.rays = rays; a.stride = ray_stride;
    hipLaunchKernelGGL(make_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_assemble_rays(const float* rays_o, const float* rays_d, long n, int ndc, int H, int W, float focal,
                                float near, float far, float* rays, int ray_stride, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(assemble_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rays_o, rays_d, n, ndc,
                       H, W, focal, near, far, rays, ray_stride);
    return hipGetLastError();
}

hipError_t launch_embed(const float* x, long n_pts, int n_freqs, float* out, hipStream_t stream) {
    const long n = n_pts * (3 + 6 * n_freqs);
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, n_pts, n_freqs, out);
    return hipGetLastError();
}

}  // namespace nerf

// ------------------------------------------------------------------ fused Adam over a flat parameter vector
// (torch.optim.Adam defaults of run_nerf.py:207: no amsgrad, no weight decay).  Same operation order as
// torch's single-tensor Adam: exp_avg.lerp_(g, 1-b1); exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2);
// denom = sqrt(exp_avg_sq)/sqrt(bias2) + eps; p -= (lr/bias1) * exp_avg/denom.
namespace nerf {
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            int n, float lr_over_bc1, float one_minus_b1, float b2, float one_minus_b2, float sqrt_bc2, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = m[i] + one_minus_b1 * (gi - m[i]);
    const float vi = v[i] * b2 + one_minus_b2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / sqrt_bc2 + eps;
    p[i] = p[i] - lr_over_bc1 * (mi / denom);
}
hipError_t launch_adam(float* p, const float* g, float* m, float* v, int n, float lr, float b1, float b2, float eps, int step,
                       hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, p, g, m, v, n, (float)(lr / bc1),
                       (float)(1.0 - (double)b1), b2, (float)(1.0 - (double)b2), (float)sqrt(bc2), eps);
    return hipGetLastError();
}
}  // namespace nerf
