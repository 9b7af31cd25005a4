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
#include "ray_device.h"

namespace nerf {

// ------------------------------------------------------------------ launchable forms (one wavefront per ray)
__global__ void sample_coarse_kernel(const float* __restrict__ rays, int ray_stride, int n_rays,
                                     const float* __restrict__ t_vals, int S, int lindisp,
                                     const float* __restrict__ t_rand, float* __restrict__ z_out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n_rays * S) return;
    const int r = (int)(idx / S), j = (int)(idx - (long)r * S);
    const float near = rays[(long)r * ray_stride + 6], far = rays[(long)r * ray_stride + 7];
    z_out[idx] = coarse_depth(near, far, [&](int k) { return t_vals[k]; }, j, S, lindisp, t_rand ? t_rand + (long)r * S : nullptr);
}

template <bool BWD>
__global__ __launch_bounds__(64) void composite_kernel(CompositeArgs a) {
    extern __shared__ float sm[];
    composite_ray<BWD>(a, blockIdx.x, threadIdx.x, sm);
}

__global__ __launch_bounds__(64) void sample_fine_kernel(FineArgs a) {
    extern __shared__ float sm[];
    sample_fine_ray<BlockSync>(a, blockIdx.x, threadIdx.x, sm);
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

// ------------------------------------------------------------------ ray-batch selection of train() (run_nerf.py:726-757)
// N_rand DISTINCT pixels of one image (of its central crop during the precrop iterations), their rays as get_rays gives them
// (run_nerf_helpers.py:153-162) and their colours, in one launch of N_rand threads.  The reference draws
// np.random.choice(H*W, N_rand, replace=False) on the host (an O(H*W) permutation per step) and builds the full [H,W,3] ray grid
// first; here pixel k of the batch is perm(k), a KEYED BIJECTION of [0, nh*nw): a 6-round Feistel network over the next power
// of four, cycle-walked back into range (a value >= n is permuted again: still a bijection of [0, n)).  Distinct by construction;
// which subset comes out is decided by the two key words the host draws per step from its own generator.
__device__ inline unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ inline unsigned keyed_perm(unsigned k, unsigned n, int half_bits, unsigned key0, unsigned key1) {
    const unsigned mask = (1u << half_bits) - 1u;
    unsigned v = k;
    do {
        unsigned L = v >> half_bits, R = v & mask;
#pragma unroll
        for (unsigned r = 0; r < 6; ++r) {
            const unsigned F = mix32((R + 0x9e3779b9u * (r + 1u)) ^ (r & 1u ? key1 : key0)) & mask;
            const unsigned t = L ^ F;
            L = R;
            R = t;
        }
        v = (L << half_bits) | R;
    } while (v >= n);
    return v;
}
struct RayBatchArgs {
    int H, W, h0, w0, nh, nw, n_rand, half_bits;
    unsigned key0, key1;
    float fx, fy, cx, cy;
    const float* pose; int pose_stride;     // c2w[:3,:4] on the DEVICE, rows pose_stride floats apart
    const float* image;                     // [H][W][3]
    float* rays;                            // [2][n_rand][3] = (rays_o, rays_d)
    float* target;                          // [n_rand][3]
    int* pixels;                            // [n_rand] j * W + i of every selected pixel (nullable)
};
__global__ void sample_ray_batch_kernel(RayBatchArgs a) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.n_rand) return;
    const unsigned sel = keyed_perm((unsigned)k, (unsigned)(a.nh * a.nw), a.half_bits, a.key0, a.key1);
    const int jj = a.h0 + (int)(sel / (unsigned)a.nw), ii = a.w0 + (int)(sel % (unsigned)a.nw);
    float m[12];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) m[4 * r + c] = a.pose[r * a.pose_stride + c];
    const float dx = ((float)ii - a.cx) / a.fx, dy = -((float)jj - a.cy) / a.fy, dz = -1.0f;
    float o[3], d[3];
    camera_ray(m, dx, dy, dz, o, d);
    const float* px = a.image + ((long)jj * a.W + ii) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        a.rays[(long)k * 3 + c] = o[c];
        a.rays[((long)a.n_rand + k) * 3 + c] = d[c];
        a.target[(long)k * 3 + c] = px[c];
    }
    if (a.pixels) a.pixels[k] = jj * a.W + ii;
}
hipError_t launch_sample_ray_batch(int H, int W, const float* K9, const float* pose_dev, int pose_stride, const float* image, int h0, int w0,
                                   int nh, int nw, int n_rand, unsigned key0, unsigned key1, float* rays, float* target, int* pixels,
                                   hipStream_t stream) {
    if (n_rand <= 0) return hipSuccess;
    RayBatchArgs a{};
    a.H = H; a.W = W; a.h0 = h0; a.w0 = w0; a.nh = nh; a.nw = nw; a.n_rand = n_rand;
    a.half_bits = 1;
    while ((1ull << (2 * a.half_bits)) < (unsigned long long)nh * (unsigned long long)nw) ++a.half_bits;
    a.key0 = key0; a.key1 = key1;
    a.fx = K9[0]; a.cx = K9[2]; a.fy = K9[4]; a.cy = K9[5];
    a.pose = pose_dev; a.pose_stride = pose_stride; a.image = image; a.rays = rays; a.target = target; a.pixels = pixels;
    hipLaunchKernelGGL(sample_ray_batch_kernel, dim3((unsigned)((n_rand + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------ launchers
// ------------------------------------------------------------------ img2mse (run_nerf_helpers.py:11) and its gradient
// mean((x - y)^2) over n elements in ONE launch, deterministic: every block sums its contiguous slice in a fixed tree, writes its
// partial sum, and the block that takes the last ticket adds the partials in block order.  scratch: [MSE_BLOCKS] partial sums +
// one ticket word (zero before the launch; the last block resets it).
constexpr int MSE_BLOCKS = 256, MSE_THREADS = 256;
__global__ __launch_bounds__(MSE_THREADS) void mse_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, long n,
                                                               float* __restrict__ scratch, float* __restrict__ out) {
    __shared__ float red[MSE_THREADS];
    __shared__ bool last;
    const int nb = gridDim.x;
    const long per = (n + nb - 1) / nb, lo = blockIdx.x * per, hi = min(lo + per, n);
    float s = 0.0f;
    for (long i = lo + threadIdx.x; i < hi; i += MSE_THREADS) { const float d = x[i] - y[i]; s += d * d; }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = MSE_THREADS / 2; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    unsigned* ticket = reinterpret_cast<unsigned*>(scratch + MSE_BLOCKS);
    if (threadIdx.x == 0) {
        scratch[blockIdx.x] = red[0];
        __threadfence();
        last = atomicAdd(ticket, 1u) == (unsigned)nb - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    // (agent-scope load: the partial sums were written by other CUs; read them from L2, not from this CU's vector cache)
    red[threadIdx.x] = threadIdx.x < nb ? __uint_as_float(__hip_atomic_load(reinterpret_cast<unsigned*>(scratch) + threadIdx.x, __ATOMIC_RELAXED,
                                                                            __HIP_MEMORY_SCOPE_AGENT)) : 0.0f;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
        for (int b = 0; b < nb; ++b) t += red[b];
        out[0] = t / (float)n;
        *ticket = 0u;
    }
}
// d/dx mean((x - y)^2) * g = (2 g / n) (x - y); g is a device scalar (the upstream gradient of the loss)
__global__ void mse_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, long n, const float* __restrict__ g,
                               float* __restrict__ dx) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float c = 2.0f * g[0] / (float)n;
    dx[i] = c * (x[i] - y[i]);
}

// ---- range check of the fp16 split (nerf_range_scan): largest |fp16| bit pattern among packed halves -- the post-ReLU rows a forward
// saved (non-negative, or NaN of either sign once the overflow has happened) or the signed deltas of a delta chain.  With the sign bits
// cleared the patterns order like integers: NaN above inf above every finite magnitude.
__global__ __launch_bounds__(256) void range_scan_kernel(const unsigned* __restrict__ rows, size_t n4, size_t n_words, unsigned* __restrict__ words) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    unsigned m = 0u;
    const u32x4* r4 = reinterpret_cast<const u32x4*>(rows);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const u32x4 v = __builtin_nontemporal_load(r4 + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned w = v[k] & 0x7fff7fffu;
            asm("v_pk_max_u16 %0, %0, %1" : "+v"(m) : "v"(w));
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n_words - 4 * n4)) {       // a tail shorter than 16 bytes (region sizes are multiples of 4 words: none in practice)
        const unsigned v = rows[4 * n4 + threadIdx.x] & 0x7fff7fffu;
        asm("v_pk_max_u16 %0, %0, %1" : "+v"(m) : "v"(v));
    }
    unsigned top = max(m & 0xffffu, m >> 16);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) top = max(top, (unsigned)__shfl_xor((int)top, o));
    __shared__ unsigned wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = top;
    __syncthreads();
    if (threadIdx.x == 0) {
        top = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
        if (top > 0u) atomicMax(words + 1, top);
        if (top >= 0x7800u) atomicOr(words, 1u);
    }
}

hipError_t launch_range_scan(const unsigned* rows, size_t n_words, unsigned* words, hipStream_t stream) {
    if (n_words == 0) return hipSuccess;
    const size_t n4 = n_words / 4;
    const unsigned blocks = (unsigned)min((size_t)4096, (n4 + 255) / 256 + 1);
    hipLaunchKernelGGL(range_scan_kernel, dim3(blocks), dim3(256), 0, stream, rows, n4, n_words, words);
    return hipGetLastError();
}

hipError_t launch_mse_fwd(const float* x, const float* y, long n, float* scratch, float* out, hipStream_t stream) {
    const int nb = (int)min((long)MSE_BLOCKS, (n + 4 * MSE_THREADS - 1) / (4 * MSE_THREADS));
    hipLaunchKernelGGL(mse_fwd_kernel, dim3(nb < 1 ? 1 : nb), dim3(MSE_THREADS), 0, stream, x, y, n, scratch, out);
    return hipGetLastError();
}
hipError_t launch_mse_bwd(const float* x, const float* y, long n, const float* g, float* dx, hipStream_t stream) {
    hipLaunchKernelGGL(mse_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, y, n, g, dx);
    return hipGetLastError();
}

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
