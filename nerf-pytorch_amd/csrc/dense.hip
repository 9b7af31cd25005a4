// Plain dense layers for the architectures the fused field kernels do not cover (run_nerf.py:435-442 netdepth / netwidth,
// --multires / --multires_views, --i_embed -1, use_viewdirs=False -> output_linear, run_nerf_helpers.py:93-94,117): the
// reference's layer stack (helpers:96-119) evaluated layer by layer.  The GEMMs are plain library SGEMMs (rocBLAS, exact
// fp32, atomics off = deterministic), loaded on first use so that libnerf_hip.so has no link-time dependency on it; bias /
// ReLU / ReLU-mask epilogues, the bias gradient and the network input (o + d z, positional encodings, direction broadcast:
// run_nerf.py:41-47, :381) are HIP kernels here.  Everything is row-major with explicit leading dimensions, so a layer can
// read a column range of a wider matrix (skip connection, view branch) and write one (rgb | alpha): no concatenations.
// The BASELINE architecture never comes here: it runs on the fused kernels.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <mutex>
#include "api_util.h"
#include "nerf_common.h"
#include "../../include/nerf_hip.h"

namespace {

using nerf_api::done;
using nerf_api::fail_arg;

// ---- rocBLAS, resolved at run time (SONAME librocblas.so.5 is shared with the copy PyTorch-ROCm has already loaded)
typedef void* rb_handle;
typedef int (*rb_create_t)(rb_handle*);
typedef int (*rb_set_stream_t)(rb_handle, hipStream_t);
typedef int (*rb_set_atomics_t)(rb_handle, int);
typedef int (*rb_sgemm_t)(rb_handle, int, int, int, int, int, const float*, const float*, int, const float*, int, const float*, float*, int);
constexpr int RB_OP_N = 111, RB_OP_T = 112;        // rocblas_operation_none / _transpose
constexpr int RB_ATOMICS_NOT_ALLOWED = 0;

struct RocBlas {
    rb_handle h = nullptr;
    rb_set_stream_t set_stream = nullptr;
    rb_sgemm_t sgemm = nullptr;
    const char* error = nullptr;
};
RocBlas g_rb;
std::mutex g_rb_mutex;

const RocBlas& rocblas() {
    std::lock_guard<std::mutex> lock(g_rb_mutex);
    if (g_rb.h || g_rb.error) return g_rb;
    void* lib = nullptr;
    for (const char* name : {"librocblas.so.5", "librocblas.so", "/opt/rocm/lib/librocblas.so"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) { g_rb.error = "rocBLAS (librocblas.so) not found: the dense-layer path needs it"; return g_rb; }
    auto create = (rb_create_t)dlsym(lib, "rocblas_create_handle");
    auto atomics = (rb_set_atomics_t)dlsym(lib, "rocblas_set_atomics_mode");
    g_rb.set_stream = (rb_set_stream_t)dlsym(lib, "rocblas_set_stream");
    g_rb.sgemm = (rb_sgemm_t)dlsym(lib, "rocblas_sgemm");
    if (!create || !atomics || !g_rb.set_stream || !g_rb.sgemm) { g_rb.error = "rocBLAS: missing symbol"; return g_rb; }
    if (create(&g_rb.h) != 0 || !g_rb.h) { g_rb.h = nullptr; g_rb.error = "rocblas_create_handle failed"; return g_rb; }
    atomics(g_rb.h, RB_ATOMICS_NOT_ALLOWED);        // no split-K atomics: run-to-run identical sums
    return g_rb;
}

// ---- epilogues
__global__ void bias_act_kernel(float* __restrict__ y, int ldy, int N, long P, const float* __restrict__ bias, int relu) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * N) return;
    const long p = i / N;
    const int n = (int)(i - p * N);
    float v = y[p * ldy + n];
    if (bias) v = v + bias[n];
    if (relu) v = fmaxf(v, 0.0f);
    y[p * ldy + n] = v;
}
// dx *= (act > 0): the ReLU of the layer that produced `act` (its post-activation output)
__global__ void relu_mask_kernel(float* __restrict__ dx, int lddx, int K, long P, const float* __restrict__ act, int ldact) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * K) return;
    const long p = i / K;
    const int k = (int)(i - p * K);
    if (!(act[p * ldact + k] > 0.0f)) dx[p * lddx + k] = 0.0f;
}
// db[n] = sum_p dy[p][n], deterministic: COLSUM_ROWS rows per block into partial[block][n], then one pass over the blocks
constexpr int COLSUM_ROWS = 2048;
__global__ void colsum_partial_kernel(const float* __restrict__ dy, int lddy, int N, long P, float* __restrict__ partial) {
    const int n = blockIdx.y * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const long p0 = (long)blockIdx.x * COLSUM_ROWS, p1 = min(p0 + COLSUM_ROWS, P);
    float s = 0.0f;
    for (long p = p0; p < p1; ++p) s += dy[p * lddy + n];
    partial[(long)blockIdx.x * N + n] = s;
}
__global__ void colsum_final_kernel(const float* __restrict__ partial, int n_blocks, int N, float* __restrict__ db, int accumulate) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.0f;
    for (int b = 0; b < n_blocks; ++b) s += partial[(long)b * N + n];
    db[n] = accumulate ? db[n] + s : s;
}

// ---- network input: row p = (ray r, sample j) of x = [enc(o + d z) | enc(viewdir)] (run_nerf.py:381, :41-47)
__device__ inline void encode3(float* out, float v0, float v1, float v2, int L) {
    out[0] = v0; out[1] = v1; out[2] = v2;
    for (int f = 0; f < L; ++f) {
        const float s = __int_as_float((127 + f) << 23);        // 2^f (log-sampled bands, helpers:32-33)
        const float a0 = v0 * s, a1 = v1 * s, a2 = v2 * s;
        float* o = out + 3 + 6 * f;
        o[0] = sinf(a0); o[1] = sinf(a1); o[2] = sinf(a2);
        o[3] = cosf(a0); o[4] = cosf(a1); o[5] = cosf(a2);
    }
}
__global__ void build_inputs_kernel(const float* __restrict__ rays, int ray_stride, const float* __restrict__ z, long P, int S,
                                    int L_xyz, int L_dir, int use_viewdirs, float* __restrict__ x, int ldx) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const long r = p / S;
    const float* rp = rays + r * ray_stride;
    const float zz = z[p];
    float* row = x + p * ldx;
    // pts = rays_o + rays_d * z with separate multiply / add like the reference
    const float x0 = rp[0] + rp[3] * zz, x1 = rp[1] + rp[4] * zz, x2 = rp[2] + rp[5] * zz;
    const int cx = L_xyz < 0 ? 3 : 3 + 6 * L_xyz;
    if (L_xyz < 0) { row[0] = x0; row[1] = x1; row[2] = x2; }
    else encode3(row, x0, x1, x2, L_xyz);
    if (use_viewdirs) {
        const float* vd = rp + (ray_stride - 3);
        if (L_dir < 0) { row[cx] = vd[0]; row[cx + 1] = vd[1]; row[cx + 2] = vd[2]; }
        else encode3(row + cx, vd[0], vd[1], vd[2], L_dir);
    }
}

// the handle is shared: stream binding and the GEMM it applies to must not interleave between host threads
std::mutex g_gemm_mutex;
int gemm(const RocBlas& rb, hipStream_t st, int ta, int tb, int m, int n, int k, const float* A, int lda, const float* B, int ldb,
         float beta, float* C, int ldc) {
    std::lock_guard<std::mutex> lock(g_gemm_mutex);
    const float one = 1.0f;
    if (rb.set_stream(rb.h, st) != 0) return 1;
    return rb.sgemm(rb.h, ta, tb, m, n, k, &one, A, lda, B, ldb, &beta, C, ldc);
}

int check_dims(const char* fn, long P, int a, int b) {
    if (P < 0 || a <= 0 || b <= 0 || P > 0x7fffffffL) return fail_arg(fn, "bad size (rows must fit an int, widths > 0)");
    return 0;
}
unsigned blocks_for(long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" {

int nerf_dense_fwd(const float* x, int ldx, int K, const float* w, int ldw, const float* bias, float* y, int ldy, int N, long P,
                   int accumulate, int relu, void* stream) {
    if (!x || !w || !y) return fail_arg(__func__, "null pointer");
    if (check_dims(__func__, P, K, N) || ldx < K || ldw < K || ldy < N) return fail_arg(__func__, "bad size / leading dimension");
    if (P == 0) return 0;
    const RocBlas& rb = rocblas();
    if (!rb.h) return fail_arg(__func__, rb.error);
    hipStream_t st = (hipStream_t)stream;
    // row-major y[P,N] = x[P,K] w[N,K]^T  ==  column-major y^T (N x P) = w (K x N, ld ldw)^T * x^T (K x P, ld ldx)
    if (gemm(rb, st, RB_OP_T, RB_OP_N, N, (int)P, K, w, ldw, x, ldx, accumulate ? 1.0f : 0.0f, y, ldy) != 0) return fail_arg(__func__, "rocblas_sgemm failed");
    if (bias || relu) hipLaunchKernelGGL(bias_act_kernel, dim3(blocks_for(P * N)), dim3(256), 0, st, y, ldy, N, P, bias, relu);
    return done(__func__, hipGetLastError());
}

int nerf_dense_dgrad(const float* dy, int lddy, int N, const float* w, int ldw, float* dx, int lddx, int K, long P, int accumulate,
                     const float* act, int ldact, void* stream) {
    if (!dy || !w || !dx) return fail_arg(__func__, "null pointer");
    if (check_dims(__func__, P, K, N) || lddy < N || ldw < K || lddx < K || (act && ldact < K)) return fail_arg(__func__, "bad size / leading dimension");
    if (P == 0) return 0;
    const RocBlas& rb = rocblas();
    if (!rb.h) return fail_arg(__func__, rb.error);
    hipStream_t st = (hipStream_t)stream;
    // row-major dx[P,K] = dy[P,N] w[N,K]  ==  column-major dx^T (K x P) = w (K x N, ld ldw) * dy^T (N x P, ld lddy)
    if (gemm(rb, st, RB_OP_N, RB_OP_N, K, (int)P, N, w, ldw, dy, lddy, accumulate ? 1.0f : 0.0f, dx, lddx) != 0) return fail_arg(__func__, "rocblas_sgemm failed");
    if (act) hipLaunchKernelGGL(relu_mask_kernel, dim3(blocks_for(P * K)), dim3(256), 0, st, dx, lddx, K, P, act, ldact);
    return done(__func__, hipGetLastError());
}

size_t nerf_dense_wgrad_scratch_floats(long P, int N) {
    if (P <= 0 || N <= 0) return 0;
    return (size_t)((P + COLSUM_ROWS - 1) / COLSUM_ROWS) * (size_t)N;
}

int nerf_dense_wgrad(const float* dy, int lddy, int N, const float* x, int ldx, int K, long P, float* dw, int lddw, float* dbias,
                     float* scratch, int accumulate, void* stream) {
    if (!dy || !x || !dw) return fail_arg(__func__, "null pointer");
    if (check_dims(__func__, P, K, N) || lddy < N || ldx < K || lddw < K) return fail_arg(__func__, "bad size / leading dimension");
    if (dbias && !scratch) return fail_arg(__func__, "the bias gradient needs nerf_dense_wgrad_scratch_floats(P, N) floats of scratch");
    if (P == 0) return 0;
    const RocBlas& rb = rocblas();
    if (!rb.h) return fail_arg(__func__, rb.error);
    hipStream_t st = (hipStream_t)stream;
    // row-major dw[N,K] = dy[P,N]^T x[P,K]  ==  column-major dw^T (K x N, ld lddw) = x^T (K x P, ld ldx) * (dy^T (N x P, ld lddy))^T
    if (gemm(rb, st, RB_OP_N, RB_OP_T, K, N, (int)P, x, ldx, dy, lddy, accumulate ? 1.0f : 0.0f, dw, lddw) != 0) return fail_arg(__func__, "rocblas_sgemm failed");
    if (dbias) {
        const int nb = (int)((P + COLSUM_ROWS - 1) / COLSUM_ROWS);
        hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)nb, (unsigned)((N + 63) / 64)), dim3(64), 0, st, dy, lddy, N, P, scratch);
        hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, st, scratch, nb, N, dbias, accumulate);
    }
    return done(__func__, hipGetLastError());
}

int nerf_build_inputs(const float* rays, int ray_stride, const float* z_vals, int n_rays, int n_samples, int multires, int multires_views,
                      int use_viewdirs, float* x, int ldx, void* stream) {
    if (!rays || !z_vals || !x) return fail_arg(__func__, "null pointer");
    const int cx = multires < 0 ? 3 : 3 + 6 * multires, cd = !use_viewdirs ? 0 : (multires_views < 0 ? 3 : 3 + 6 * multires_views);
    if (n_rays < 0 || n_samples <= 0 || multires > 24 || multires_views > 24 || ldx < cx + cd ||
        ray_stride < (use_viewdirs ? 11 : 8)) return fail_arg(__func__, "bad size (rays: o3 d3 near far [viewdir3])");
    const long P = (long)n_rays * n_samples;
    if (P == 0) return 0;
    hipLaunchKernelGGL(build_inputs_kernel, dim3(blocks_for(P)), dim3(256), 0, (hipStream_t)stream, rays, ray_stride, z_vals, P, n_samples,
                       multires, multires_views, use_viewdirs, x, ldx);
    return done(__func__, hipGetLastError());
}

}  // extern "C"
