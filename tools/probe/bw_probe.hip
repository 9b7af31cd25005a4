// HBM read-bandwidth probes (not part of the product library): what can a streaming read reach on this GPU?
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// variant 0: plain dwordx4 loads, 8 in flight per thread, grid-stride over contiguous 1 KiB per wave-instruction
template <bool NT>
__global__ __launch_bounds__(512) void read_vgpr(const f32x4* __restrict__ src, size_t n16, float* out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    f32x4 s = {0, 0, 0, 0};
    for (; i + 7 * stride < n16; i += 8 * stride) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = 1.0f;
}
// variant 1: per-block contiguous chunk (like a GEMM operand stream): block b reads [b*chunk, (b+1)*chunk)
template <bool NT>
__global__ __launch_bounds__(512) void read_chunked(const f32x4* __restrict__ src, size_t chunk16, float* out) {
    const f32x4* p = src + (size_t)blockIdx.x * chunk16;
    f32x4 s = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i + 7 * 512 < chunk16; i += 8 * 512) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * 512) : p[i + u * 512];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = 1.0f;
}
extern "C" int probe_read(int variant, int nt, const void* src, size_t bytes, int blocks, float* out, void* stream) {
    const size_t n16 = bytes / 16;
    hipStream_t s = (hipStream_t)stream;
    if (variant == 0) {
        if (nt) hipLaunchKernelGGL(read_vgpr<true>, dim3(blocks), dim3(512), 0, s, (const f32x4*)src, n16, out);
        else hipLaunchKernelGGL(read_vgpr<false>, dim3(blocks), dim3(512), 0, s, (const f32x4*)src, n16, out);
    } else {
        const size_t chunk16 = n16 / blocks;
        if (nt) hipLaunchKernelGGL(read_chunked<true>, dim3(blocks), dim3(512), 0, s, (const f32x4*)src, chunk16, out);
        else hipLaunchKernelGGL(read_chunked<false>, dim3(blocks), dim3(512), 0, s, (const f32x4*)src, chunk16, out);
    }
    return (int)hipGetLastError();
}
