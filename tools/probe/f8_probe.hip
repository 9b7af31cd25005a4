// Probe for the fp8 correction terms (round 4): v_mfma_scale_f32_16x16x128_f8f6f4 operand layout and scale operands,
// v_cvt_scalef32_pk_fp8_f32 semantics.  a / b: per lane 8 dwords as loaded (the host builds them for the layout hypothesis).
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(64) void mfma8_k(const int* a, const int* b, float* d, int sa, int sb) {
    const int l = threadIdx.x;
    i32x8 A, B;
    for (int j = 0; j < 8; ++j) { A[j] = a[l * 8 + j]; B[j] = b[l * 8 + j]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, c, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
__global__ void cvt_k(const float* vals, int n, int* out, float scale) {
    const int i = threadIdx.x + blockIdx.x * 64;
    if (2 * i + 1 >= n + 1) return;
    i16x2 old = {0x1111, 0x2222};
    const i16x2 lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, vals[2 * i], vals[2 * i + 1], scale, false);
    const i16x2 hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, vals[2 * i], vals[2 * i + 1], scale, true);
    out[2 * i] = __builtin_bit_cast(int, lo);
    out[2 * i + 1] = __builtin_bit_cast(int, hi);
}
extern "C" int probe_mfma8(const int* a, const int* b, float* d, int sa, int sb, void* s) {
    hipLaunchKernelGGL(mfma8_k, dim3(1), dim3(64), 0, (hipStream_t)s, a, b, d, sa, sb);
    return (int)hipGetLastError();
}
extern "C" int probe_cvt8(const float* v, int n, int* out, float scale, void* s) {
    hipLaunchKernelGGL(cvt_k, dim3((n / 2 + 63) / 64), dim3(64), 0, (hipStream_t)s, v, n, out, scale);
    return (int)hipGetLastError();
}
