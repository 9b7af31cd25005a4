// Store-coalescing probe (not product code): every wave writes 256 B per instruction as four 64-byte segments (one per
// 16-lane quarter, 128 B apart, like rows 4q of a 16-point bf16 tile), 16 instructions per loop iteration so that a
// wave covers 2 x 2 KiB contiguous.  pattern 0: lane-linear inside the segment; 1: even lanes first half, odd lanes
// second half interleaved (lane 2j -> dword j, lane 2j+1 -> dword 8 + j: the paired-row store of field_fwd16_kernel<2>);
// 2: like 1 but both halves written by separate instructions (32-byte runs).  nt stores.
#include <hip/hip_runtime.h>
template <int PATTERN>
__global__ __launch_bounds__(512) void coal_k(unsigned* __restrict__ dst, size_t n_dwords) {
    const int lane = threadIdx.x & 63, q = lane >> 4, l16 = lane & 15;
    const size_t wave = (size_t)blockIdx.x * 8 + (threadIdx.x >> 6), n_waves = (size_t)gridDim.x * 8;
    int within;
    if (PATTERN == 0) within = l16;
    else within = (l16 & 1) * 8 + (l16 >> 1);
    for (size_t blk = wave; (blk + 1) * 1024 <= n_dwords; blk += n_waves) {      // 4 KiB per wave and iteration
        unsigned* base = dst + blk * 1024;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            // instruction i: segment (q, i): dwords [ (i>>1)*128 + q*32 + (i&1)*16 , +16 )
            unsigned* p = base + (i >> 1) * 128 + q * 32 + (i & 1) * 16 + within;
            if (PATTERN == 2) { if ((l16 & 1) == (i & 1)) __builtin_nontemporal_store((unsigned)lane, p); else __builtin_nontemporal_store((unsigned)lane + 1u, p); }
            else __builtin_nontemporal_store((unsigned)lane, p);
        }
    }
}
extern "C" int probe_coal(int pattern, void* dst, size_t bytes, int blocks, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (pattern == 0) hipLaunchKernelGGL(coal_k<0>, dim3(blocks), dim3(512), 0, s, (unsigned*)dst, bytes / 4);
    else if (pattern == 1) hipLaunchKernelGGL(coal_k<1>, dim3(blocks), dim3(512), 0, s, (unsigned*)dst, bytes / 4);
    else hipLaunchKernelGGL(coal_k<2>, dim3(blocks), dim3(512), 0, s, (unsigned*)dst, bytes / 4);
    return (int)hipGetLastError();
}
