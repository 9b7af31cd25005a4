// HBM write-bandwidth probes (not part of the product library): what do the save patterns of the field kernels reach?
//   mode 0: dwordx4 stores, lane-linear (1 KiB per wave-instruction, full lines)
//   mode 1: dword stores, lane-linear (256 B per wave-instruction = 2 full lines)
//   mode 2: dword stores in the 16-point forward's pattern: per instruction 4 runs of 64 B, 512 B apart
//           (lane = (pt = lane&15, q = lane>>4) -> feature 4q + j of a [feature][32 points] tile; the other 64 B of
//           each line are written by the partner wave of the tile, i.e. by another wave of the same workgroup)
//   mode 3: as mode 2 but a wave writes BOTH halves itself (two instructions back to back)
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool NT, typename T> __device__ inline void st(T* p, T v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <int MODE, bool NT>
__global__ __launch_bounds__(512) void write_k(float* __restrict__ dst, size_t nfloats) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // every workgroup owns contiguous "tiles" of 256 features x 32 points x 4 B = 32 KiB; 8 waves: wave w writes tile w/2,
    // point half w&1 (mode 2) of the workgroup's group of 4 tiles
    const size_t tile_f = 256 * 32;
    const size_t groups = nfloats / (4 * tile_f);
    for (size_t g = blockIdx.x; g < groups; g += gridDim.x) {
        float* base = dst + g * 4 * tile_f;
        if (MODE == 0) {
            f32x4* p = reinterpret_cast<f32x4*>(base) + wave * 1024 + lane;          // 16 KiB per wave
#pragma unroll
            for (int i = 0; i < 16; ++i) st<NT>(p + i * 64, f32x4{1.f, 2.f, 3.f, (float)i});
        } else if (MODE == 1) {
            float* p = base + wave * 4096 + lane;
#pragma unroll
            for (int i = 0; i < 64; ++i) st<NT>(p + i * 64, (float)i);
        } else {
            const int pt = lane & 15, q = lane >> 4;
            float* t = base + (wave >> 1) * tile_f;
            if (MODE == 2) {
                const int hsel = wave & 1;
#pragma unroll
                for (int nb = 0; nb < 16; ++nb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) st<NT>(t + (16 * nb + 4 * q + j) * 32 + 16 * hsel + pt, (float)j);
            } else {
                // mode 3: wave w writes features [128*(w&1), +128) of tile w/2, both point halves
#pragma unroll
                for (int nb = 0; nb < 8; ++nb)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int h = 0; h < 2; ++h) st<NT>(t + (128 * (wave & 1) + 16 * nb + 4 * q + j) * 32 + 16 * h + pt, (float)j);
            }
        }
    }
}
extern "C" int probe_write(int mode, int nt, void* dst, size_t bytes, int blocks, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const size_t nf = bytes / 4;
#define L(M, N) hipLaunchKernelGGL((write_k<M, N>), dim3(blocks), dim3(512), 0, s, (float*)dst, nf)
    switch (mode * 2 + (nt ? 1 : 0)) {
        case 0: L(0, false); break; case 1: L(0, true); break;
        case 2: L(1, false); break; case 3: L(1, true); break;
        case 4: L(2, false); break; case 5: L(2, true); break;
        case 6: L(3, false); break; case 7: L(3, true); break;
    }
    return (int)hipGetLastError();
}
