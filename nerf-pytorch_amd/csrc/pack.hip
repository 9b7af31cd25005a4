// Repack canonical (state_dict-order) parameters into the MFMA A-fragment
// streams consumed by the field kernels (layout: nerf_common.h).
// Replaces nothing in the reference: it is the price of keeping nn.Module
// parameters in PyTorch layout (run_nerf_helpers.py:79-94) while the kernels
// read lane-linear fragments.  ~1.2 M elements, one launch per optimizer step.
#include <hip/hip_runtime.h>
#include "nerf_common.h"

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
