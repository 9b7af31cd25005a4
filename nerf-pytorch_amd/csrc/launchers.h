// Host-side launch interface between api.hip and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace nerf {

struct CompositeArgs {
    const float* raw;       // [N][S][4]
    const float* z;         // [N][S]
    const float* dirs;      // ray directions: dirs[r*dir_stride + 0..2]
    const float* noise;     // [N][S] standard normal, nullable
    float noise_std;
    int dir_stride, n_rays, S, white_bkgd;
    // forward outputs (weights / depth nullable)
    float* rgb; float* disp; float* acc; float* weights; float* depth;
    // backward inputs / output
    const float* d_rgb;     // [N][3]
    const float* d_acc;     // [N] nullable
    const float* d_disp;    // [N] nullable
    float* d_raw;           // [N][S][4]
    const float* d_weights; // [N][S] nullable: upstream gradient of the `weights` output
    const float* d_depth;   // [N] nullable: upstream gradient of `depth_map`
};

struct FineArgs {
    // direct == 0: in0 = coarse depths z[N][Sc], in1 = coarse weights[N][Sc]
    //              bins = mid-points, pdf from weights[1:-1]               (run_nerf.py:392-393)
    // direct == 1: in0 = bins[N][nb], in1 = weights[N][nb-1]               (sample_pdf as called standalone)
    const float* in0; const float* in1;
    const float* u;         // [N][Nf] uniform draws, nullable -> u_lin
    const float* u_lin;     // [Nf] torch.linspace(0,1,Nf)
    float* z_all;           // [N][Sc+Nf] sorted union (direct == 0 only), nullable
    float* z_samples;       // [N][Nf] nullable
    float* z_std;           // [N] nullable
    int n_rays, n_in, Nf, direct;   // n_in = Sc (direct == 0) or nb (direct == 1)
};

// render_rays without gradients in one launch (render_fused.hip)
struct RenderInferArgs {
    const float* packed_c;  // packed3 of the coarse network
    const float* packed_f;  // packed3 of the fine network (== packed_c when there is none)
    const float* rays;
    int ray_stride, n_rays, n_c, n_f, lindisp, white_bkgd;
    float noise_std;
    const float* t_rand;    // [N][n_c] or null
    const float* noise_c;   // [N][n_c] or null
    const float* u;         // [N][n_f] or null (deterministic: linspace)
    const float* noise_f;   // [N][n_c + n_f] or null
    // coarse pass (the only pass when n_f == 0: then these ARE the outputs)
    float *z_c, *raw_c, *w_c, *rgb_c, *disp_c, *acc_c;
    // fine pass
    float *z_f, *z_std, *raw_f, *rgb_f, *disp_f, *acc_f;
    int split;              // 16-bit type of the three-term split the packed buffers were made for: 0 bf16, 1 fp16
};
bool render_infer_fused_ok(int n_c, int n_f);
hipError_t launch_render_infer(const RenderInferArgs& a, hipStream_t stream);

hipError_t launch_pack(const float* canon_params, float* packed, hipStream_t stream);
hipError_t launch_sample_coarse(const float* rays, int ray_stride, int n_rays, const float* t_vals, int S,
                                int lindisp, const float* t_rand, float* z_out, hipStream_t stream);
constexpr int MSE_SCRATCH_FLOATS = 256 + 1;      // per-block partial sums + the ticket word of mse_fwd_kernel
// max |fp16| pattern over packed words (range check of the fp16 split's saved rows / deltas): words[1] = max pattern, words[0] |= (>= 0x7800)
hipError_t launch_range_scan(const unsigned* rows, size_t n_words, unsigned* words, hipStream_t stream);
hipError_t launch_mse_fwd(const float* x, const float* y, long n, float* scratch, float* out, hipStream_t stream);
hipError_t launch_mse_bwd(const float* x, const float* y, long n, const float* g, float* dx, hipStream_t stream);
hipError_t launch_embed(const float* x, long n_pts, int n_freqs, float* out, hipStream_t stream);
hipError_t launch_make_rays(int H, int W, const float* K9, const float* pose12, const float* pose_static12, int ndc,
                            float near, float far, float* rays, int ray_stride, hipStream_t stream);
hipError_t launch_assemble_rays(const float* rays_o, const float* rays_d, long n, int ndc, int H, int W, float focal,
                                float near, float far, float* rays, int ray_stride, hipStream_t stream);
hipError_t launch_sample_ray_batch(int H, int W, const float* K9, const float* pose_dev, int pose_stride, const float* image, int h0, int w0,
                                   int nh, int nw, int n_rand, unsigned key0, unsigned key1, float* rays, float* target, int* pixels,
                                   hipStream_t stream);
hipError_t launch_composite(const CompositeArgs& a, bool bwd, hipStream_t stream);
hipError_t launch_sample_fine(const FineArgs& a, hipStream_t stream);
hipError_t launch_field_fwd(const float* packed, const float* rays, int ray_stride, const float* z_vals,
                            int n_rays, int S, float* raw, float* act, hipStream_t stream);
hipError_t launch_field_bwd(const float* packed, const float* act, const float* d_raw, int n_rays, int S,
                            float* delta, float* partial, float* grad, int accumulate, hipStream_t stream);
hipError_t launch_field_dgrad(const float* packed, const float* act, const float* d_raw, int n_rays, int S,
                              float* delta, hipStream_t stream);
hipError_t launch_field_wgrad(const float* act, const float* delta, const float* d_raw, int n_rays, int S,
                              float* partial, float* grad, int accumulate, int datapath /* 0 fp32, 4 bf16 operands, 5 fp16 operands */,
                              int phases, hipStream_t stream, const float* params);     // canonical parameters: required by the split datapaths
size_t wgrad_partial_floats(long P);
hipError_t launch_adam(float* p, const float* g, float* m, float* v, int n, float lr, float b1, float b2, float eps, int step,
                       hipStream_t stream);
void pack_table_host(int* out);
void pack3_table_host(int* out);
void pack16_table_host(int* out);
hipError_t launch_pack3_sel(const float* canon_params, float* packed, int streams, hipStream_t stream, int split = 0);
hipError_t launch_pack3_pair(const float* params_a, float* packed_a, const float* params_b, float* packed_b, int streams, hipStream_t stream, int split);
// largest launches of the 16-point forward: n_rays * S points in all, and with saving (4.8 KB per point: 2^26 points = 320 GB)
constexpr long FWD16R_MAX_POINTS = (1L << 31) - 1, FWD16R_MAX_SAVED_POINTS = 1L << 26;
// split (ring kernels, repack, streaming weight-gradient GEMM): 0 = bf16 three-term split, 1 = fp16 (csrc/split_types.h)
hipError_t launch_field_fwd16r(const float* packed3, const float* rays, int ray_stride, const float* z_vals,
                               int n_rays, int S, float* raw, float* act, int split, hipStream_t stream);
hipError_t launch_field_fwd16r_last(const float* packed3, const float* rays, int ray_stride, const float* z_vals,
                                    int n_rays, int S, float* raw, const float* packed3_next, float* raw_next, int S_next,
                                    hipStream_t stream);
hipError_t launch_field_dgrad3r(const float* packed3, const float* act, const float* d_raw, int n_rays, int S,
                                float* delta, int split, hipStream_t stream);

}  // namespace nerf
