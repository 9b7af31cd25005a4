// One call per direction: render_rays (run_nerf.py:308-418) and its backward as TWO entry points that chain the launchers
// of this library in C, for hosts that do not want to sequence ~12 kernels and track scratch layouts themselves
// (declared in include/nerf_hip.h, section "render_rays in one call").  The Python binding's own path
// (render.py: _field_pass / _RenderRays.backward) issues the same launches in the same order, so the two produce
// bit-identical results (tests/test_gpu_round3.py::test_one_call_abi_*).
#include <hip/hip_runtime.h>
#include "nerf_common.h"
#include "api_util.h"
#include "launchers.h"

using namespace nerf_api;

namespace {

inline size_t up4(size_t n) { return (n + 3) & ~(size_t)3; }

// scratch layout of one render_rays call (floats; every region 16-byte aligned)
struct RenderWs {
    size_t t_lin, u_lin;            // torch.linspace(0, 1, n_coarse) / (0, 1, n_fine)
    size_t z_c, raw_c, w_c;         // coarse pass: depths, raw, compositing weights
    size_t z_f;                     // sorted union of coarse and fine depths
    size_t act_c, act_f, d_raw, delta, partial;     // training only
    size_t zero_rgb;                // training only: [N][3] zeros, the d_rgb of a pass that receives only d_disp / d_acc
    size_t total;
};
RenderWs render_ws(int n_rays, int Sc, int Sf, int training, int precision) {
    const int dp = precision == 0 ? 0 : (precision == 5 ? 2 : 1);      // scratch regions sized for the configured datapath (16-bit tiles on the split ones, hi + lo on 5)
    RenderWs w{};
    const size_t N = (size_t)n_rays;
    const int S2 = Sc + Sf;
    size_t o = 0;
    w.t_lin = o; o += up4((size_t)Sc);
    w.u_lin = o; o += up4((size_t)(Sf > 0 ? Sf : 1));
    w.z_c = o;   o += up4(N * Sc);
    w.raw_c = o; o += up4(N * Sc * 4);
    w.w_c = o;   o += up4(N * Sc);
    w.z_f = o;   o += up4(Sf > 0 ? N * S2 : 0);
    if (training) {
        w.act_c = o; o += up4(nerf_act_floats_dp(n_rays, Sc, dp));
        w.act_f = o; o += up4(Sf > 0 ? nerf_act_floats_dp(n_rays, S2, dp) : 0);
        w.d_raw = o; o += up4(N * (size_t)S2 * 4);
        w.delta = o; o += up4(nerf_delta_floats_dp(n_rays, S2, dp));
        w.partial = o; o += up4(nerf_wgrad_partial_floats(n_rays, S2));
        w.zero_rgb = o; o += up4(N * 3);
    }
    w.total = o;
    return w;
}

// torch.linspace(0, 1, n) in fp32, bit for bit (ATen RangeFactories: start + step * i for the first half,
// end - step * (n - 1 - i) for the second, step = (end - start) / (n - 1) -- each evaluated as ONE fused multiply-add, on the
// CPU and on the GPU alike; checked against torch for n = 37 .. 192, tests/test_gpu_round3.py)
__global__ void linspace01_kernel(float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (n == 1) { out[0] = 0.0f; return; }
    const float step = (1.0f - 0.0f) / (float)(n - 1);
    out[i] = i < n / 2 ? fmaf(step, (float)i, 0.0f) : fmaf(-step, (float)(n - 1 - i), 1.0f);
}
__global__ void add_inplace_kernel(float* dst, const float* src, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

bool cfg_ok(const NerfRenderCfg* c) {
    return c && c->n_coarse >= 3 && c->n_fine >= 0 && c->n_coarse + c->n_fine <= 4096 && (c->precision == 0 || c->precision == 1 || c->precision == 3 || c->precision == 5) &&
           c->raw_noise_std >= 0.0f;
}

// field evaluation of one pass in the configured datapath; act == nullptr: inference
hipError_t field_forward(const NerfRenderCfg* c, const float* packed, const float* rays, int stride, const float* z, int n, int S,
                         float* raw, float* act, hipStream_t st) {
    if (c->precision == 0) {
        if (act) tag_record(act, 0, ACT_ROWS_F32, n, S);
        return nerf::launch_field_fwd(packed, rays, stride, z, n, S, raw, act, st);
    }
    if (c->precision == 5) {        // fp16 split, two-word saves
        if (act) tag_record(act, 0, ACT_TILE16_F16X2, n, S);
        return nerf::launch_field_fwd16r(packed, rays, stride, z, n, S, raw, act, act ? 5 : 1, st);
    }
    if (c->precision == 3) {        // fp16 split: 16-bit (fp16) rows always
        if (act) tag_record(act, 0, ACT_TILE16_F16, n, S);
        return nerf::launch_field_fwd16r(packed, rays, stride, z, n, S, raw, act, 1, st);
    }
    if (act) tag_record(act, 0, ACT_TILE16_BF16, n, S);
    return nerf::launch_field_fwd16r(packed, rays, stride, z, n, S, raw, act, 0, st);
}

// parameter gradient of one pass: dgrad + weight gradients into `grad`
hipError_t field_backward(const NerfRenderCfg* c, const float* packed, const float* params, const float* act, const float* d_raw, int n, int S,
                          float* delta, float* partial, float* grad, int accumulate, hipStream_t st) {
    if (c->precision == 0) return nerf::launch_field_bwd(packed, act, d_raw, n, S, delta, partial, grad, accumulate, st);
    hipError_t e;
    if (c->precision == 5) {
        e = nerf::launch_field_dgrad3r(packed, act, d_raw, n, S, delta, 5, st);
        if (e != hipSuccess) return e;
        tag_record(delta, 1, DELTA_TILE32_F16X2, n, S);
        return nerf::launch_field_wgrad(act, delta, d_raw, n, S, partial, grad, accumulate, 6, 7, st, params);
    }
    if (c->precision == 3) {
        e = nerf::launch_field_dgrad3r(packed, act, d_raw, n, S, delta, 1, st);
        if (e != hipSuccess) return e;
        tag_record(delta, 1, DELTA_TILE32_F16, n, S);
        return nerf::launch_field_wgrad(act, delta, d_raw, n, S, partial, grad, accumulate, 5, 7, st, params);
    }
    e = nerf::launch_field_dgrad3r(packed, act, d_raw, n, S, delta, 0, st);
    if (e != hipSuccess) return e;
    tag_record(delta, 1, DELTA_TILE32_BF16, n, S);
    return nerf::launch_field_wgrad(act, delta, d_raw, n, S, partial, grad, accumulate, 4, 7, st, params);
}

}  // namespace

extern "C" {

size_t nerf_render_workspace_floats(const NerfRenderCfg* cfg, int n_rays, int training) {
    if (!cfg_ok(cfg) || n_rays <= 0) return 0;
    return render_ws(n_rays, cfg->n_coarse, cfg->n_fine, training, cfg->precision).total;
}

int nerf_render_rays_fwd(const NerfRenderCfg* cfg, const float* packed_c, const float* packed_f, const float* rays, int ray_stride,
                         int n_rays, const float* t_rand, const float* noise_c, const float* u, const float* noise_f,
                         float* rgb, float* disp, float* acc, float* raw, float* rgb0, float* disp0, float* acc0, float* z_std,
                         float* workspace, int training, void* stream) {
    REQUIRE(cfg_ok(cfg), "bad NerfRenderCfg (n_coarse >= 3, n_fine >= 0, precision 0 / 1 / 3 / 5, raw_noise_std >= 0)");
    REQUIRE(packed_c && rays && rgb && disp && acc && raw && workspace, "null pointer");
    REQUIRE(ray_stride >= 11, "rays must carry view directions (ray_stride >= 11): use_viewdirs=True architecture");
    REQUIRE(n_rays >= 0, "bad size");
    const int Sc = cfg->n_coarse, Sf = cfg->n_fine, S2 = Sc + Sf;
    const bool fine = Sf > 0;
    REQUIRE(!fine || (rgb0 && disp0 && acc0 && z_std), "n_fine > 0 needs the coarse outputs rgb0 / disp0 / acc0 and z_std");
    REQUIRE(!(cfg->raw_noise_std > 0.0f) || (noise_c && (!fine || noise_f)), "raw_noise_std > 0 needs the noise draws");
    REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0 && (reinterpret_cast<uintptr_t>(raw) & 15) == 0 &&
            (reinterpret_cast<uintptr_t>(packed_c) & 15) == 0, "workspace / raw / packed must be 16-byte aligned");
    if (n_rays == 0) return 0;
    const float* pf = (fine && packed_f) ? packed_f : packed_c;          // network_fine == None: the coarse network (run_nerf.py:400)
    const RenderWs w = render_ws(n_rays, Sc, Sf, training, cfg->precision);
    hipStream_t st = (hipStream_t)stream;
    float* ws = workspace;
    hipLaunchKernelGGL(linspace01_kernel, dim3((Sc + 255) / 256), dim3(256), 0, st, ws + w.t_lin, Sc);
    if (fine) hipLaunchKernelGGL(linspace01_kernel, dim3((Sf + 255) / 256), dim3(256), 0, st, ws + w.u_lin, Sf);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return done(__func__, e);
    // coarse pass (run_nerf.py:357-386)
    e = nerf::launch_sample_coarse(rays, ray_stride, n_rays, ws + w.t_lin, Sc, cfg->lindisp, t_rand, ws + w.z_c, st);
    if (e != hipSuccess) return done(__func__, e);
    float* raw_c = fine ? ws + w.raw_c : raw;
    e = field_forward(cfg, packed_c, rays, ray_stride, ws + w.z_c, n_rays, Sc, raw_c, training ? ws + w.act_c : nullptr, st);
    if (e != hipSuccess) return done(__func__, e);
    {
        nerf::CompositeArgs a{raw_c, ws + w.z_c, rays + 3, cfg->raw_noise_std > 0.0f ? noise_c : nullptr, cfg->raw_noise_std,
                              ray_stride, n_rays, Sc, cfg->white_bkgd,
                              fine ? rgb0 : rgb, fine ? disp0 : disp, fine ? acc0 : acc, fine ? ws + w.w_c : nullptr, nullptr,
                              nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        e = nerf::launch_composite(a, false, st);
        if (e != hipSuccess) return done(__func__, e);
    }
    if (!fine) return 0;
    // hierarchical pass (run_nerf.py:388-412)
    {
        nerf::FineArgs a{ws + w.z_c, ws + w.w_c, u, u ? nullptr : ws + w.u_lin, ws + w.z_f, nullptr, z_std, n_rays, Sc, Sf, 0};
        e = nerf::launch_sample_fine(a, st);
        if (e != hipSuccess) return done(__func__, e);
    }
    e = field_forward(cfg, pf, rays, ray_stride, ws + w.z_f, n_rays, S2, raw, training ? ws + w.act_f : nullptr, st);
    if (e != hipSuccess) return done(__func__, e);
    nerf::CompositeArgs a{raw, ws + w.z_f, rays + 3, cfg->raw_noise_std > 0.0f ? noise_f : nullptr, cfg->raw_noise_std,
                          ray_stride, n_rays, S2, cfg->white_bkgd, rgb, disp, acc, nullptr, nullptr,
                          nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    return done(__func__, nerf::launch_composite(a, false, st));
}

int nerf_render_infer_supported(const NerfRenderCfg* cfg) {
    return cfg_ok(cfg) && cfg->precision != 0 && nerf::render_infer_fused_ok(cfg->n_coarse, cfg->n_fine) ? 1 : 0;
}

int nerf_render_rays_infer(const NerfRenderCfg* cfg, const float* packed_c, const float* packed_f, const float* rays, int ray_stride,
                           int n_rays, const float* t_rand, const float* noise_c, const float* u, const float* noise_f,
                           float* rgb, float* disp, float* acc, float* raw, float* rgb0, float* disp0, float* acc0, float* z_std,
                           float* workspace, void* stream) {
    REQUIRE(cfg_ok(cfg), "bad NerfRenderCfg (n_coarse >= 3, n_fine >= 0, precision 0 / 1 / 3 / 5, raw_noise_std >= 0)");
    REQUIRE(nerf_render_infer_supported(cfg), "one-launch inference: a three-term split datapath (precision 1 / 3), 16 * n_coarse and 16 * (n_coarse + n_fine) "
            "multiples of 128, n_coarse + n_fine <= 1024 (use nerf_render_rays_fwd(training = 0) otherwise)");
    REQUIRE(packed_c && rays && rgb && disp && acc && raw && workspace, "null pointer");
    REQUIRE(ray_stride >= 11, "rays must carry view directions (ray_stride >= 11): use_viewdirs=True architecture");
    REQUIRE(n_rays >= 0, "bad size");
    const int Sc = cfg->n_coarse, Sf = cfg->n_fine;
    const bool fine = Sf > 0;
    REQUIRE(!fine || (rgb0 && disp0 && acc0 && z_std), "n_fine > 0 needs the coarse outputs rgb0 / disp0 / acc0 and z_std");
    REQUIRE(!(cfg->raw_noise_std > 0.0f) || (noise_c && (!fine || noise_f)), "raw_noise_std > 0 needs the noise draws");
    REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0 && (reinterpret_cast<uintptr_t>(raw) & 15) == 0 &&
            (reinterpret_cast<uintptr_t>(packed_c) & 15) == 0, "workspace / raw / packed must be 16-byte aligned");
    if (n_rays == 0) return 0;
    const RenderWs w = render_ws(n_rays, Sc, Sf, 0, cfg->precision);
    float* ws = workspace;
    const bool noisy = cfg->raw_noise_std > 0.0f;
    nerf::RenderInferArgs a{};
    a.packed_c = packed_c;
    a.packed_f = (fine && packed_f) ? packed_f : packed_c;          // network_fine == None: the coarse network (run_nerf.py:400)
    a.rays = rays; a.ray_stride = ray_stride; a.n_rays = n_rays; a.n_c = Sc; a.n_f = Sf;
    a.lindisp = cfg->lindisp; a.white_bkgd = cfg->white_bkgd; a.noise_std = cfg->raw_noise_std;
    a.t_rand = t_rand; a.noise_c = noisy ? noise_c : nullptr; a.u = u; a.noise_f = noisy ? noise_f : nullptr;
    a.z_c = ws + w.z_c; a.w_c = ws + w.w_c;
    a.raw_c = fine ? ws + w.raw_c : raw;
    a.rgb_c = fine ? rgb0 : rgb; a.disp_c = fine ? disp0 : disp; a.acc_c = fine ? acc0 : acc;
    a.z_f = ws + w.z_f; a.z_std = z_std; a.raw_f = raw; a.rgb_f = rgb; a.disp_f = disp; a.acc_f = acc;
    a.split = (cfg->precision == 3 || cfg->precision == 5) ? 1 : 0;
    return done(__func__, nerf::launch_render_infer(a, (hipStream_t)stream));
}

int nerf_render_rays_bwd(const NerfRenderCfg* cfg, const float* packed_c, const float* packed_f, const float* params_c,
                         const float* params_f, const float* rays, int ray_stride, int n_rays, const float* noise_c,
                         const float* noise_f, const float* raw, const float* d_rgb, const float* d_disp, const float* d_acc,
                         const float* d_raw_out, const float* d_rgb0, const float* d_disp0, const float* d_acc0,
                         float* workspace, float* grad_c, float* grad_f, int accumulate, void* stream) {
    REQUIRE(cfg_ok(cfg), "bad NerfRenderCfg");
    REQUIRE(packed_c && rays && raw && workspace && grad_c, "null pointer");
    REQUIRE(cfg->precision == 0 || params_c, "the split datapaths need the canonical parameters (folded feature layer)");
    REQUIRE(ray_stride >= 11 && n_rays >= 0, "bad size");
    REQUIRE(!(cfg->raw_noise_std > 0.0f) || noise_c, "raw_noise_std > 0 needs the noise draws of the forward");
    if (n_rays == 0) return 0;
    const int Sc = cfg->n_coarse, Sf = cfg->n_fine, S2 = Sc + Sf;
    const bool fine = Sf > 0;
    const bool same_net = !fine || !packed_f || packed_f == packed_c;
    REQUIRE(same_net || (grad_f && (cfg->precision == 0 || params_f)), "a separate fine network needs grad_f (and params_f)");
    const RenderWs w = render_ws(n_rays, Sc, Sf, 1, cfg->precision);
    hipStream_t st = (hipStream_t)stream;
    float* ws = workspace;
    hipError_t e;
    // one pass: upstream (d_rgb, d_disp, d_acc[, d_raw]) -> d_raw (raw2outputs' adjoint, run_nerf.py:262-305) -> parameters
    auto pass = [&](const float* packed, const float* params, const float* act, const float* raw_p, const float* z, const float* noise,
                    int S, const float* g_rgb, const float* g_disp, const float* g_acc, const float* g_raw, float* grad, int accum) -> hipError_t {
        float* d_raw = ws + w.d_raw;
        const size_t n4 = (size_t)n_rays * S * 4;
        if (g_rgb || g_disp || g_acc) {
            if (!g_rgb) {       // d_disp / d_acc without d_rgb: a zero d_rgb (its own region of the workspace)
                hipError_t err = hipMemsetAsync(ws + w.zero_rgb, 0, (size_t)n_rays * 3 * sizeof(float), st);
                if (err != hipSuccess) return err;
                g_rgb = ws + w.zero_rgb;
            }
            nerf::CompositeArgs a{raw_p, z, rays + 3, cfg->raw_noise_std > 0.0f ? noise : nullptr, cfg->raw_noise_std, ray_stride, n_rays, S,
                                  cfg->white_bkgd, nullptr, nullptr, nullptr, nullptr, nullptr, g_rgb, g_acc, g_disp, d_raw, nullptr, nullptr};
            hipError_t err = nerf::launch_composite(a, true, st);
            if (err != hipSuccess) return err;
            if (g_raw) {
                hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, d_raw, g_raw, n4);
                err = hipGetLastError();
                if (err != hipSuccess) return err;
            }
        } else {
            hipError_t err = hipMemcpyAsync(d_raw, g_raw, n4 * sizeof(float), hipMemcpyDeviceToDevice, st);
            if (err != hipSuccess) return err;
        }
        return field_backward(cfg, packed, params, act, d_raw, n_rays, S, ws + w.delta, ws + w.partial, grad, accum, st);
    };
    bool wrote_c = false;
    if (fine) {
        const bool up_c = d_rgb0 || d_disp0 || d_acc0, up_f = d_rgb || d_disp || d_acc || d_raw_out;
        REQUIRE(up_c || up_f, "no upstream gradient");
        if (up_c) {
            e = pass(packed_c, params_c, ws + w.act_c, ws + w.raw_c, ws + w.z_c, noise_c, Sc, d_rgb0, d_disp0, d_acc0, nullptr, grad_c, accumulate);
            if (e != hipSuccess) return done(__func__, e);
            wrote_c = true;
        }
        if (up_f) {
            e = pass(same_net ? packed_c : packed_f, same_net ? params_c : params_f, ws + w.act_f, raw, ws + w.z_f, noise_f, S2, d_rgb, d_disp,
                     d_acc, d_raw_out, same_net ? grad_c : grad_f, same_net ? (accumulate || wrote_c) : accumulate);
            if (e != hipSuccess) return done(__func__, e);
        }
        // accumulate == 0 promises that every gradient vector is WRITTEN: a network whose pass has no upstream gradient gets zeros
        if (!accumulate) {
            const size_t gbytes = (size_t)nerf::N_PARAMS * sizeof(float);
            if (!up_c && !same_net) { e = hipMemsetAsync(grad_c, 0, gbytes, st); if (e != hipSuccess) return done(__func__, e); }
            if (!up_f && !same_net) { e = hipMemsetAsync(grad_f, 0, gbytes, st); if (e != hipSuccess) return done(__func__, e); }
        }
        return 0;
    }
    REQUIRE(d_rgb || d_disp || d_acc || d_raw_out, "no upstream gradient");
    return done(__func__, pass(packed_c, params_c, ws + w.act_c, raw, ws + w.z_c, noise_c, Sc, d_rgb, d_disp, d_acc, d_raw_out, grad_c, accumulate));
}

}  // extern "C"
