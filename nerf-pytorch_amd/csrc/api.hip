// extern "C" entry points of libnerf_hip.so (declared in include/nerf_hip.h).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <iterator>
#include <mutex>
#include <unordered_map>
#include "nerf_common.h"
#include "api_util.h"

#include "launchers.h"

using nerf_api::done;
using nerf_api::fail_arg;

namespace nerf_api {

static std::mutex g_tag_mutex;
static std::unordered_map<const void*, BufTag> g_tags;

static unsigned long g_tag_seq = 0;

void tag_record(const void* buf, int is_delta, int kind, int n_rays, int n_samples) {
    if (!buf) return;
    std::lock_guard<std::mutex> lock(g_tag_mutex);
    // a training loop re-uses a handful of buffers; the bound only limits leaks.  Eviction is by AGE (the 2048 most recently
    // written records always survive): the record of an `act` whose backward is still pending is never dropped by a burst
    // of unrelated writes, as a wholesale clear() would.
    if (g_tags.size() > 4096)
        for (auto it = g_tags.begin(); it != g_tags.end();)
            it = (g_tag_seq - it->second.seq > 2048) ? g_tags.erase(it) : std::next(it);
    g_tags[buf] = BufTag{is_delta, kind, n_rays, n_samples, ++g_tag_seq};
}
bool tag_lookup(const void* buf, BufTag* out) {
    std::lock_guard<std::mutex> lock(g_tag_mutex);
    auto it = g_tags.find(buf);
    if (it == g_tags.end()) return false;
    *out = it->second;
    return true;
}
int datapath_of(int act_kind, int delta_kind) {
    if (act_kind == ACT_ROWS_F32 && delta_kind == DELTA_ROWS_F32) return 0;
    if (act_kind == ACT_TILE16_BF16 && delta_kind == DELTA_TILE32_BF16) return 4;
    if (act_kind == ACT_TILE16_F16 && delta_kind == DELTA_TILE32_F16) return 5;
    if (act_kind == ACT_TILE16_F16X2 && delta_kind == DELTA_TILE32_F16X2) return 6;
    return -1;
}

// the save buffer a dgrad is about to read: written by a forward of the same family (fp32 rows vs split-bf16 tiles: the
// ReLU bitmasks differ) for the same point count?  0 = fine / unknown buffer
static int check_act_for_dgrad(const char* fn, const void* act, bool split_bf16, int n_rays, int n_samples, int two_word = -1) {
    BufTag t;
    if (!tag_lookup(act, &t)) return 0;
    if (t.is_delta) return fail_arg(fn, "`act` is a buffer this library last wrote DELTAS into");
    if (two_word >= 0 && (t.kind == ACT_TILE16_F16X2) != (two_word == 1))
        return fail_arg(fn, "`act` and this dgrad disagree about two-word saves (split = 5 on both the forward and the dgrad, or on neither): "
                            "the three-term weight-gradient GEMM contracts the lo deltas with the lo rows only a split = 5 forward saved");
    if ((t.kind != ACT_ROWS_F32) != split_bf16)
        return fail_arg(fn, split_bf16 ? "`act` was saved by the exact-fp32 forward (point-major rows, fp32 bitmask order): not readable by a split datapath's dgrad"
                                       : "`act` was saved by a split datapath's forward (tiles, its bitmask order): not readable by the fp32 dgrad");
    if (t.n_rays != n_rays || t.n_samples != n_samples) {
        snprintf(g_err, sizeof(g_err), "%s: `act` was saved for %d rays x %d samples, this call says %d x %d", fn, t.n_rays, t.n_samples, n_rays, n_samples);
        return NERF_E_BADARG;
    }
    return 0;
}

}  // namespace nerf_api
using namespace nerf_api;

extern "C" {

int nerf_abi_version(void) { return NERF_ABI_VERSION; }
const char* nerf_last_error(void) { return g_err; }
int nerf_param_count(void) { return nerf::N_PARAMS; }
int nerf_packed_floats(void) { return nerf::PACKED_FLOATS; }

int nerf_param_offset(int idx, int* rows, int* cols) {
    constexpr nerf::Canon c = nerf::canon();
    int off = -1, r = 0, cc = 1;
    if (idx >= 0 && idx < 2 * nerf::D) {
        const int l = idx / 2;
        if (idx % 2 == 0) { off = c.w[l]; r = nerf::W; cc = nerf::fan_in(l); } else { off = c.b[l]; r = nerf::W; }
    } else switch (idx) {
        case 16: off = c.wv; r = nerf::WV; cc = nerf::W + nerf::IN_DIR; break;
        case 17: off = c.bv; r = nerf::WV; break;
        case 18: off = c.wf; r = nerf::W; cc = nerf::W; break;
        case 19: off = c.bf; r = nerf::W; break;
        case 20: off = c.wa; r = 1; cc = nerf::W; break;
        case 21: off = c.ba; r = 1; break;
        case 22: off = c.wr; r = 3; cc = nerf::WV; break;
        case 23: off = c.br; r = 3; break;
        default: break;
    }
    if (rows) *rows = r;
    if (cols) *cols = cc;
    return off;
}

int nerf_debug_pack_table(int* out_host) {
    REQUIRE(out_host, "null pointer");
    nerf::pack_table_host(out_host);
    return 0;
}

int nerf_pack_params(const float* params, float* packed, void* stream) {
    REQUIRE(params && packed, "null pointer");
    return done(__func__, nerf::launch_pack(params, packed, (hipStream_t)stream));
}

int nerf_embed(const float* x, long n_pts, int n_freqs, float* out, void* stream) {
    REQUIRE(x && out, "null pointer");
    REQUIRE(n_pts >= 0 && n_freqs >= 0 && n_freqs <= 30, "bad size");
    return done(__func__, nerf::launch_embed(x, n_pts, n_freqs, out, (hipStream_t)stream));
}

int nerf_make_rays(int H, int W, const float* K_host, const float* c2w_host, const float* c2w_staticcam_host, int ndc,
                   float near, float far, float* rays, int ray_stride, void* stream) {
    REQUIRE(K_host && c2w_host && rays, "null pointer");
    REQUIRE(H >= 0 && W >= 0 && ray_stride >= 11, "bad size");
    REQUIRE(K_host[0] != 0.0f && K_host[4] != 0.0f, "K has a zero focal length");
    return done(__func__, nerf::launch_make_rays(H, W, K_host, c2w_host, c2w_staticcam_host, ndc, near, far, rays,
                                                 ray_stride, (hipStream_t)stream));
}

int nerf_assemble_rays(const float* rays_o, const float* rays_d, long n_rays, int ndc, int H, int W, float focal, float near,
                       float far, float* rays, int ray_stride, void* stream) {
    REQUIRE(rays_o && rays_d && rays, "null pointer");
    REQUIRE(n_rays >= 0 && ray_stride >= 11, "bad size");
    REQUIRE(!ndc || (H > 0 && W > 0 && focal != 0.0f), "ndc needs H, W and a non-zero focal length");
    return done(__func__, nerf::launch_assemble_rays(rays_o, rays_d, n_rays, ndc, H, W, focal, near, far, rays, ray_stride,
                                                     (hipStream_t)stream));
}

int nerf_sample_ray_batch(int H, int W, const float* K_host, const float* pose, int pose_row_stride, const float* image, int h0,
                          int w0, int nh, int nw, int n_rand, unsigned key0, unsigned key1, float* batch_rays, float* target,
                          int* pixels, void* stream) {
    REQUIRE(K_host && pose && image && batch_rays && target, "null pointer");
    REQUIRE(H > 0 && W > 0 && nh > 0 && nw > 0 && h0 >= 0 && w0 >= 0 && h0 + nh <= H && w0 + nw <= W, "the crop window must lie inside the image");
    REQUIRE((long)H * W < (1L << 31), "image too large");
    REQUIRE(n_rand >= 0 && (long)n_rand <= (long)nh * nw, "cannot take more distinct pixels than the window holds (np.random.choice(..., replace=False) raises too)");
    REQUIRE(pose_row_stride >= 4, "pose rows are at least 4 floats apart");
    REQUIRE(K_host[0] != 0.0f && K_host[4] != 0.0f, "K has a zero focal length");
    return done(__func__, nerf::launch_sample_ray_batch(H, W, K_host, pose, pose_row_stride, image, h0, w0, nh, nw, n_rand, key0, key1,
                                                        batch_rays, target, pixels, (hipStream_t)stream));
}

int nerf_sample_coarse(const float* rays, int ray_stride, int n_rays, const float* t_vals, int n_samples,
                       int lindisp, const float* t_rand, float* z_vals, void* stream) {
    REQUIRE(rays && t_vals && z_vals, "null pointer");
    REQUIRE(ray_stride >= 8 && n_rays >= 0 && n_samples >= 1, "bad size");
    return done(__func__, nerf::launch_sample_coarse(rays, ray_stride, n_rays, t_vals, n_samples, lindisp, t_rand,
                                                     z_vals, (hipStream_t)stream));
}

size_t nerf_act_floats_dp(int n_rays, int n_samples, int datapath) {
    if (n_rays <= 0 || n_samples <= 0 || datapath < 0 || datapath > 2) return 0;
    const size_t P = (size_t)n_rays * n_samples;
    return datapath == 0 ? nerf::act_layout(P, (size_t)n_rays).total : nerf::act_layout3(P, (size_t)n_rays, datapath == 2).total;
}
size_t nerf_delta_floats_dp(int n_rays, int n_samples, int datapath) {
    if (n_rays <= 0 || n_samples <= 0 || datapath < 0 || datapath > 2) return 0;
    const size_t P = (size_t)n_rays * n_samples;
    return datapath == 0 ? nerf::delta_layout(P).total : nerf::delta_layout3(P, datapath == 2).total;
}
size_t nerf_act_floats(int n_rays, int n_samples) {       // a buffer either datapath may write
    const size_t a = nerf_act_floats_dp(n_rays, n_samples, 0), b = nerf_act_floats_dp(n_rays, n_samples, 1);
    return a > b ? a : b;
}
size_t nerf_delta_floats(int n_rays, int n_samples) {
    const size_t a = nerf_delta_floats_dp(n_rays, n_samples, 0), b = nerf_delta_floats_dp(n_rays, n_samples, 1);
    return a > b ? a : b;
}
size_t nerf_wgrad_partial_floats(int n_rays, int n_samples) {
    if (n_rays <= 0 || n_samples <= 0) return 0;
    return nerf::wgrad_partial_floats((long)n_rays * n_samples);
}

size_t nerf_workspace_floats(int n_rays, int n_coarse, int n_fine, int training) {
    if (!training || n_rays <= 0 || n_coarse <= 0 || n_fine < 0) return 0;
    const int s_big = n_coarse + n_fine;
    return nerf_act_floats(n_rays, n_coarse) + (n_fine > 0 ? nerf_act_floats(n_rays, s_big) : 0) +
           nerf_delta_floats(n_rays, s_big) + nerf_wgrad_partial_floats(n_rays, s_big);
}
size_t nerf_workspace_floats_dp(int n_rays, int n_coarse, int n_fine, int training, int datapath) {
    if (!training || n_rays <= 0 || n_coarse <= 0 || n_fine < 0 || datapath < 0 || datapath > 2) return 0;
    const int s_big = n_coarse + n_fine;
    return nerf_act_floats_dp(n_rays, n_coarse, datapath) + (n_fine > 0 ? nerf_act_floats_dp(n_rays, s_big, datapath) : 0) +
           nerf_delta_floats_dp(n_rays, s_big, datapath) + nerf_wgrad_partial_floats(n_rays, s_big);
}

int nerf_debug_layout(int n_rays, int n_samples, int family, int is_delta, long long* out_host) {
    REQUIRE(out_host, "null pointer");
    REQUIRE(n_rays > 0 && n_samples > 0 && family >= 0 && family <= 2, "bad size / family (0 fp32 rows, 1 split tiles, 2 split tiles with two-word saves)");
    const size_t P = (size_t)n_rays * n_samples;
    for (int i = 0; i < 16; ++i) out_host[i] = -1;
    auto put = [&](int i, size_t v) { out_host[i] = (long long)v; };
    if (!is_delta && family == 0) {
        const nerf::ActLayout a = nerf::act_layout(P, (size_t)n_rays);
        for (int l = 0; l < nerf::D; ++l) put(l, a.h[l]);
        put(8, a.feat); put(9, a.hv); put(10, a.enc); put(11, a.dir); put(12, a.dir_pt); put(13, a.mask); put(14, a.total);
    } else if (!is_delta) {
        const nerf::ActLayout3 a = nerf::act_layout3(P, (size_t)n_rays, family == 2);
        for (int l = 0; l < nerf::D; ++l) put(l, a.h[l]);
        put(8, a.feat); put(9, a.hv); put(10, a.enc); put(11, a.dir); put(12, a.dir_pt); put(13, a.mask); put(14, a.total);
        if (family == 2) put(15, a.lo);
    } else if (family == 0) {
        const nerf::DeltaLayout a = nerf::delta_layout(P);
        for (int l = 0; l < nerf::D; ++l) put(l, a.h[l]);
        put(8, a.feat); put(9, a.hv); put(14, a.total);
    } else {
        const nerf::DeltaLayout3 a = nerf::delta_layout3(P, family == 2);
        for (int l = 0; l < nerf::D; ++l) put(l, a.h[l]);
        put(8, a.feat); put(9, a.hv); put(10, a.graw); put(11, a.scale); put(14, a.total);
        if (family == 2) put(15, a.lo);
    }
    return 0;
}

int nerf_buffer_layout(const float* buf, int* is_delta, int* n_rays, int* n_samples) {
    BufTag t;
    if (!buf || !tag_lookup(buf, &t)) return -1;
    if (is_delta) *is_delta = t.is_delta;
    if (n_rays) *n_rays = t.n_rays;
    if (n_samples) *n_samples = t.n_samples;
    return t.kind;
}

int nerf_field_fwd(const float* packed, const float* rays, int ray_stride, const float* z_vals, int n_rays,
                   int n_samples, float* raw, float* act, void* stream) {
    REQUIRE(packed && rays && z_vals && raw, "null pointer");
    REQUIRE(ray_stride >= 11, "rays must carry view directions (ray_stride >= 11): use_viewdirs=True architecture");
    REQUIRE(n_rays >= 0 && n_samples >= 1, "bad size");
    REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0 && (reinterpret_cast<uintptr_t>(raw) & 15) == 0 &&
            (reinterpret_cast<uintptr_t>(act) & 15) == 0, "packed/raw/act must be 16-byte aligned");
    if (act) tag_record(act, 0, ACT_ROWS_F32, n_rays, n_samples);
    return done(__func__, nerf::launch_field_fwd(packed, rays, ray_stride, z_vals, n_rays, n_samples, raw, act,
                                                 (hipStream_t)stream));
}

int nerf_raw2outputs(const float* raw, const float* z_vals, const float* rays_d, int dir_stride, int n_rays,
                     int n_samples, const float* noise, float raw_noise_std, int white_bkgd, float* rgb_map,
                     float* disp_map, float* acc_map, float* weights, float* depth_map, void* stream) {
    REQUIRE(raw && z_vals && rays_d && rgb_map && disp_map && acc_map, "null pointer");
    REQUIRE(dir_stride >= 3 && n_rays >= 0 && n_samples >= 1 && n_samples <= 4096, "bad size");
    nerf::CompositeArgs a{raw, z_vals, rays_d, raw_noise_std > 0.0f ? noise : nullptr, raw_noise_std,
                          dir_stride, n_rays, n_samples, white_bkgd,
                          rgb_map, disp_map, acc_map, weights, depth_map, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    REQUIRE(!(raw_noise_std > 0.0f) || noise, "raw_noise_std > 0 needs noise draws");
    return done(__func__, nerf::launch_composite(a, false, (hipStream_t)stream));
}

int nerf_raw2outputs_bwd(const float* raw, const float* z_vals, const float* rays_d, int dir_stride, int n_rays,
                         int n_samples, const float* noise, float raw_noise_std, int white_bkgd,
                         const float* d_rgb, const float* d_acc, const float* d_disp, const float* d_weights,
                         const float* d_depth, float* d_raw, void* stream) {
    REQUIRE(raw && z_vals && rays_d && d_rgb && d_raw, "null pointer");
    REQUIRE(dir_stride >= 3 && n_rays >= 0 && n_samples >= 1 && n_samples <= 4096, "bad size");
    REQUIRE(!(raw_noise_std > 0.0f) || noise, "raw_noise_std > 0 needs noise draws");
    nerf::CompositeArgs a{raw, z_vals, rays_d, raw_noise_std > 0.0f ? noise : nullptr, raw_noise_std,
                          dir_stride, n_rays, n_samples, white_bkgd,
                          nullptr, nullptr, nullptr, nullptr, nullptr, d_rgb, d_acc, d_disp, d_raw, d_weights, d_depth};
    return done(__func__, nerf::launch_composite(a, true, (hipStream_t)stream));
}

int nerf_sample_fine(const float* z_vals, const float* weights, int n_rays, int n_coarse, int n_fine,
                     const float* u, const float* u_lin, float* z_all, float* z_samples, float* z_std, void* stream) {
    REQUIRE(z_vals && weights && z_all && z_std, "null pointer");
    REQUIRE(u || u_lin, "need u or u_lin");
    REQUIRE(n_rays >= 0 && n_coarse >= 3 && n_fine >= 1 && n_coarse + n_fine <= 8192, "bad size");
    nerf::FineArgs a{z_vals, weights, u, u_lin, z_all, z_samples, z_std, n_rays, n_coarse, n_fine, 0};
    return done(__func__, nerf::launch_sample_fine(a, (hipStream_t)stream));
}

int nerf_sample_pdf(const float* bins, const float* weights, int n_rays, int n_bins, int n_samples,
                    const float* u, const float* u_lin, float* samples, void* stream) {
    REQUIRE(bins && weights && samples, "null pointer");
    REQUIRE(u || u_lin, "need u or u_lin");
    REQUIRE(n_rays >= 0 && n_bins >= 2 && n_samples >= 1 && n_bins + n_samples <= 8192, "bad size");
    nerf::FineArgs a{bins, weights, u, u_lin, nullptr, samples, nullptr, n_rays, n_bins, n_samples, 1};
    return done(__func__, nerf::launch_sample_fine(a, (hipStream_t)stream));
}

int nerf_field_bwd(const float* packed, const float* act, const float* d_raw, int n_rays, int n_samples,
                   float* delta, float* partial, float* grad, int accumulate, void* stream) {
    REQUIRE(packed && act && d_raw && delta && partial && grad, "null pointer");
    REQUIRE(n_rays >= 0 && n_samples >= 1, "bad size");
    REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0 && (reinterpret_cast<uintptr_t>(act) & 15) == 0 &&
            (reinterpret_cast<uintptr_t>(d_raw) & 15) == 0 && (reinterpret_cast<uintptr_t>(delta) & 15) == 0,
            "packed/act/d_raw/delta must be 16-byte aligned");
    if (int rc = check_act_for_dgrad(__func__, act, false, n_rays, n_samples)) return rc;
    tag_record(delta, 1, DELTA_ROWS_F32, n_rays, n_samples);
    return done(__func__, nerf::launch_field_bwd(packed, act, d_raw, n_rays, n_samples, delta, partial, grad, accumulate,
                                                 (hipStream_t)stream));
}

int nerf_field_dgrad(const float* packed, const float* act, const float* d_raw, int n_rays, int n_samples,
                     float* delta, void* stream) {
    REQUIRE(packed && act && d_raw && delta, "null pointer");
    REQUIRE(n_rays >= 0 && n_samples >= 1, "bad size");
    REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0 && (reinterpret_cast<uintptr_t>(act) & 15) == 0 &&
            (reinterpret_cast<uintptr_t>(d_raw) & 15) == 0 && (reinterpret_cast<uintptr_t>(delta) & 15) == 0,
            "packed/act/d_raw/delta must be 16-byte aligned");
    if (int rc = check_act_for_dgrad(__func__, act, false, n_rays, n_samples)) return rc;
    tag_record(delta, 1, DELTA_ROWS_F32, n_rays, n_samples);
    return done(__func__, nerf::launch_field_dgrad(packed, act, d_raw, n_rays, n_samples, delta, (hipStream_t)stream));
}

int nerf_field_wgrad(const float* act, const float* delta, const float* d_raw, int n_rays, int n_samples,
                     float* partial, float* grad, int accumulate, void* stream) {
    REQUIRE(act && delta && d_raw && partial && grad, "null pointer");
    REQUIRE(n_rays >= 0 && n_samples >= 1, "bad size");
    return done(__func__, nerf::launch_field_wgrad(act, delta, d_raw, n_rays, n_samples, partial, grad, accumulate, 0, 7,
                                                   (hipStream_t)stream, nullptr));
}

// the datapath the (act, delta) pair of a weight-gradient call needs, from what the library wrote into them; -1 = unknown
// buffers.  *rc != 0: the pair is known and impossible (or of another point count): refused.
static int datapath_from_tags(const char* fn, const void* act, const void* delta, int n_rays, int n_samples, int* rc) {
    BufTag ta, td;
    *rc = 0;
    const bool ka = tag_lookup(act, &ta), kd = tag_lookup(delta, &td);
    if (ka && ta.is_delta) { *rc = fail_arg(fn, "`act` is a buffer this library last wrote deltas into"); return -1; }
    if (kd && !td.is_delta) { *rc = fail_arg(fn, "`delta` is a buffer this library last wrote saved activations into"); return -1; }
    if ((ka && (ta.n_rays != n_rays || ta.n_samples != n_samples)) || (kd && (td.n_rays != n_rays || td.n_samples != n_samples))) {
        *rc = fail_arg(fn, "act / delta were written for another ray or sample count than this call's");
        return -1;
    }
    if (!(ka && kd)) return -1;
    const int dp = datapath_of(ta.kind, td.kind);
    if (dp < 0) {
        snprintf(g_err, sizeof(g_err), "%s: act layout %d (0 fp32 rows, 4 / 5 = 16-point tiles of bf16 / fp16) cannot be "
                 "contracted with delta kind %d (0 fp32 rows, 2 / 3 = tiles of bf16 / fp16): the forward and the dgrad that wrote them belong to different datapaths",
                 fn, ta.kind, td.kind);
        *rc = NERF_E_BADARG;
    }
    return dp;
}

int nerf_field_wgrad_phase(const float* act, const float* delta, const float* d_raw, int n_rays, int n_samples,
                           float* partial, float* grad, int accumulate, int datapath, int phases, const float* params,
                           void* stream) {
    REQUIRE(act && delta && d_raw && partial && grad, "null pointer");
    REQUIRE(n_rays >= 0 && n_samples >= 1 && phases >= 1 && phases <= 7 && (datapath == -1 || datapath == 0 || (datapath >= 4 && datapath <= 6)), "bad size / datapath (-1, 0, 4, 5, 6)");
    int rc;
    const int recorded = datapath_from_tags(__func__, act, delta, n_rays, n_samples, &rc);
    if (rc) return rc;
    if (datapath < 0) {
        REQUIRE(recorded >= 0, "datapath = -1 (as recorded) needs act and delta written by this library's forward / dgrad entry points");
        datapath = recorded;
    } else if (recorded >= 0 && recorded != datapath) {
        snprintf(g_err, sizeof(g_err), "%s: datapath %d requested, but act / delta were written for datapath %d (the tiling and element type of "
                 "the saved rows and deltas follow from the forward and dgrad entry points that produced them)", __func__, datapath, recorded);
        return NERF_E_BADARG;
    }
    REQUIRE(datapath == 0 || params, "the split datapaths need the canonical parameters (folded feature layer)");
    return done(__func__, nerf::launch_field_wgrad(act, delta, d_raw, n_rays, n_samples, partial, grad, accumulate,
                                                   datapath, phases, (hipStream_t)stream, params));
}

int nerf_field_dgrad_split(const float* packed3, const float* act, const float* d_raw, int n_rays, int n_samples,
                           float* delta, int split, void* stream) {
    REQUIRE(packed3 && act && d_raw && delta, "null pointer");
    REQUIRE(n_rays >= 0 && n_samples >= 1 && (split == 0 || split == 1 || split == 5), "bad size / split (0 bf16, 1 fp16, 5 fp16 with two-word saves)");
    REQUIRE((reinterpret_cast<uintptr_t>(packed3) & 15) == 0 && (reinterpret_cast<uintptr_t>(act) & 15) == 0 &&
            (reinterpret_cast<uintptr_t>(d_raw) & 15) == 0 && (reinterpret_cast<uintptr_t>(delta) & 15) == 0,
            "packed/act/d_raw/delta must be 16-byte aligned");
    if (int rc = check_act_for_dgrad(__func__, act, true, n_rays, n_samples, split == 5 ? 1 : 0)) return rc;
    tag_record(delta, 1, split == 5 ? DELTA_TILE32_F16X2 : split ? DELTA_TILE32_F16 : DELTA_TILE32_BF16, n_rays, n_samples);
    return done(__func__, nerf::launch_field_dgrad3r(packed3, act, d_raw, n_rays, n_samples, delta, split, (hipStream_t)stream));
}

/* ---- three-term split datapaths (bf16 / fp16 parts) */
int nerf_packed3_floats(void) { return nerf::PACKED3_WORDS; }

int nerf_debug_pack3_table(int* out_host) {
    REQUIRE(out_host, "null pointer");
    nerf::pack3_table_host(out_host);
    return 0;
}

int nerf_debug_pack16_table(int* out_host) {
    REQUIRE(out_host, "null pointer");
    nerf::pack16_table_host(out_host);
    return 0;
}

int nerf_field_fwd_split(const float* packed3, const float* rays, int ray_stride, const float* z_vals, int n_rays,
                         int n_samples, float* raw, float* act, int split, void* stream) {
    REQUIRE(packed3 && rays && z_vals && raw, "null pointer");
    REQUIRE(ray_stride >= 11, "rays must carry view directions (ray_stride >= 11): use_viewdirs=True architecture");
    REQUIRE(n_rays >= 0 && n_samples >= 1 && ((split >= 0 && split <= 3) || split == 5), "bad size / split (0, 1, 2, 3, 5)");
    REQUIRE(split < 2 || split == 5 || !act, "split = 2 / 3 (fp16 main term + fp8 corrections) is an inference form: act must be NULL");
    REQUIRE((long)n_rays * n_samples <= nerf::FWD16R_MAX_POINTS && (!act || (long)n_rays * n_samples <= nerf::FWD16R_MAX_SAVED_POINTS),
            "too many points for one launch (2^31 - 1; 2^26 when saving): split the ray batch");
    REQUIRE((reinterpret_cast<uintptr_t>(packed3) & 15) == 0 && (reinterpret_cast<uintptr_t>(raw) & 15) == 0 &&
            (reinterpret_cast<uintptr_t>(act) & 15) == 0, "packed/raw/act must be 16-byte aligned");
    if (act) tag_record(act, 0, split == 5 ? ACT_TILE16_F16X2 : split ? ACT_TILE16_F16 : ACT_TILE16_BF16, n_rays, n_samples);
    return done(__func__, nerf::launch_field_fwd16r(packed3, rays, ray_stride, z_vals, n_rays, n_samples, raw, act, split,
                                                    (hipStream_t)stream));
}

int nerf_range_scan(const float* buf, int n_rays, int n_samples, unsigned* words, void* stream) {
    REQUIRE(buf && words, "null pointer");
    REQUIRE(n_rays >= 0 && n_samples >= 1, "bad size");
    REQUIRE((reinterpret_cast<uintptr_t>(buf) & 15) == 0 && (reinterpret_cast<uintptr_t>(words) & 3) == 0, "buf must be 16-byte, words 4-byte aligned");
    BufTag t;
    REQUIRE(tag_lookup(buf, &t) && (t.is_delta ? (t.kind == DELTA_TILE32_F16 || t.kind == DELTA_TILE32_F16X2) : (t.kind == ACT_TILE16_F16 || t.kind == ACT_TILE16_F16X2)),
            "`buf` must be a save buffer an fp16 forward (split = 1 / 5) or a delta buffer an fp16 dgrad of this library wrote: only fp16 "
            "rows and deltas have a range to check");
    REQUIRE(t.n_rays == n_rays && t.n_samples == n_samples, "`buf` was written for another ray / sample count");
    if (n_rays == 0) return 0;
    const size_t P = (size_t)n_rays * n_samples, Pp = nerf::pad32(P);
    // (the hi part: the same offsets in the two-word layouts)  h[0..7] are contiguous; the view branch's region follows the dump region
    size_t h0, hv;
    if (t.is_delta) { const nerf::DeltaLayout3 dl = nerf::delta_layout3(P); h0 = dl.h[0]; hv = dl.hv; }
    else { const nerf::ActLayout3 al = nerf::act_layout3(P, (size_t)n_rays); h0 = al.h[0]; hv = al.hv; }
    unsigned* out = words + (t.is_delta ? 2 : 0);
    hipError_t e = nerf::launch_range_scan(reinterpret_cast<const unsigned*>(buf + h0), 8 * nerf::region_words3(Pp, nerf::W), out, (hipStream_t)stream);
    if (e == hipSuccess)
        e = nerf::launch_range_scan(reinterpret_cast<const unsigned*>(buf + hv), nerf::region_words3(Pp, nerf::WV), out, (hipStream_t)stream);
    return done(__func__, e);
}

int nerf_field_fwd_last_sample(const float* packed3, const float* rays, int ray_stride, const float* z_vals, int n_rays,
                               int n_samples, float* raw, const float* packed3_next, float* raw_next, int n_samples_next, void* stream) {
    REQUIRE(packed3 && rays && z_vals && raw, "null pointer");
    REQUIRE(ray_stride >= 11, "rays must carry view directions (ray_stride >= 11): use_viewdirs=True architecture");
    REQUIRE(n_rays >= 0 && n_samples >= 1, "bad size");
    REQUIRE((reinterpret_cast<uintptr_t>(packed3) & 15) == 0 && (reinterpret_cast<uintptr_t>(raw) & 15) == 0, "packed/raw must be 16-byte aligned");
    REQUIRE(!packed3_next || (raw_next && n_samples_next >= n_samples), "packed3_next needs raw_next and the refining pass's sample count");
    REQUIRE((reinterpret_cast<uintptr_t>(packed3_next) & 15) == 0 && (reinterpret_cast<uintptr_t>(raw_next) & 15) == 0, "packed/raw must be 16-byte aligned");
    return done(__func__, nerf::launch_field_fwd16r_last(packed3, rays, ray_stride, z_vals, n_rays, n_samples, raw, packed3_next, raw_next,
                                                         n_samples_next, (hipStream_t)stream));
}

int nerf_pack_params_split(const float* params, float* packed3, int streams, int split, void* stream) {
    REQUIRE(params && packed3, "null pointer");
    REQUIRE(streams >= 0 && (streams & ~5) == 0 && split >= 0 && split <= 2, "streams is a mask of 1 (forward stream) and 4 (transposed streams of the delta chain); split 0 (bf16), 1 (fp16) or 2 (reduced inference stream)");
    REQUIRE(split != 2 || (streams & 1), "split = 2 refills the 16-point forward stream: streams must include bit 0");
    return done(__func__, nerf::launch_pack3_sel(params, packed3, streams, (hipStream_t)stream, split));
}

int nerf_pack_params_split_pair(const float* params_a, float* packed3_a, const float* params_b, float* packed3_b, int streams, int split, void* stream) {
    REQUIRE(params_a && packed3_a && params_b && packed3_b, "null pointer");
    REQUIRE(packed3_a != packed3_b, "the two networks need their own packed buffers");
    REQUIRE(streams >= 0 && (streams & ~5) == 0 && (split == 0 || split == 1), "streams is a mask of 1 (forward stream) and 4 (transposed streams); split 0 (bf16) or 1 (fp16)");
    return done(__func__, nerf::launch_pack3_pair(params_a, packed3_a, params_b, packed3_b, streams, (hipStream_t)stream, split));
}


int nerf_mse_scratch_floats(void) { return nerf::MSE_SCRATCH_FLOATS; }
int nerf_mse_fwd(const float* x, const float* y, long n, float* scratch, float* out, void* stream) {
    REQUIRE(x && y && scratch && out, "null pointer");
    REQUIRE(n > 0, "bad size");
    return done(__func__, nerf::launch_mse_fwd(x, y, n, scratch, out, (hipStream_t)stream));
}
int nerf_mse_bwd(const float* x, const float* y, long n, const float* grad_out, float* dx, void* stream) {
    REQUIRE(x && y && grad_out && dx, "null pointer");
    REQUIRE(n > 0, "bad size");
    return done(__func__, nerf::launch_mse_bwd(x, y, n, grad_out, dx, (hipStream_t)stream));
}

int nerf_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, float lr, float beta1,
                   float beta2, float eps, int step, void* stream) {
    REQUIRE(params && grads && exp_avg && exp_avg_sq, "null pointer");
    REQUIRE(n >= 0 && step >= 1 && beta1 >= 0.0f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f, "bad argument");
    return done(__func__, nerf::launch_adam(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step,
                                            (hipStream_t)stream));
}

}  // extern "C"
