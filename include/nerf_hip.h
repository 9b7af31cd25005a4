/* nerf_hip.h -- C ABI of libnerf_hip.so, the MI355X (gfx950) implementation of the
 * nerf-pytorch volumetric-rendering hot path.
 *
 * The reference (yenchenlin/nerf-pytorch) is pure Python and has no FFI: the seam this
 * library plugs into is the call surface of its hot-path functions.  Every entry point
 * below names the reference function (file:line under /root/reference) it replaces.
 * A maintainer binds these with ctypes (INTEGRATION.md shows the stub); the in-repo
 * binding is nerf-pytorch_amd/hip_backend.py.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 data owned by the caller (torch tensors);
 *     the library allocates nothing and keeps no pointer after return;
 *   - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream); all work is
 *     enqueued asynchronously on it, no host synchronisation inside;
 *   - return 0 on success, a negative NERF_E_* code for argument errors, a positive
 *     hipError_t for runtime errors; nerf_last_error() describes the last failure of the
 *     calling thread.  Nothing throws, nothing exits;
 *   - rays are [n_rays][ray_stride] records (o3, d3, near, far, viewdir3), ray_stride = 11
 *     (run_nerf.py:117-123);
 *   - the field model is fixed to the reference architecture NeRF(D=8, W=256,
 *     input_ch=63, input_ch_views=27, skips=[4], use_viewdirs=True)
 *     (run_nerf_helpers.py:67-94); parameters are exchanged as ONE flat fp32 vector in
 *     state_dict order (nerf_param_count() = 595,844 floats, layout: nerf_param_offset()).
 */
#ifndef NERF_HIP_H
#define NERF_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define NERF_ABI_VERSION 10
#define NERF_E_BADARG (-1)      /* null pointer / non-positive size / unsupported shape */
#define NERF_E_UNSUPPORTED (-2) /* configuration outside the fixed architecture */

int nerf_abi_version(void);
const char* nerf_last_error(void);

/* ---- parameter vector (replaces nn.Module storage, run_nerf_helpers.py:79-94) ---------- */
int nerf_param_count(void);
/* offset (floats) of tensor #idx of the state_dict (0..23: pts_linears.0.weight, .bias, ...,
 * views_linears.0.*, feature_linear.*, alpha_linear.*, rgb_linear.*); returns -1 if idx is out of range.
 * rows/cols receive the tensor shape (cols = 1 for biases). */
int nerf_param_offset(int idx, int* rows, int* cols);
/* size of the MFMA-fragment repack of one network (floats) */
int nerf_packed_floats(void);
/* canonical parameters -> fragment streams consumed by nerf_field_fwd / nerf_field_bwd.
 * Call after every optimizer step. */
int nerf_pack_params(const float* params, float* packed, void* stream);

/* test hook (host only, no GPU): out_host[i] = index into the canonical vector that packed[i] is gathered
 * from, or -1 for zero padding; nerf_packed_floats() entries. */
int nerf_debug_pack_table(int* out_host);

/* ---- Embedder.embed (run_nerf_helpers.py:15-45): x[n_pts][3] -> out[n_pts][3 + 6*n_freqs] */
int nerf_embed(const float* x, long n_pts, int n_freqs, float* out, void* stream);

/* ---- z_vals of render_rays (run_nerf.py:357-379).  t_vals = torch.linspace(0,1,n_samples);
 * t_rand [n_rays][n_samples] uniform draws or NULL (perturb == 0). */
int nerf_sample_coarse(const float* rays, int ray_stride, int n_rays, const float* t_vals, int n_samples,
                       int lindisp, const float* t_rand, float* z_vals, void* stream);

/* ---- ray set-up of render(c2w=...) (run_nerf.py:95-123): get_rays (run_nerf_helpers.py:153-162), normalised view
 * directions (:100-107; taken from c2w even when c2w_staticcam supplies the rays), ndc_rays(H, W, K[0][0], 1., ...)
 * (helpers:175-192) when ndc != 0, and the near / far columns, written as rays[H*W][ray_stride] records
 * (o3, d3, near, far, viewdir3), pixel (row j, column i) at j*W + i.  K_host: HOST pointer to the 3x3 intrinsics,
 * row-major; c2w_host / c2w_staticcam_host (nullable): HOST pointers to 3x4 row-major poses (they travel as kernel
 * arguments; no device copy).  rays: device. */
int nerf_make_rays(int H, int W, const float* K_host, const float* c2w_host, const float* c2w_staticcam_host, int ndc,
                   float near, float far, float* rays, int ray_stride, void* stream);

/* ---- ray records of render(rays=(rays_o, rays_d), use_viewdirs=True) (run_nerf.py:95-123, no c2w / c2w_staticcam):
 * rays_o / rays_d [n_rays][3] device, fp32, contiguous (world space).  View directions rays_d / |rays_d| (:100-107), then
 * ndc_rays(H, W, focal, 1., ...) (run_nerf_helpers.py:175-192) iff ndc != 0, then the near / far columns (:117-123):
 * rays[n_rays][ray_stride] = (o3, d3, near, far, viewdir3).  focal = K[0][0]; H, W, focal are read only when ndc != 0. */
int nerf_assemble_rays(const float* rays_o, const float* rays_d, long n_rays, int ndc, int H, int W, float focal, float near,
                       float far, float* rays, int ray_stride, void* stream);

/* ---- the ray batch of one train() step in the reference's 'no_batching' mode (run_nerf.py:726-757): n_rand DISTINCT pixels of one
 * image -- of the window (h0, w0, nh, nw): the central crop of the first precrop_iters steps (:738-747), or the whole image --, their
 * rays as get_rays gives them (run_nerf_helpers.py:153-162) and their colours:
 *   batch_rays[2][n_rand][3] = (rays_o, rays_d), target[n_rand][3] = image[j][i], pixels[n_rand] = j * W + i (nullable).
 * The reference builds the [H,W,3] ray grid and a meshgrid and draws np.random.choice(H*W, N_rand, replace=False) on the host every
 * step; here pixel k of the batch is perm(k), a keyed bijection of [0, nh*nw) (6-round Feistel network, cycle-walked): distinct by
 * construction, and WHICH subset is decided by (key0, key1), two words the host draws per step from its own generator -- no O(H*W)
 * work, no device synchronisation.  K_host: HOST 3x3 intrinsics; pose: DEVICE c2w[:3,:4], rows pose_row_stride floats apart (a
 * slice of a [N,4,4] pose table stays where it is); image: DEVICE [H][W][3] fp32. */
int nerf_sample_ray_batch(int H, int W, const float* K_host, const float* pose, int pose_row_stride, const float* image, int h0,
                          int w0, int nh, int nw, int n_rand, unsigned key0, unsigned key1, float* batch_rays, float* target,
                          int* pixels /* nullable */, void* stream);

/* ---- network_query_fn(pts, viewdirs, network_fn) with pts = o + d*z
 * (run_nerf.py:381,385 -> run_network :37-51 -> Embedder :44-45 -> NeRF.forward helpers:96-119).
 * raw[n_rays][n_samples][4] = (rgb pre-sigmoid, sigma pre-relu).
 * act: NULL for inference; otherwise nerf_act_floats() floats that receive what the backward needs. */
size_t nerf_act_floats(int n_rays, int n_samples);
/* Scratch of one training call of render_rays (run_nerf.py:308-418) on n_rays rays, in floats: the saved activations of
 * the coarse pass (n_coarse samples) and of the fine pass (n_coarse + n_fine samples; none if n_fine == 0), plus the
 * deltas and the per-chunk partial weight gradients of the larger pass (reused by both backward calls).  The caller
 * owns this memory (persistent across steps; any datapath); 0 when training == 0 -- inference needs no scratch.
 * = nerf_act_floats(coarse) + nerf_act_floats(fine) + nerf_delta_floats(larger) + nerf_wgrad_partial_floats(larger). */
size_t nerf_workspace_floats(int n_rays, int n_coarse, int n_fine, int training);
/* The same three sizes for ONE datapath (ABI v9).  datapath: 0 = exact fp32 (fp32 rows, 10.6 KB / point each way), 1 = the split
 * datapaths (16-bit tiles of either type, 4.8 / 4.4 KB / point), 2 (ABI v10) = the fp16 split with TWO-WORD saves (hi and lo words:
 * twice datapath 1); the two-argument forms above return the larger of datapaths 0 and 1, i.e. a buffer either of those may write.  A caller that knows its datapath keeps ~2.2x more rays resident per GB of HBM with these:
 * the 32,768-ray batch of BASELINE configs[3] holds ~40 GB of saved activations on the split datapaths instead of ~90 GB. */
size_t nerf_act_floats_dp(int n_rays, int n_samples, int datapath);
size_t nerf_delta_floats_dp(int n_rays, int n_samples, int datapath);
size_t nerf_workspace_floats_dp(int n_rays, int n_coarse, int n_fine, int training, int datapath);
int nerf_field_fwd(const float* packed, const float* rays, int ray_stride, const float* z_vals, int n_rays,
                   int n_samples, float* raw, float* act, void* stream);

/* ---- raw2outputs (run_nerf.py:262-305).  rays_d points at the first direction component of ray 0,
 * consecutive rays are dir_stride floats apart (3 for a bare [N,3] tensor, 11 for rays + 3).
 * noise: standard-normal draws [n_rays][n_samples] or NULL; weights / depth_map may be NULL. */
int nerf_raw2outputs(const float* raw, const float* z_vals, const float* rays_d, int dir_stride, int n_rays,
                     int n_samples, const float* noise, float raw_noise_std, int white_bkgd, float* rgb_map,
                     float* disp_map, float* acc_map, float* weights, float* depth_map, void* stream);
/* autograd of raw2outputs w.r.t. raw (all five outputs of run_nerf.py:305 are differentiable in the reference):
 * d_rgb[n_rays][3] required; d_acc / d_disp / d_depth [n_rays] and d_weights [n_rays][n_samples] may be NULL. */
int nerf_raw2outputs_bwd(const float* raw, const float* z_vals, const float* rays_d, int dir_stride, int n_rays,
                         int n_samples, const float* noise, float raw_noise_std, int white_bkgd,
                         const float* d_rgb, const float* d_acc, const float* d_disp, const float* d_weights,
                         const float* d_depth, float* d_raw, void* stream);

/* ---- hierarchical sampling (run_nerf.py:392-396,412 + sample_pdf, run_nerf_helpers.py:196-239):
 * z_mid, sample_pdf(z_mid, weights[1:-1], n_fine, det = (u == NULL)), sort(cat(z_vals, z_samples)), z_std.
 * u [n_rays][n_fine] uniform draws or NULL; u_lin = torch.linspace(0,1,n_fine) (read when u == NULL).
 * z_all [n_rays][n_coarse+n_fine]; z_samples [n_rays][n_fine] may be NULL. */
int nerf_sample_fine(const float* z_vals, const float* weights, int n_rays, int n_coarse, int n_fine,
                     const float* u, const float* u_lin, float* z_all, float* z_samples, float* z_std, void* stream);

/* ---- sample_pdf (run_nerf_helpers.py:196-239) in its standalone form: bins[n_rays][n_bins],
 * weights[n_rays][n_bins-1] -> samples[n_rays][n_samples]; det = (u == NULL). */
int nerf_sample_pdf(const float* bins, const float* weights, int n_rays, int n_bins, int n_samples,
                    const float* u, const float* u_lin, float* samples, void* stream);

/* ---- backward of nerf_field_fwd: d_raw[n_rays][n_samples][4] -> gradient of the flat parameter vector
 * (autograd of run_nerf_helpers.py:96-119; parameters only).
 * delta / partial: scratch of nerf_delta_floats() / nerf_wgrad_partial_floats() floats.
 * accumulate != 0 adds into grad, otherwise grad is overwritten. */
size_t nerf_delta_floats(int n_rays, int n_samples);
size_t nerf_wgrad_partial_floats(int n_rays, int n_samples);
int nerf_field_bwd(const float* packed, const float* act, const float* d_raw, int n_rays, int n_samples,
                   float* delta, float* partial, float* grad, int accumulate, void* stream);

/* the two halves of nerf_field_bwd, separately launchable (bench.py brackets each with HIP events):
 * dgrad: d_raw -> per-layer deltas;  wgrad: (deltas, saved activations) -> parameter gradient. */
int nerf_field_dgrad(const float* packed, const float* act, const float* d_raw, int n_rays, int n_samples,
                     float* delta, void* stream);
int nerf_field_wgrad(const float* act, const float* delta, const float* d_raw, int n_rays, int n_samples,
                     float* partial, float* grad, int accumulate, void* stream);

/* ---- the three-term SPLIT datapaths of the same functions (csrc/split_types.h): every product W x of the MLP is evaluated
 * as  W_hi x_hi + W_hi x_lo + W_lo x_hi  on 16-bit MFMAs with fp32 accumulation, hi = T(v), lo = T(v - hi):
 *   split = 1, T = IEEE half ("fp16x3", the host code's default): v_mfma_f32_16x16x32_f16 / 32x32x16_f16, ~2^-22 per product --
 *              fp32-class; the rows / deltas saved for the weight-gradient GEMM are the hi words (11 significant bits);
 *   split = 0, T = bfloat16 ("bf16x3"): ~2^-17 per product, 8-bit saved operands; fp32's exponent range (no overflow possible).
 * Judged by the north-star PSNR criterion (measured: 4e-6 / 2.4e-4 dB) and by fp64 comparisons of the gradients (tests/).
 * Parameters are repacked into (hi, lo) fragment streams of nerf_packed3_floats() 32-bit words by nerf_pack_params_split
 * (streams = mask of 1: 16-point forward stream, 4: transposed streams of the delta chain -- 5 = everything these entry points
 * read).  feature_linear is
 * FOLDED into the view branch: W' = Wv[:, :256] Wf, b' = Wv[:, :256] bf + bv are derived at pack time (helpers:111-115: no activation
 * between the two layers); `feature` and its delta are neither computed nor saved, and the weight-gradient entry point takes the
 * canonical parameter vector `params` (the one that was packed) to produce the gradients of Wf, bf and Wv[:, :256] from
 * G = delta_hv^T h7:  dWv[:, :256] = G Wf^T + dbv bf^T,  dWf = Wv[:, :256]^T G,  dbf = Wv[:, :256]^T dbv.
 *   Kernels: 16 points per wavefront at 2 waves / SIMD, weights through a 17-slot LDS ring of 8 KiB fragment units (csrc/field_ring.h);
 * the delta chain 32 points per wavefront on the same ring.  The save buffers hold 16-bit elements in tiles (rows of the 256- /
 * 128-wide regions in 16-point tiles, row16h order; deltas and encodings in 32-point feature-major tiles; csrc/nerf_common.h), the
 * ReLU bitmask words are in these kernels' lane order: forward, dgrad and weight gradients of one evaluation must use the same
 * split (nerf_field_wgrad_phase(datapath = -1) picks the GEMM from the buffer records: 4 = bf16 operands, 5 = fp16 operands).
 *   Range (split = 1): weights, encodings and activations must stay below 65520 in magnitude (a NeRF's are O(1..100)); an overflow
 * turns into inf / NaN in `raw` (the ReLU propagates NaN).  Deltas are tiny (upstream gradients ~1e-6): nerf_field_dgrad_split(split
 * = 1) first reduces max|d_raw| on the device and runs the chain on s * d_raw with s the power of two that puts that maximum in
 * [16, 32) (the chain is linear; the maximum lives in the delta buffer), every stored delta and partial weight gradient carries s, and
 * the reduction phase of nerf_field_wgrad_phase multiplies by 1/s -- exact.  A non-finite d_raw propagates to the gradient.
 *   split = 5 (ABI v10, "fp16x3w"; nerf_field_fwd_split with act, nerf_field_dgrad_split): the fp16 split with TWO-WORD saves -- the
 * forward and the delta chain store the lo words (T(v - hi)) next to the hi words (mirror regions at ActLayout3::lo /
 * DeltaLayout3::lo; buffers of nerf_act_floats_dp / nerf_delta_floats_dp(datapath = 2) floats), and the weight-gradient GEMM
 * (nerf_field_wgrad_phase datapath 6) contracts  d_hi^T X_hi + d_hi^T X_lo + d_lo^T X_hi : the products of the forward's class
 * (~2^-22) instead of 11-bit operands, at twice the saved bytes and three times the GEMM's MFMAs.  Same packed3 as split = 1.  Not
 * the default: it prices and bounds what the one-word operand storage costs the gradients.
 * Replace run_nerf.py:37-51 + run_nerf_helpers.py:15-45, :96-119 and their autograd like nerf_field_fwd / nerf_field_bwd. */
int nerf_packed3_floats(void);
int nerf_pack_params_split(const float* params, float* packed3, int streams, int split, void* stream);
/* the repack of TWO networks (a training step's coarse and fine network after the optimizer step) in the same two launches as one
 * (ABI v9; split 0 / 1; results identical to two nerf_pack_params_split calls) */
int nerf_pack_params_split_pair(const float* params_a, float* packed3_a, const float* params_b, float* packed3_b, int streams,
                                int split, void* stream);
int nerf_field_fwd_split(const float* packed3, const float* rays, int ray_stride, const float* z_vals, int n_rays,
                         int n_samples, float* raw, float* act /* nullable: inference */, int split, void* stream);
int nerf_field_dgrad_split(const float* packed3, const float* act, const float* d_raw, int n_rays, int n_samples,
                           float* delta, int split, void* stream);
/* Range check of the fp16 split (ABI v10; replaces the reference's DEBUG-gated NaN / Inf check, run_nerf.py:414-416, with a warning
 * that can come BEFORE the NaN).  buf = what a SAVING fp16 forward left in `act` (nerf_field_fwd_split(split = 1 / 5) with act: the
 * post-ReLU rows of the eight trunk layers and of the view branch, i.e. every value that enters a contraction as an fp16 operand), or
 * what an fp16 delta chain left in `delta` (nerf_field_dgrad_split(split = 1 / 5): the scaled deltas of the same layers, by magnitude).
 * Merges into FOUR caller-owned 32-bit device words: rows -> words[1] = max(words[1], largest fp16 bit pattern), words[0] |= 1 when that
 * pattern is >= 0x7800 (32768: half way to the 65520 beyond which `raw` / the gradient turn NaN); deltas -> words[3], words[2] likewise
 * (NaN patterns compare above inf).  One streaming read (4.2 KB per point at HBM rate: 0.6 ms for 786 k points); the hot kernels carry no
 * monitor (a running maximum costs the saving forward the 6 registers it has left: measured, spills).  The host code scans every
 * NERF_RANGE_CHECK_EVERY-th training render (default 64: 0.2 % of the step time) and reads the words without synchronising. */
int nerf_range_scan(const float* buf, int n_rays, int n_samples, unsigned* words, void* stream);
/* test hooks (host only): gather tables of the fragment streams -- out_host[e] for every 16-bit element e: 2 * canonical_index +
 * is_low_part, or -1 for zero padding (nerf_debug_pack3_table: the (hi, lo) streams incl. the transposed ones, declared below;
 * nerf_debug_pack16_table: the 16-point forward stream). */
int nerf_debug_pack16_table(int* out_host);
/* ---- reduced product class for INFERENCE: split = 2 of nerf_pack_params_split / nerf_field_fwd_split (act must be NULL).
 * Every product of the 256-wide contractions (layers 1..7, the trunk part of the view branch) is  W_hi16 x_hi16  on the fp16 MFMA
 * plus the two correction terms  W_hi8 x_lo8 + W_lo8 x_hi8  as block-scaled fp8 e4m3 MFMAs of K = 128 (v_mfma_scale_f32_16x16x128_
 * f8f6f4): ~2^-15 per product at 2 instead of 3 MFMA-equivalents (csrc/field_ring8.h).  Admitted for inference by the north-star
 * gate with >= 30x margin on both fixtures (profiles/r04_accuracy_classes.md), never used for training.  Activations must stay
 * below 224 in magnitude (NaN beyond).  split = 3: the same pass with sample n_samples - 1 of every ray left unwritten.
 *   nerf_field_fwd_last_sample evaluates every ray's LAST sample of a pass with the three-term fp16 products (packed3 = the
 * split = 1 repack of the same parameters) into the pass's raw: the reference's dists[-1] = 1e10 (run_nerf.py:277-278) makes that
 * sample's alpha a step function of the sign of its density (:293), the one place where a 2^-15 error can move a ray's opacity
 * by O(1).  Call it after a split = 2 pass or in any order with a split = 3 pass, before raw2outputs.
 *   packed3_next (nullable): in the same launch, the last sample of the hierarchical pass that refines this one, evaluated by THAT
 * pass's network into raw_next[n_rays][n_samples_next][4].  Its depth is this pass's last depth: sample_pdf draws inside
 * [z_mid[0], z_mid[-1]] (run_nerf.py:392-396, helpers:196-239), so the sorted union of run_nerf.py:396 ends with z_vals[:, -1].
 * (One launch costs one pass of a workgroup through the network, ~70 us, for one network or two.) */
int nerf_field_fwd_last_sample(const float* packed3, const float* rays, int ray_stride, const float* z_vals, int n_rays,
                               int n_samples, float* raw, const float* packed3_next /* nullable */, float* raw_next,
                               int n_samples_next, void* stream);
/* ---- which layout a scratch buffer holds.  The save buffer of a forward and the delta buffer of a dgrad exist in three layouts each
 * (fp32 point-major rows; bf16 tiles; fp16 tiles: csrc/nerf_common.h); the entry point that writes a buffer decides, and the entry
 * points that read it must agree.  The library remembers per buffer ADDRESS what its own entry points last wrote there (host-side
 * table only; nothing is stored on the device and no pointer is dereferenced later) and
 *   - every dgrad / weight-gradient entry point returns NERF_E_BADARG when given a buffer of the wrong family, of another
 *     ray / sample count, or an (act, delta) pair that no datapath contracts -- instead of computing garbage;
 *   - nerf_field_wgrad_phase(datapath = -1) takes the datapath from the record.
 * Buffers the library has not written (copies, foreign producers) are not checked.
 * nerf_buffer_layout: the recorded kind, or -1 for an unknown buffer.  act: 0 fp32 point-major rows (nerf_field_fwd), 4 / 5 = rows
 * in 16-point tiles of bf16 / fp16 (nerf_field_fwd_split(split = 0 / 1)), 6 = fp16 hi + lo rows (split = 5); delta (*is_delta = 1):
 * 0 fp32 rows, 2 / 3 = 32-point tiles of bf16 / fp16, 4 = fp16 hi + lo tiles (split = 5). */
int nerf_buffer_layout(const float* buf, int* is_delta, int* n_rays, int* n_samples);
/* test hook (host only): where the regions of such a buffer start, in floats from its base, for n_rays x n_samples points.
 * family 0 = fp32 point-major rows, 1 = the split datapaths' tiles of 16-bit elements, 2 = the same with two-word saves (the offsets
 * of family 1 plus [15] = offset of the mirror that holds the lo words; [14] total = twice family 1's).  out_host[16]:
 *   save buffer  (is_delta = 0): [0..7] post-ReLU rows of trunk layers 0..7, [8] feature (fp32 rows only; the split datapaths fold
 *                that layer and use the slot as the dump tile of out-of-range waves), [9] view branch, [10] xyz encoding,
 *                [11] direction encoding per ray, [12] its per-point / per-ray-record expansion, [13] ReLU bitmasks, [14] total;
 *   delta buffer (is_delta = 1): [0..7], [8], [9] as above, [10] tiled copy of d_raw, [11] the launch's max|d_raw| word
 *                (family 1 only), [14] total.   Unused slots are -1. */
int nerf_debug_layout(int n_rays, int n_samples, int family, int is_delta, long long* out_host);
/* The weight gradients dW = delta^T x, db = sum delta of one evaluation (the second half of nerf_field_bwd; the only form on the
 * split datapaths), split into launches a profiler can bracket: phases bit 0 = the GEMM jobs (fp32: the eight full-width 256x256
 * jobs; split datapaths: all jobs on the streaming 16-bit GEMM), bit 1 = the six narrow jobs (fp32 datapath only), bit 2 =
 * reduction of the per-chunk partial gradients into grad (deterministic, no atomics; + the folded feature layer's gradients).
 * Calling it with phases 1, 2, 4 in that order equals one call with 7. */
int nerf_field_wgrad_phase(const float* act, const float* delta, const float* d_raw, int n_rays, int n_samples,
                           float* partial, float* grad, int accumulate,
                           int datapath /* -1 as recorded for act / delta (above); 0 fp32; 4 bf16 operands; 5 fp16 operands;
                                           6 fp16 two-word operands (split = 5 buffers) */,
                           int phases, const float* params /* canonical parameters; may be NULL for datapath 0 */,
                           void* stream);
/* ---- render_rays in one call (run_nerf.py:308-418 and its autograd): the whole of a ray batch's forward, and the whole of
 * its backward, as ONE entry point each.  They chain the launches above in C -- coarse depths -> field -> raw2outputs
 * [-> sample_pdf + sort -> field -> raw2outputs]; raw2outputs' adjoint -> delta chain -> weight gradients per pass -- in the
 * order the in-repo binding issues them, so the results are bit-identical to it.  What a host provides: the ray records, the
 * random draws in the reference's order (NULL where the reference draws nothing), the packed and canonical parameters, the
 * outputs, and ONE scratch buffer of nerf_render_workspace_floats() floats that lives from the forward to its backward
 * (depths, coarse raw, compositing weights; when training also the saved activations, deltas and partial gradients: ~9.2 KB
 * per sample point on the split datapaths, ~21 KB on fp32 -- split larger ray batches, the reference's `chunk` argument does exactly that).
 *   precision 0: exact fp32 datapath; 1: split-bf16; 3: split-fp16 (packed buffers from nerf_pack_params_split with the
 *   matching split; 16-bit operand storage of the weight-gradient GEMM either way); 5: split-fp16 with two-word saves and the
 *   three-term weight-gradient GEMM (packed buffers of split 1; workspace of nerf_render_workspace_floats for THIS cfg).
 *   Backward with accumulate = 0: every gradient vector handed in is written -- a network whose pass received no upstream
 *   gradient gets zeros; d_disp / d_acc may be given without d_rgb.
 *   packed_f / params_f / grad_f NULL (or packed_f == packed_c): the fine pass uses the coarse network (network_fine None).
 *   Outputs as the reference's dict: rgb/disp/acc/raw = the last pass (rgb_map, disp_map, acc_map, raw [n, n_coarse + n_fine, 4]);
 *   rgb0/disp0/acc0/z_std = the coarse pass and the std of the fine samples (n_fine > 0 only).
 *   Backward: upstream gradients of the same outputs (any may be NULL; d_raw_out = gradient of `raw`), the forward's `raw`,
 *   noise draws and workspace; the parameter gradients are written (accumulate = 0) or added into grad_c / grad_f. */
typedef struct NerfRenderCfg {
    int n_coarse, n_fine;          /* N_samples, N_importance */
    int lindisp, white_bkgd;
    float raw_noise_std;
    int precision;                 /* 0 fp32, 1 split-bf16, 3 split-fp16, 5 split-fp16 with two-word saves */
    int reserved;                  /* (round 3: operand storage switch; ignored) */
} NerfRenderCfg;
size_t nerf_render_workspace_floats(const NerfRenderCfg* cfg, int n_rays, int training);
int nerf_render_rays_fwd(const NerfRenderCfg* cfg, const float* packed_c, const float* packed_f, const float* rays, int ray_stride,
                         int n_rays, const float* t_rand, const float* noise_c, const float* u, const float* noise_f,
                         float* rgb, float* disp, float* acc, float* raw, float* rgb0, float* disp0, float* acc0, float* z_std,
                         float* workspace, int training, void* stream);
/* nerf_render_rays_fwd(training = 0) as ONE kernel launch (csrc/render_fused.hip): a workgroup owns 16 rays from the coarse
 * depths (run_nerf.py:357-379) through both networks, raw2outputs (:262-305) and sample_pdf + sort (:392-396) to the final
 * colours.  Same device code as the separate launches: every output is bit-identical to nerf_render_rays_fwd.  Same arguments
 * and workspace (nerf_render_workspace_floats(cfg, n_rays, 0)).  Split datapaths only (precision 1 / 3), and only sample counts
 * for which 16 rays fill whole 128-point tiles in both passes (n_coarse and n_coarse + n_fine multiples of 8, at most 1024
 * samples per ray): nerf_render_infer_supported(cfg) tells; otherwise NERF_E_BADARG. */
int nerf_render_infer_supported(const NerfRenderCfg* cfg);
int nerf_render_rays_infer(const NerfRenderCfg* cfg, const float* packed_c, const float* packed_f, const float* rays, int ray_stride,
                           int n_rays, const float* t_rand, const float* noise_c, const float* u, const float* noise_f,
                           float* rgb, float* disp, float* acc, float* raw, float* rgb0, float* disp0, float* acc0, float* z_std,
                           float* workspace, void* stream);
int nerf_render_rays_bwd(const NerfRenderCfg* cfg, const float* packed_c, const float* packed_f, const float* params_c,
                         const float* params_f, const float* rays, int ray_stride, int n_rays, const float* noise_c,
                         const float* noise_f, const float* raw, const float* d_rgb, const float* d_disp, const float* d_acc,
                         const float* d_raw_out, const float* d_rgb0, const float* d_disp0, const float* d_acc0,
                         float* workspace, float* grad_c, float* grad_f, int accumulate, void* stream);

/* ---- architectures outside the fused kernels (csrc/dense.hip): the reference's layer stack (run_nerf_helpers.py:96-119) for any
 * netdepth / netwidth / skips / multires / multires_views / i_embed = -1 / use_viewdirs = False (run_nerf.py:435-442,
 * helpers:48-50, :93-94, :117), one layer per call.  Row-major matrices with explicit leading dimensions (a layer may read or
 * write a column range of a wider matrix: skip connection, view branch, rgb | alpha -- no concatenations).  The GEMMs are
 * plain library SGEMMs (rocBLAS, exact fp32, atomics off), loaded on first use; the epilogues are HIP kernels.
 *   nerf_build_inputs : x[p] = [enc(o + d z_p) | enc(viewdir)] for every sample point (run_nerf.py:381, :41-47);
 *                       multires / multires_views = number of frequencies, -1 = identity (i_embed = -1);
 *                       rays [N][ray_stride] = (o3, d3, near, far[, viewdir3 in the LAST three columns])
 *   nerf_dense_fwd    : y[P,N] (+)= x[P,K] w[N,K]^T (+ bias) (then ReLU)               = F.linear (+ F.relu), helpers:99-100
 *   nerf_dense_dgrad  : dx[P,K] (+)= dy[P,N] w[N,K]; then dx *= (act > 0) if act       (act = post-ReLU output of the layer below)
 *   nerf_dense_wgrad  : dw[N,K] (+)= dy^T x; dbias[N] (+)= column sums of dy (deterministic two-pass sum; scratch of
 *                       nerf_dense_wgrad_scratch_floats(P, N) floats) */
int nerf_build_inputs(const float* rays, int ray_stride, const float* z_vals, int n_rays, int n_samples, int multires, int multires_views,
                      int use_viewdirs, float* x, int ldx, void* stream);
int nerf_dense_fwd(const float* x, int ldx, int K, const float* w, int ldw, const float* bias, float* y, int ldy, int N, long P,
                   int accumulate, int relu, void* stream);
int nerf_dense_dgrad(const float* dy, int lddy, int N, const float* w, int ldw, float* dx, int lddx, int K, long P, int accumulate,
                     const float* act, int ldact, void* stream);
size_t nerf_dense_wgrad_scratch_floats(long P, int N);
int nerf_dense_wgrad(const float* dy, int lddy, int N, const float* x, int ldx, int K, long P, float* dw, int lddw, float* dbias,
                     float* scratch, int accumulate, void* stream);

/* ---- img2mse (run_nerf_helpers.py:11: torch.mean((x - y) ** 2), the loss of run_nerf.py:765-772) in one launch, and its
 * gradient w.r.t. x in one more: out[0] = mean((x - y)^2) over n elements (deterministic block-ordered sum); dx = (2 g / n)(x - y)
 * with g = grad_out[0] read on the device.  scratch: nerf_mse_scratch_floats() floats, ZERO before its first use (the kernel
 * leaves it zeroed). */
int nerf_mse_scratch_floats(void);
int nerf_mse_fwd(const float* x, const float* y, long n, float* scratch, float* out, void* stream);
int nerf_mse_bwd(const float* x, const float* y, long n, const float* grad_out, float* dx, void* stream);

/* ---- optimizer.step() of run_nerf.py:776 for torch.optim.Adam(lr, betas=(beta1, beta2), eps) (run_nerf.py:207), fused over
 * a flat vector: params / grads / exp_avg / exp_avg_sq [n]; step = 1-based step count (bias correction). */
int nerf_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, float lr, float beta1,
                   float beta2, float eps, int step, void* stream);
/* test hook (host only): out_host[e] for every 16-bit element e of the weight streams (2 * stream words):
 * 2 * canonical_index + is_low_part, or -1 for zero padding. */
int nerf_debug_pack3_table(int* out_host);

#ifdef __cplusplus
}
#endif
#endif
